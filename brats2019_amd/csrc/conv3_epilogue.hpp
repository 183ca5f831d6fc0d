// conv3_epilogue.hpp -- shared epilogue of the 3x3x3 convolution kernels (f32 and split-bf16): each lane holds, per
// accumulator, 4 consecutive x voxels (rows (lane>>4)*4 + r) of one output channel (column lane&15).  Adds bias /
// residual, accumulates per-tile (sum, sumsq) for the consumer GroupNorm, applies the sigmoid, stores 16 bytes.
#pragma once
#include "ru_common.h"

namespace ru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_swizzle(int b, int nb) {
    // give each XCD (block b runs on XCD b % 8) a contiguous run of tiles so halo re-reads hit its L2
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <int MT, int NT>
__device__ __forceinline__ void conv3_epilogue(const Conv3Args& a, f32x4 (&acc)[MT][NT], float* smem, int n, int z0, int y0, int x0,
                                               int mz, int my0, int co0, int tz, int ty, int tx, int ntz, int nty, int ntx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const bool vec = (W & 3) == 0;
    const int zz = z0 + mz;
    const int xq = x0 + (lane >> 4) * 4;
    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int yy = y0 + my0 + i;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = co0 + t * 16 + (lane & 15);
            const bool ok = zz < D && yy < H && co < a.Cout && xq < W;
            if (!ok) continue;
            const size_t idx = (((size_t)n * a.Cout + co) * D + zz) * HW + (size_t)yy * W + xq;
            f32x4 v = acc[i][t];
            if (a.bias) { const float bv = a.bias[co]; v += bv; }
            const int nvalid = (W - xq) < 4 ? (W - xq) : 4;
            if (a.add) {
                if (vec) {
                    const float4 r4 = *reinterpret_cast<const float4*>(a.add + idx);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                } else {
                    for (int r = 0; r < nvalid; ++r) v[r] += a.add[idx + r];
                }
            }
            if (a.stat_partials) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nvalid) { s1[t] += v[r]; s2[t] += v[r] * v[r]; }
            }
            if (a.sigmoid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + expf(-v[r]));
            }
            if (vec) {
                *reinterpret_cast<float4*>(a.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int r = 0; r < nvalid; ++r) a.y[idx + r] = v[r];
            }
        }
    }
    if (a.stat_partials) {
        // lanes sharing (lane & 15) hold the same output channel: fold the 4 row groups, then the 4 waves
        __syncthreads();   // all waves are done reading xs/ws
        float* red = smem;  // [4 waves][NT*16][2]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float u1 = s1[t], u2 = s2[t];
            u1 += __shfl_xor(u1, 16); u2 += __shfl_xor(u2, 16);
            u1 += __shfl_xor(u1, 32); u2 += __shfl_xor(u2, 32);
            if (lane < 16) {
                red[(wave * NT * 16 + t * 16 + lane) * 2 + 0] = u1;
                red[(wave * NT * 16 + t * 16 + lane) * 2 + 1] = u2;
            }
        }
        __syncthreads();
        if (tid < NT * 16) {
            const int co = co0 + tid;
            if (co < a.Cout) {
                float u1 = 0.f, u2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { u1 += red[(w * NT * 16 + tid) * 2]; u2 += red[(w * NT * 16 + tid) * 2 + 1]; }
                const int nblk = ntz * nty * ntx;
                const int t = (tz * nty + ty) * ntx + tx;
                float* p = a.stat_partials + (((size_t)n * a.Cout + co) * nblk + t) * 2;
                p[0] = u1; p[1] = u2;
            }
        }
    }
}

}  // namespace ru
