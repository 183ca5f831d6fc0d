// conv3_sb.hip -- 3x3x3 stride-1 pad-1 convolution (forward and data gradient) on the bf16 matrix cores with
// SPLIT operands: every fp32 value v is carried as hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits together) and
// each product is formed as  hi*hi + lo*hi + hi*lo  in fp32 accumulators -- 3 x v_mfma_f32_16x16x32_bf16 per K-step.
// SURVEY section 7 measured this scheme at max |dp| = 4.7e-5 on the whole network (plain bf16 operands: 3e-2, outside
// the 1e-3 bar), for 16/3 = 5.3x the f32-MFMA rate.  Same call sites as conv3_f32.hip (model.py:72-73,336,348).
//
//   HBM: activations stay NCDHW fp32 (coalesced float4 loads along W); nothing is stored in bf16.
//   GEMM: M = 16 consecutive x voxels per MFMA tile (TZ x TY x 16 voxel tile per workgroup, MT tiles per wave),
//         N = 16 output channels per workgroup (blockIdx.y), K = 2 taps x 16 input channels per K-step
//         (27 taps padded to 28 -> 14 K-steps per 16-channel chunk).
//   LDS image (built while staging: fused affine + LeakyReLU, split, pack, transpose): 16-byte packets of 8 bf16,
//         lds[hl][half][pos] with hl = hi/lo plane, half = channels 0-7 / 8-15, pos = halo-linear voxel.  The A
//         fragment of lane l (row = l&15, k-group g = l>>4) is ONE ds_read_b128 at
//         lds[hl][g&1][pos(tap(2*ks + (g>>1))) + row]: the 16 lanes of a k-group read 16 consecutive packets (256 B,
//         all 64 banks once); the four hardware lane groups of ds_read_b128 each cover 8 rows of one half-plane and the
//         other 8 rows of the second, and the half-planes are a multiple of 256 B apart -> conflict free for every tap.
//   B (weights): pre-packed on the device into per-lane fragment order (hi and lo), loaded once per chunk into
//         registers: 14 K-steps x 2 x 4 VGPRs.
//   C/D: lane holds 4 consecutive x voxels of output channel l&15 -> shared epilogue (conv3_epilogue.hpp).
#include "conv3_epilogue.hpp"
#include <stdlib.h>

namespace ru {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_KSTEPS = 14;       // ceil(27 taps / 2)

template <int TZ, int TY>
struct SB {
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = 18;
    static constexpr int HVOL = HZ * HY * HX;
    static constexpr int HVOLP = (HVOL + 15) / 16 * 16;          // packets per half-plane: multiple of 16 (256 B)
    static constexpr int LDS_BYTES = 4 * HVOLP * 16;              // [hi,lo] x [half0,half1]
    static constexpr int MT = TZ * TY / 4;
    static constexpr int NROW = HZ * HY;
    static constexpr int NSV = (NROW * 6 + 255) / 256;
    static_assert(MT >= 1 && TY % MT == 0, "a wave's M-tiles must lie in one z-slab");
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// v = hi + lo with hi = bf16_rne(v), lo = bf16_rne(v - hi); two values per v_cvt_pk_bf16_f32, the hi halves are
// re-expanded with one shift / one mask (3 VALU per value in total)
__device__ __forceinline__ void split8(const float (&t)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x2 h;
        h[0] = (__bf16)t[2 * i];
        h[1] = (__bf16)t[2 * i + 1];
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        const float h0 = __builtin_bit_cast(float, hb << 16);
        const float h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
        bf16x2 l;
        l[0] = (__bf16)(t[2 * i] - h0);
        l[1] = (__bf16)(t[2 * i + 1] - h1);
        hi[i] = hb;
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}

template <int TZ, int TY>
__global__ __launch_bounds__(256, 2) void conv3_sb_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx, int nchunk) {
    using P = SB<TZ, TY>;
    constexpr int MT = P::MT, HY = P::HY, HX = P::HX, HVOLP = P::HVOLP, NSV = P::NSV, NROW = P::NROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = b % ntx; b /= ntx;
    const int ty = b % nty; b /= nty;
    const int tz = b % ntz;
    const int n = b / ntz;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16;
    const int cog = blockIdx.y, co0 = cog * 16;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;

    // staging slots (W % 4 == 0): halo row [x0-1, x0+17) = six aligned 16-byte segments [x0-4+4q, +4); slot = (row, q)
    int gv[NSV], lp[NSV];
#pragma unroll
    for (int j = 0; j < NSV; ++j) {
        const int item = tid + j * 256;
        const int row = item / 6, q = item - row * 6;
        const int hz = row / HY, hy = row - hz * HY;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 - 4 + 4 * q;
        const bool slot = item < NROW * 6;
        const bool ok = slot && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && gx >= 0 && gx < W;
        gv[j] = !slot ? -2 : (ok ? (gz * H + gy) * W + gx : -1);
        lp[j] = row * HX + 4 * q - 3;           // packet index of element 0 (elements outside [0,18) of the row are skipped)
    }

    // A-fragment packet offsets per K-step (hi plane; lo plane = + 2*HVOLP)
    const int mz = (wave * MT) / TY, my0 = (wave * MT) % TY;
    const int kg = lane >> 4;
    int aoff[SB_KSTEPS];
#pragma unroll
    for (int ks = 0; ks < SB_KSTEPS; ++ks) {
        int tap = 2 * ks + (kg >> 1);
        if (tap > 26) tap = 26;                  // phantom 28th tap: its weights are zero, any valid address will do
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        aoff[ks] = (kg & 1) * HVOLP + ((mz + dz) * HY + my0 + dy) * HX + dx + (lane & 15);
    }

    f32x4 acc[MT][1];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;    // neutral constants make the fused transform branch-free

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        if (chunk) __syncthreads();
        // ---- weight fragments of (cog, chunk) straight into registers (L2-resident, lane-linear 16-byte loads)
        u32x4 wreg[SB_KSTEPS][2];
        {
            const u32x4* wp = wfrag + ((size_t)(cog * nchunk + chunk) * (SB_KSTEPS * 2)) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < SB_KSTEPS; ++ks) {
                wreg[ks][0] = wp[(ks * 2 + 0) * 64];
                wreg[ks][1] = wp[(ks * 2 + 1) * 64];
            }
        }
        // ---- stage 16 channels: per slot and channel-half, 8 float4 loads (8 channels x 4 voxels) in flight, then
        //      transform + split + transpose into 4 hi and 4 lo packets
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cb = chunk * 16 + half * 8;
            float sc[8], sh[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                sc[c] = 1.f; sh[c] = 0.f;
                if (xform && cb + c < a.Cin) { sc[c] = a.in_scale[n * a.Cin + cb + c]; sh[c] = a.in_shift[n * a.Cin + cb + c]; }
            }
            float4 v[NSV][8];
#pragma unroll
            for (int j = 0; j < NSV; ++j) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // UNCONDITIONAL load from a clamped (always valid) address; zeroed by select below.  A conditional load
                    // makes hipcc branch around it and wait vmcnt(0) per load: one load in flight per thread.
                    const int cg = cb + c < a.Cin ? cb + c : a.Cin - 1;
                    v[j][c] = *reinterpret_cast<const float4*>(a.x + ((size_t)n * a.Cin + cg) * DHW + (gv[j] > 0 ? gv[j] : 0));
                }
            }
#pragma unroll
            for (int j = 0; j < NSV; ++j) {
                if (gv[j] == -2) continue;
                const bool inb = gv[j] >= 0;
                const int q = (tid + j * 256) % 6;
                const int e0 = q == 0 ? 3 : 0, e1 = q == 5 ? 1 : 4;        // elements of this segment inside the halo row
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < e0 || e >= e1) continue;
                    float t[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float u = e == 0 ? v[j][c].x : (e == 1 ? v[j][c].y : (e == 2 ? v[j][c].z : v[j][c].w));
                        u = fmaf(u, sc[c], sh[c]);                          // branch-free: (1, 0, slope 1) when no transform
                        u = fmaxf(u, u * slope);                            // LeakyReLU for 0 < slope <= 1
                        t[c] = (inb && cb + c < a.Cin) ? u : 0.f;          // zero padding applies to the ACTIVATED tensor
                    }
                    u32x4 hi, lo;
                    split8(t, hi, lo);
                    lds[half * HVOLP + lp[j] + e] = hi;
                    lds[(2 + half) * HVOLP + lp[j] + e] = lo;
                }
            }
        }
        __syncthreads();
        // ---- 14 K-steps x MT M-tiles x 3 products
#pragma unroll
        for (int ks = 0; ks < SB_KSTEPS; ++ks) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[ks][0]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[ks][1]);
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, lds[aoff[ks] + i * HX]);
                al[i] = __builtin_bit_cast(bf16x8, lds[aoff[ks] + 2 * HVOLP + i * HX]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][0], 0, 0, 0);
        }
    }
    conv3_epilogue<MT, 1>(a, acc, smem, n, z0, y0, x0, mz, my0, co0, tz, ty, tx, ntz, nty, ntx);
}

// ------------------------------------------------------------------ v2: persistent producer / consumer workgroups
// 512 threads: waves 0-3 are CONSUMERS (A fragments from LDS, weights in registers, 3 MFMAs per K-step, epilogue),
// waves 4-7 are PRODUCERS (global float4 loads -> fused affine + LeakyReLU -> hi/lo split -> transposed LDS image).
// A workgroup walks a contiguous run of tiles (halo re-reads stay in its XCD's L2); the LDS image is double buffered:
// while the consumers are on item w the producers finish item w+1 in the other buffer and already have the global
// loads of item w+2 in flight.  One __syncthreads per item.  Each SIMD hosts one consumer and one producer wave, so
// the matrix pipe and the VALU/LDS-store work of the staging overlap instead of alternating.
// Per-tile GroupNorm statistics are written per consumer WAVE (no cross-wave reduction -> no extra barrier).
template <int MT>
__device__ __forceinline__ void sb2_epilogue(const Conv3Args& a, f32x4 (&acc)[MT], int n, int z0, int y0, int x0, int mz, int my0,
                                             int co0, int tile_in_sample, int nblk, int wave, int lane) {
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int zz = z0 + mz;
    const int xq = x0 + (lane >> 4) * 4;
    const int co = co0 + (lane & 15);
    float s1 = 0.f, s2 = 0.f;
    const float bv = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    float4 radd[MT];
    if (a.add) {       // residual: all loads first (unconditional, clamped), then use
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int yy = y0 + my0 + i;
            const bool ok = zz < D && yy < H && co < a.Cout && xq < W;
            const size_t idx = ok ? (((size_t)n * a.Cout + co) * D + zz) * HW + (size_t)yy * W + xq : 0;
            radd[i] = *reinterpret_cast<const float4*>(a.add + idx);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int yy = y0 + my0 + i;
        const bool ok = zz < D && yy < H && co < a.Cout && xq < W;       // W % 4 == 0: the 4 voxels are in or out together
        if (!ok) continue;
        const size_t idx = (((size_t)n * a.Cout + co) * D + zz) * HW + (size_t)yy * W + xq;
        f32x4 v = acc[i];
        v += bv;
        if (a.add) { v[0] += radd[i].x; v[1] += radd[i].y; v[2] += radd[i].z; v[3] += radd[i].w; }
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        if (a.sigmoid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + expf(-v[r]));
        }
        *reinterpret_cast<float4*>(a.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (a.stat_partials) {
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lane < 16 && co < a.Cout) {
            float* p = a.stat_partials + (((size_t)n * a.Cout + co) * nblk + tile_in_sample * 4 + wave) * 2;
            p[0] = s1; p[1] = s2;
        }
    }
}

template <int TZ, int TY>
__global__ __launch_bounds__(512, 2) void conv3_sb2_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx, int nchunk, int dbg) {
    // dbg (RU_SB2_DEBUG, ablation only; results are wrong when set): 1 = producers skip transform/split/LDS store,
    // 2 = producers skip global loads, 4 = consumers skip the MFMAs, 8 = consumers skip the epilogue
    using P = SB<TZ, TY>;
    constexpr int MT = P::MT, HY = P::HY, HX = P::HX, HVOLP = P::HVOLP, NROW = P::NROW;
    constexpr int BUF = 4 * HVOLP;                      // packets per LDS buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3;                             // wave index inside its role group
    const int ptid = tid & 255;
    const int cog = blockIdx.y, co0 = cog * 16;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    // tile of step k: k*G + swz(b).  Workgroup b runs on XCD b % 8 (observed, used for speed only): at every step the 256
    // resident workgroups cover 256 consecutive tiles and each XCD a compact run of G/8 of them, so the x/y halos of
    // neighbouring tiles are shared in that XCD's L2 while they are hot.
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int t_begin = swz;
    const int nsteps = swz < ntile ? (ntile - swz + G - 1) / G : 0;
    const int nitems = nsteps * nchunk;
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;    // neutral constants make the fused transform branch-free

    auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0, int& tis) {
        int b = tile;
        n = b / tiles_per_sample;
        tis = b - n * tiles_per_sample;
        b = tis;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty;
        const int tz = b / nty;
        z0 = tz * TZ; y0 = ty * TY; x0 = tx * 16;
    };

    if (producer) {
        // ---------------------------------------------------------------- producers
        // Work items of one (tile, 16-channel chunk):
        //   interior: halo row x 4 aligned float4 segments [x0+4q, +4)  -> NROW*4 items, every lane has 4 valid voxels
        //   edge    : halo row x {x0-1, x0+16}                           -> NROW*2 items, one scalar per channel
        // (the six-segment cover of the 18-wide row wasted 6 of 24 loaded floats and half of the lanes' VALU work)
        constexpr int NI = NROW * 4, NE = NROW * 2;
        static_assert(NI <= 256 && NE <= 128, "one interior and one edge item per producer thread");
        const bool has_i = ptid < NI;
        const int e_id = ptid - (256 - NE);                  // edge items live on the last NE producer threads
        const bool has_e = e_id >= 0;
        const int irow = ptid / 4, iq = ptid & 3;
        const int erow = has_e ? e_id >> 1 : 0, eside = e_id & 1;
        int gi = -1, ge = -1, n_cur = 0;                     // global offsets inside one channel volume (-1: zero fill)
        float4 vi[16];
        float ve[16];
        auto issue = [&](int item) {                          // issue every load of `item` (no waits)
            if (dbg & 2) return;
            const int tile = t_begin + (item / nchunk) * G, chunk = item % nchunk;
            int z0, y0, x0, tis;
            tile_origin(tile, n_cur, z0, y0, x0, tis);
            {
                const int hz = irow / HY, hy = irow - hz * HY;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + 4 * iq;
                const bool ok = has_i && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && gx < W;
                gi = ok ? (gz * H + gy) * W + gx : -1;
            }
            {
                const int hz = erow / HY, hy = erow - hz * HY;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = eside ? x0 + 16 : x0 - 1;
                const bool ok = has_e && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                ge = ok ? (gz * H + gy) * W + gx : -1;
            }
            // ablation (dbg 16 / 32): every lane reads element 0 of the channel -> same instruction stream, one cache line
            const int ge_eff = (dbg & 16) ? 0 : (ge > 0 ? ge : 0), gi_eff = (dbg & 32) ? 0 : (gi > 0 ? gi : 0);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int cc = chunk * 16 + c;
                const float* xp = a.x + ((size_t)n_cur * a.Cin + (cc < a.Cin ? cc : a.Cin - 1)) * DHW;   // unconditional, clamped
                vi[c] = *reinterpret_cast<const float4*>(xp + gi_eff);
                ve[c] = xp[ge_eff];
            }
        };
        auto store = [&](int item, u32x4* buf) {              // consume the in-flight loads: transform, split, transpose
            if (dbg & 1) {
                if (!(dbg & 2)) {
                    float acc0 = 0.f;
#pragma unroll
                    for (int c = 0; c < 16; ++c) acc0 += vi[c].x + ve[c];
                    if (acc0 == 12345.678f) buf[0] = u32x4{1u, 2u, 3u, 4u};
                }
                return;
            }
            const int chunk = item % nchunk;
            const float mi = gi >= 0 ? 1.f : 0.f, me = ge >= 0 ? 1.f : 0.f;   // zero padding applies to the ACTIVATED tensor
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cb = chunk * 16 + half * 8;
                float sci[8], shi[8], sce[8], she[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool cok = cb + c < a.Cin;
                    float sc = cok ? 1.f : 0.f, sh = 0.f;     // channels beyond Cin: (0, 0) -> exact zeros without a select
                    if (xform && cok) { sc = a.in_scale[n_cur * a.Cin + cb + c]; sh = a.in_shift[n_cur * a.Cin + cb + c]; }
                    sci[c] = sc * mi; shi[c] = sh * mi; sce[c] = sc * me; she[c] = sh * me;
                }
                if (has_i) {
                    const int lp = irow * HX + 4 * iq + 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float4 f = vi[half * 8 + c];
                            float u = e == 0 ? f.x : (e == 1 ? f.y : (e == 2 ? f.z : f.w));
                            u = fmaf(u, sci[c], shi[c]);
                            t[c] = fmaxf(u, u * slope);       // LeakyReLU for 0 < slope <= 1 (slope 1: identity)
                        }
                        u32x4 hi, lo;
                        split8(t, hi, lo);
                        buf[half * HVOLP + lp + e] = hi;
                        buf[(2 + half) * HVOLP + lp + e] = lo;
                    }
                }
                if (has_e) {
                    const int lp = erow * HX + (eside ? 17 : 0);
                    float t[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float u = fmaf(ve[half * 8 + c], sce[c], she[c]);
                        t[c] = fmaxf(u, u * slope);
                    }
                    u32x4 hi, lo;
                    split8(t, hi, lo);
                    buf[half * HVOLP + lp] = hi;
                    buf[(2 + half) * HVOLP + lp] = lo;
                }
            }
        };
        if (nitems > 0) {
            issue(0);
            store(0, lds);
            if (nitems > 1) issue(1);
        }
        __syncthreads();
        for (int w = 0; w < nitems; ++w) {
            if (w + 1 < nitems) {
                store(w + 1, lds + ((w + 1) & 1) * BUF);
                if (w + 2 < nitems) issue(w + 2);
            }
            __syncthreads();
        }
    } else {
        // ---------------------------------------------------------------- consumers
        const int mz = (rw * MT) / TY, my0 = (rw * MT) % TY;
        const int kg = lane >> 4;
        int aoff[SB_KSTEPS];
#pragma unroll
        for (int ks = 0; ks < SB_KSTEPS; ++ks) {
            int tap = 2 * ks + (kg >> 1);
            if (tap > 26) tap = 26;
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            aoff[ks] = (kg & 1) * HVOLP + ((mz + dz) * HY + my0 + dy) * HX + dx + (lane & 15);
        }
        u32x4 wreg[SB_KSTEPS][2];
        auto load_w = [&](int chunk) {
            const u32x4* wp = wfrag + ((size_t)(cog * nchunk + chunk) * (SB_KSTEPS * 2)) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < SB_KSTEPS; ++ks) {
                wreg[ks][0] = wp[(ks * 2 + 0) * 64];
                wreg[ks][1] = wp[(ks * 2 + 1) * 64];
            }
        };
        if (nchunk == 1) load_w(0);                     // one chunk: the weights stay in registers for the whole run of tiles
        f32x4 acc[MT];
        __syncthreads();                                // item 0 is staged
        for (int w = 0; w < nitems; ++w) {
            const int chunk = w % nchunk;
            if (chunk == 0) {
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (nchunk > 1) load_w(chunk);
            const u32x4* buf = lds + (w & 1) * BUF;
            if (!(dbg & 4)) {
                // software pipeline over two register sets (M-tiles 0..MT/2-1 and MT/2..MT-1): the A fragments of the next
                // K-step are read while the other set's MFMAs run (one wave per SIMD cannot hide LDS latency by itself)
                constexpr int HM = MT / 2;
                bf16x8 ah0[HM], al0[HM], ah1[HM], al1[HM];
#pragma unroll
                for (int i = 0; i < HM; ++i) {
                    ah0[i] = __builtin_bit_cast(bf16x8, buf[aoff[0] + i * HX]);
                    al0[i] = __builtin_bit_cast(bf16x8, buf[aoff[0] + 2 * HVOLP + i * HX]);
                }
#pragma unroll
                for (int i = 0; i < HM; ++i) {
                    ah1[i] = __builtin_bit_cast(bf16x8, buf[aoff[0] + (HM + i) * HX]);
                    al1[i] = __builtin_bit_cast(bf16x8, buf[aoff[0] + 2 * HVOLP + (HM + i) * HX]);
                }
#pragma unroll
                for (int ks = 0; ks < SB_KSTEPS; ++ks) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[ks][0]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[ks][1]);
#pragma unroll
                    for (int i = 0; i < HM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0[i], bh, acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < HM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0[i], bl, acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < HM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0[i], bh, acc[i], 0, 0, 0);
                    if (ks + 1 < SB_KSTEPS) {
#pragma unroll
                        for (int i = 0; i < HM; ++i) {
                            ah0[i] = __builtin_bit_cast(bf16x8, buf[aoff[ks + 1] + i * HX]);
                            al0[i] = __builtin_bit_cast(bf16x8, buf[aoff[ks + 1] + 2 * HVOLP + i * HX]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < HM; ++i) acc[HM + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1[i], bh, acc[HM + i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < HM; ++i) acc[HM + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1[i], bl, acc[HM + i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < HM; ++i) acc[HM + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1[i], bh, acc[HM + i], 0, 0, 0);
                    if (ks + 1 < SB_KSTEPS) {
#pragma unroll
                        for (int i = 0; i < HM; ++i) {
                            ah1[i] = __builtin_bit_cast(bf16x8, buf[aoff[ks + 1] + (HM + i) * HX]);
                            al1[i] = __builtin_bit_cast(bf16x8, buf[aoff[ks + 1] + 2 * HVOLP + (HM + i) * HX]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (chunk == nchunk - 1 && !(dbg & 8)) {
                int n, z0, y0, x0, tis;
                tile_origin(t_begin + (w / nchunk) * G, n, z0, y0, x0, tis);
                sb2_epilogue<MT>(a, acc, n, z0, y0, x0, mz, my0, co0, tis, tiles_per_sample * 4, rw, lane);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ weight fragments
// unit u = ((cog*nchunk + chunk)*14 + ks)*2 + hl, 64 lanes x 16 bytes each: lane l (col = l&15, k-group g = l>>4) holds,
// for e = 0..7, W[cout = cog*16 + col][cin = chunk*16 + (g&1)*8 + e][tap = 2*ks + (g>>1)].
__global__ void conv3_sb_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ wfrag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = ncog * nchunk * SB_KSTEPS * 64;
    if (i >= total) return;
    const int lane = i & 63;
    const int ks = (i >> 6) % SB_KSTEPS;
    const int chunk = ((i >> 6) / SB_KSTEPS) % nchunk;
    const int cog = (i >> 6) / (SB_KSTEPS * nchunk);
    const int col = lane & 15, g = lane >> 4;
    const int tap = 2 * ks + (g >> 1);
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int co = cog * 16 + col;
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = chunk * 16 + (g & 1) * 8 + e;
        float v = 0.f;
        if (tap < 27 && ci < cin_conv && co < cout_conv)
            v = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
        t[e] = v;
    }
    u32x4 hi, lo;
    split8(t, hi, lo);
    const size_t unit = ((size_t)(cog * nchunk + chunk) * SB_KSTEPS + ks) * 2;
    wfrag[(unit + 0) * 64 + lane] = hi;
    wfrag[(unit + 1) * 64 + lane] = lo;
}

size_t conv3_sb_frag_bytes(int Cin_conv, int Cout_conv) {
    return (size_t)cdiv(Cout_conv, 16) * cdiv(Cin_conv, 16) * SB_KSTEPS * 2 * 64 * 16;
}

int conv3_sb_pack_weights(const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, hipStream_t s) {
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int nchunk = cdiv(cin_conv, 16), ncog = cdiv(cout_conv, 16);
    const int total = ncog * nchunk * SB_KSTEPS * 64;
    hipLaunchKernelGGL(conv3_sb_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, (u32x4*)wfrag, Cin_f, Cout_f, mode, nchunk, ncog);
    RU_CHECK_LAUNCH("conv3_sb_pack_kernel");
    return RU_OK;
}

struct SBChoice { int tz, ty; };
static SBChoice sb_choose(int N, int Cout, int D, int H, int W) {
    const int ncog = cdiv(Cout, 16);
    auto blocks = [&](int tz, int ty) { return (long)N * cdiv(D, tz) * cdiv(H, ty) * cdiv(W, 16) * ncog; };
    if (blocks(4, 8) >= 1024) return {4, 8};
    if (blocks(2, 8) >= 1024) return {2, 8};
    return {2, 4};
}

// v2 (persistent producer/consumer) handles the large-tile case; it writes one statistics partial per consumer wave
static bool sb_use_v2(const SBChoice& c) { return c.tz == 4 && c.ty == 8; }

int conv3_sb_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W) {
    (void)Cin;
    const SBChoice c = sb_choose(N, Cout, D, H, W);
    return cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, 16) * (sb_use_v2(c) ? 4 : 1);
}

template <int TZ, int TY>
static int sb2_cfg(const Conv3Args& a, hipStream_t s) {
    using P = SB<TZ, TY>;
    static bool attr_done = false;
    static int ncu = 256;
    constexpr int LDS2 = 2 * P::LDS_BYTES;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_sb2_kernel<TZ, TY>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_sb2)");
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        attr_done = true;
    }
    const int ntz = cdiv(a.D, TZ), nty = cdiv(a.H, TY), ntx = cdiv(a.W, 16);
    const long ntile = (long)a.N * ntz * nty * ntx;
    const int ncog = cdiv(a.Cout, 16);
    long gx = ncu / (ncog < ncu ? ncog : ncu);          // one resident workgroup per CU in total
    if (gx < 1) gx = 1;
    if (gx > ntile) gx = ntile;
    dim3 grid((unsigned)gx, (unsigned)ncog);
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("RU_SB2_DEBUG"); dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL((conv3_sb2_kernel<TZ, TY>), grid, dim3(512), LDS2, s, a, (const u32x4*)a.wfrag, ntz, nty, ntx, cdiv(a.Cin, 16), dbg);
    RU_CHECK_LAUNCH("conv3_sb2_kernel");
    return RU_OK;
}

template <int TZ, int TY>
static int sb_cfg(const Conv3Args& a, hipStream_t s) {
    using P = SB<TZ, TY>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_sb_kernel<TZ, TY>), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_sb)");
        attr_done = true;
    }
    const int ntz = cdiv(a.D, TZ), nty = cdiv(a.H, TY), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)((long)a.N * ntz * nty * ntx), (unsigned)cdiv(a.Cout, 16));
    hipLaunchKernelGGL((conv3_sb_kernel<TZ, TY>), grid, dim3(256), P::LDS_BYTES, s, a, (const u32x4*)a.wfrag, ntz, nty, ntx, cdiv(a.Cin, 16));
    RU_CHECK_LAUNCH("conv3_sb_kernel");
    return RU_OK;
}

int conv3_sb_launch(const Conv3Args& a, hipStream_t s) {
    RU_REQUIRE((a.W & 3) == 0, "conv3_sb: W must be a multiple of 4");
    const SBChoice c = sb_choose(a.N, a.Cout, a.D, a.H, a.W);
    if (sb_use_v2(c)) return sb2_cfg<4, 8>(a, s);
    if (c.ty == 8) return sb_cfg<2, 8>(a, s);
    return sb_cfg<2, 4>(a, s);
}

}  // namespace ru
