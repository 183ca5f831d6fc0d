// Device helpers shared by pointwise.hip (NCDHW) and pointwise_c16.hip (voxel-major working layout).
#pragma once
#include "ru_common.h"

namespace ru {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over a 256-thread block; result valid in thread 0 (and broadcast through `buf[0]`)
__device__ __forceinline__ float block_sum(float v, float* buf) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    return buf[0] + buf[1] + buf[2] + buf[3];
}
__device__ __forceinline__ double block_sum_d(double v, double* buf) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    return buf[0] + buf[1] + buf[2] + buf[3];
}
__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

static inline unsigned grid1d(size_t n, int per_block, unsigned cap = 1u << 20) {
    size_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

// 8 floats -> 8 bf16 hi (bf16_rne(v)) and 8 bf16 lo (bf16_rne(v - hi)), 16 bytes each (the split-bf16 operand format)
typedef __bf16 pw_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int pw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pw_split8(const float (&t)[8], pw_u32x4& hi, pw_u32x4& lo) {
    split_n<4>(t, hi, lo);
}

// ------------------------------------------------------------------ trilinear x2 index helpers (model.py:12-14)
// source index of output o: src = max(o/2 - 0.25, 0); i0 = floor(src); l1 = src - i0; i1 = i0 + (i0 < n-1)
__device__ __forceinline__ void up2_src(int o, int n, int& i0, int& i1, float& l0, float& l1) {
    float src = 0.5f * (float)o - 0.25f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
}

// transpose of the above in gather form: dx[k] collects from outputs 2k-1 .. 2k+2 on each axis
__device__ __forceinline__ float up2_coef(int o, int n, int k) {
    int i0, i1; float l0, l1;
    up2_src(o, n, i0, i1, l0, l1);
    return (i0 == k ? l0 : 0.f) + (i1 == k ? l1 : 0.f);
}

}  // namespace ru
