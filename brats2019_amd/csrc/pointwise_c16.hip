// pointwise_c16.hip -- the streaming kernels of the ResUNet path on the voxel-major working layout "C16":
// activations [N][C/16][D][H][W][16] (16 channels of a voxel contiguous, C % 16 == 0).  The split-bf16 engine keeps every
// tensor between the first and the last convolution in this layout (include/resunet_hip.h); these kernels are the C16
// forms of pointwise.hip: GroupNorm apply / backward, trilinear x2 and its transpose, the 1x1x1 / 2x2x2-stride-2
// convolution (exact-f32 MFMA) -- same arithmetic per element, different addressing.
//
// Addressing: one (sample, channel block) = V voxels x 16 floats = 4V float4.  float4 index f -> voxel f>>2, channels
// cb*16 + 4*(f&3) .. +3.  Grid strides are multiples of 4, so a thread keeps its channel quad and loads the per-channel
// parameters once.
#include "pw_helpers.hpp"
#include "fin_tail.hpp"
#include <type_traits>
#include <stdlib.h>

namespace ru {

typedef float f32x4_c16 __attribute__((ext_vector_type(4)));
template <int I, int N, class F> __device__ __forceinline__ void static_for_c1_impl(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_c1_impl<I + 1, N>(f); }
}
template <int N, class F> __device__ __forceinline__ void static_for_c1(F&& f) { static_for_c1_impl<0, N>(f); }

static inline dim3 c16_grid(size_t V, int blocks_nc, unsigned cap = 2048) {
    size_t bx = (V * 4 + 255) / 256;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)blocks_nc);
}

// ------------------------------------------------------------------ GroupNorm apply: y = (res) + lrelu(x*scale[n,c] + shift[n,c])
// RES: 0 none, 1 plain residual, 2 residual = lrelu(res*rscale[n,c] + rshift[n,c]) (a GroupNorm output that was never written)
template <int RES, bool NT = false>
__global__ __launch_bounds__(256) void gn_apply16_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ res, float* __restrict__ y, int C, size_t V, float slope,
                                                         const float* __restrict__ rscale, const float* __restrict__ rshift, float rslope) {
    const int nb = blockIdx.y, CB = C >> 4;
    const int n = nb / CB, cb = nb - n * CB;
    const int q = threadIdx.x & 3;
    const size_t pc = (size_t)n * C + cb * 16 + 4 * q;
    const float4 a = *reinterpret_cast<const float4*>(scale + pc);
    const float4 b = *reinterpret_cast<const float4*>(shift + pc);
    float4 ra = make_float4(1.f, 1.f, 1.f, 1.f), rb = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (RES == 2) { ra = *reinterpret_cast<const float4*>(rscale + pc); rb = *reinterpret_cast<const float4*>(rshift + pc); }
    const size_t base = (size_t)nb * V * 4, F = V * 4;
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    const float4* rp = reinterpret_cast<const float4*>(res) + base;
    float4* yp = reinterpret_cast<float4*>(y) + base;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < F; f += (size_t)gridDim.x * 256) {
        float4 t;
        if constexpr (NT) { const f32x4_c16 tv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_c16*>(xp + f)); t = make_float4(tv[0], tv[1], tv[2], tv[3]); }
        else t = xp[f];
        float4 o;
        o.x = lrelu(t.x * a.x + b.x, slope); o.y = lrelu(t.y * a.y + b.y, slope);
        o.z = lrelu(t.z * a.z + b.z, slope); o.w = lrelu(t.w * a.w + b.w, slope);
        if constexpr (RES != 0) {
            float4 r = rp[f];
            if constexpr (RES == 2) {
                r.x = lrelu(r.x * ra.x + rb.x, rslope); r.y = lrelu(r.y * ra.y + rb.y, rslope);
                r.z = lrelu(r.z * ra.z + rb.z, rslope); r.w = lrelu(r.w * ra.w + rb.w, rslope);
            }
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if constexpr (NT) __builtin_nontemporal_store(f32x4_c16{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_c16*>(yp + f));
        else yp[f] = o;
    }
}
int gn_apply16_launch(const float* x, const float* scale, const float* shift, const float* res, float* y, int N, int C, size_t V, float slope, hipStream_t s,
                      const float* rscale, const float* rshift, float rslope) {
    RU_REQUIRE(C % 16 == 0, "gn_apply16: C must be a multiple of 16");
    RU_REQUIRE(!rscale || (res && rshift), "gn_apply16: a residual transform needs the residual and both of its vectors");
    const dim3 grid = c16_grid(V, N * (C / 16));
    // nontemporal stores for tensors that outlast the caches (>= 128 MB: the 16-channel level); smaller ones are read again from L2 / MALL
    const bool nt = (size_t)N * C * V * sizeof(float) >= ((size_t)128 << 20);
    if (!res) hipLaunchKernelGGL(gn_apply16_kernel<0>, grid, dim3(256), 0, s, x, scale, shift, res, y, C, V, slope, rscale, rshift, rslope);
    else if (!rscale && nt) hipLaunchKernelGGL((gn_apply16_kernel<1, true>), grid, dim3(256), 0, s, x, scale, shift, res, y, C, V, slope, rscale, rshift, rslope);
    else if (!rscale) hipLaunchKernelGGL(gn_apply16_kernel<1>, grid, dim3(256), 0, s, x, scale, shift, res, y, C, V, slope, rscale, rshift, rslope);
    else if (nt) hipLaunchKernelGGL((gn_apply16_kernel<2, true>), grid, dim3(256), 0, s, x, scale, shift, res, y, C, V, slope, rscale, rshift, rslope);
    else hipLaunchKernelGGL(gn_apply16_kernel<2>, grid, dim3(256), 0, s, x, scale, shift, res, y, C, V, slope, rscale, rshift, rslope);
    RU_CHECK_LAUNCH("gn_apply16_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ GroupNorm backward
// reduce: per (n, c) and voxel chunk S1 = sum dyh, S2 = sum dyh * xhat (dyh = dy * lrelu'(x*scale+shift)); partials
// [N][C][nblk][2] exactly like the NCDHW kernel, so gn_bwd_finalize is shared.  Fixed reduction order: deterministic.
// voxels of one (n, channel block) reduced by one workgroup: large volumes take 8192 so the whole grid is resident at once,
// 16^3 volumes 512 (2048 left 64 workgroups for 256 CUs)
static inline int gn16_chunk(size_t V) { return V >= ((size_t)1 << 20) ? 8192 : (V >= ((size_t)1 << 15) ? 2048 : 512); }   // small volumes: enough workgroups to fill the chip
int gn_bwd_tiles16(size_t V) { const int ch = gn16_chunk(V); return (int)((V + ch - 1) / ch); }

__global__ __launch_bounds__(256) void gn_bwd_reduce16_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              float slope, float* __restrict__ partials, int C, size_t V, int G, int nblk, int chunk,
                                                              const FinTail fin) {
    extern __shared__ __attribute__((aligned(16))) float red_dyn[];      // [4][16][2] floats, then the tail's scratch
    float (*red)[16][2] = reinterpret_cast<float (*)[16][2]>(red_dyn);
    const int nb = blockIdx.y, CB = C >> 4;
    const int n = nb / CB, cb = nb - n * CB;
    const int q = threadIdx.x & 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = cb * 16 + 4 * q, cpg = C / G;
    const float4 a4 = *reinterpret_cast<const float4*>(scale + (size_t)n * C + c4);
    const float4 b4 = *reinterpret_cast<const float4*>(shift + (size_t)n * C + c4);
    const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    float mu[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int g = (c4 + k) / cpg; mu[k] = mean[n * G + g]; rs[k] = rstd[n * G + g]; }
    const size_t base = (size_t)nb * V * 4;
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    const float4* dp = reinterpret_cast<const float4*>(dy) + base;
    const size_t v0 = (size_t)blockIdx.x * chunk;
    const size_t v1 = v0 + chunk < V ? v0 + chunk : V;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (size_t f = v0 * 4 + threadIdx.x; f < v1 * 4; f += 256) {
        const float4 t = xp[f], d = dp[f];
        const float tx[4] = {t.x, t.y, t.z, t.w}, dx[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dh = (tx[k] * a[k] + b[k]) > 0.f ? dx[k] : dx[k] * slope;
            s1[k] += dh;
            s2[k] += dh * ((tx[k] - mu[k]) * rs[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { s1[k] += __shfl_xor(s1[k], o); s2[k] += __shfl_xor(s2[k], o); }
    }
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[wave][lane * 4 + k][0] = s1[k]; red[wave][lane * 4 + k][1] = s2[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        float* p = partials + (((size_t)n * C + cb * 16 + c) * nblk + blockIdx.x) * 2;
        stat_publish(p, (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]), (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]));
    }
    fin_tail(fin, partials, red_dyn + 128);              // RU_FUSE_TAIL_FINALIZE: the last workgroup writes coef / dgamma / dbeta
}
int gn_bwd_reduce16_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd,
                           float slope, float* partials, int N, int C, size_t V, int G, hipStream_t s, const FinTail* fin) {
    RU_REQUIRE(C % 16 == 0 && C % G == 0, "gn_bwd_reduce16: bad channel count");
    const int nblk = gn_bwd_tiles16(V);
    FinTail f{};
    if (fin) f = *fin;
    RU_REQUIRE(!f.ticket || (f.kind == 2 && f.nblk == nblk && f.N == N && f.C == C && f.G == G), "gn_bwd_reduce16: tail descriptor does not match the launch");
    const size_t lds = 512 + fin_tail_lds_bytes(f);
    RU_REQUIRE(lds <= 64 * 1024, "gn_bwd_reduce16: batch x channels too large for the in-launch finalize");
    hipLaunchKernelGGL(gn_bwd_reduce16_kernel, dim3(nblk, N * (C / 16)), dim3(256), lds, s, x, dy, scale, shift, mean, rstd, slope, partials, C, V, G, nblk, gn16_chunk(V), f);
    RU_CHECK_LAUNCH("gn_bwd_reduce16_kernel");
    return RU_OK;
}

// apply: dx = cA*dyh + cB*x + cC with coef[N][C][3]
__global__ __launch_bounds__(256) void gn_bwd_apply16_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ coef, float slope,
                                                             float* __restrict__ dx, int C, size_t V) {
    const int nb = blockIdx.y, CB = C >> 4;
    const int n = nb / CB, cb = nb - n * CB;
    const int q = threadIdx.x & 3;
    const size_t row = (size_t)n * C + cb * 16 + 4 * q;
    float a[4], b[4], cA[4], cB[4], cC[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = scale[row + k]; b[k] = shift[row + k];
        cA[k] = coef[(row + k) * 3]; cB[k] = coef[(row + k) * 3 + 1]; cC[k] = coef[(row + k) * 3 + 2];
    }
    const size_t base = (size_t)nb * V * 4, F = V * 4;
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    const float4* dp = reinterpret_cast<const float4*>(dy) + base;
    float4* op = reinterpret_cast<float4*>(dx) + base;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < F; f += (size_t)gridDim.x * 256) {
        const float4 t = xp[f], d = dp[f];
        float4 o;
        o.x = cA[0] * ((t.x * a[0] + b[0]) > 0.f ? d.x : d.x * slope) + (cB[0] * t.x + cC[0]);
        o.y = cA[1] * ((t.y * a[1] + b[1]) > 0.f ? d.y : d.y * slope) + (cB[1] * t.y + cC[1]);
        o.z = cA[2] * ((t.z * a[2] + b[2]) > 0.f ? d.z : d.z * slope) + (cB[2] * t.z + cC[2]);
        o.w = cA[3] * ((t.w * a[3] + b[3]) > 0.f ? d.w : d.w * slope) + (cB[3] * t.w + cC[3]);
        op[f] = o;
    }
}
// the same, output in split form: a thread owns 8 channels (one half) of a voxel: hi packet at voxel*64 + half*16, lo at + 32
__global__ __launch_bounds__(256) void gn_bwd_apply16_split_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ coef, float slope,
                                                                   float* __restrict__ dx, int C, size_t V) {
    const int nb = blockIdx.y, CB = C >> 4;
    const int n = nb / CB, cb = nb - n * CB;
    const int hf = threadIdx.x & 1;
    const size_t row = (size_t)n * C + cb * 16 + 8 * hf;
    float a[8], b[8], cA[8], cB[8], cC[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = scale[row + k]; b[k] = shift[row + k];
        cA[k] = coef[(row + k) * 3]; cB[k] = coef[(row + k) * 3 + 1]; cC[k] = coef[(row + k) * 3 + 2];
    }
    const size_t base = (size_t)nb * V * 4, F = V * 2;           // float4 units; F = (voxel, half) pairs
    const float4* xp = reinterpret_cast<const float4*>(x) + base;
    const float4* dp = reinterpret_cast<const float4*>(dy) + base;
    pw_u32x4* op = reinterpret_cast<pw_u32x4*>(dx) + base;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < F; f += (size_t)gridDim.x * 256) {
        const size_t v = f >> 1;
        const float4 t0 = xp[v * 4 + 2 * hf], t1 = xp[v * 4 + 2 * hf + 1], d0 = dp[v * 4 + 2 * hf], d1 = dp[v * 4 + 2 * hf + 1];
        const float tx[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w}, dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = cA[k] * ((tx[k] * a[k] + b[k]) > 0.f ? dd[k] : dd[k] * slope) + (cB[k] * tx[k] + cC[k]);
        pw_u32x4 hi, lo;
        pw_split8(o, hi, lo);
        op[v * 4 + hf] = hi;
        op[v * 4 + 2 + hf] = lo;
    }
}
int gn_bwd_apply16_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* coef, float slope, float* dx,
                          int N, int C, size_t V, int split, hipStream_t s) {
    RU_REQUIRE(C % 16 == 0, "gn_bwd_apply16: C must be a multiple of 16");
    if (split) {
        size_t bx = (V * 2 + 255) / 256;
        if (bx > 2048) bx = 2048;
        hipLaunchKernelGGL(gn_bwd_apply16_split_kernel, dim3((unsigned)bx, (unsigned)(N * (C / 16))), dim3(256), 0, s, x, dy, scale, shift, coef, slope, dx, C, V);
    } else {
        hipLaunchKernelGGL(gn_bwd_apply16_kernel, c16_grid(V, N * (C / 16)), dim3(256), 0, s, x, dy, scale, shift, coef, slope, dx, C, V);
    }
    RU_CHECK_LAUNCH("gn_bwd_apply16_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ trilinear x2 (model.py:12-14), same nesting z(y(x)) as the NCDHW kernel
// A thread owns one output x (and channel quad) of one source cell pair (kz, ky), kz in [-1, D-1]: the outputs zo = 2kz+1, 2kz+2 and
// yo = 2ky+1, 2ky+2 interpolate between the same source planes / rows (clamped at the borders, where up2_src gives the second
// corner weight 0), so 8 loads serve 4 outputs (the one-output-per-thread form issued 8 loads per output and ran at half the
// write bandwidth).  Lanes run over (quad, xo): every store instruction of a wave covers 1 KB of one output row.
__global__ __launch_bounds__(256) void up2_fwd16_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, float slope) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const size_t nb = blockIdx.y;
    const size_t Vi = (size_t)D * H * W, Vo = Vi * 8;
    const float4* xp = reinterpret_cast<const float4*>(x) + nb * Vi * 4;
    float4* yp = reinterpret_cast<float4*>(y) + nb * Vo * 4;
    const bool nt_out = (size_t)gridDim.y * Vo * 16 * sizeof(float) >= ((size_t)128 << 20);      // the output outlasts the caches (conv1_16_kernel's rule)
    const size_t total = (size_t)(D + 1) * (H + 1) * Wo * 4;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (size_t)gridDim.x * 256) {
        const int q = (int)(f & 3);
        size_t r = f >> 2;
        const int xo = (int)(r % Wo); r /= Wo;
        const int ky = (int)(r % (H + 1)) - 1;
        const int kz = (int)(r / (H + 1)) - 1;
        int x0, x1; float lx0, lx1;
        up2_src(xo, W, x0, x1, lx0, lx1);
        const int zA = kz < 0 ? 0 : kz, zB = kz + 1 > D - 1 ? D - 1 : kz + 1;
        const int yA = ky < 0 ? 0 : ky, yB = ky + 1 > H - 1 ? H - 1 : ky + 1;
        const size_t rAA = ((size_t)zA * H + yA) * W, rAB = ((size_t)zA * H + yB) * W, rBA = ((size_t)zB * H + yA) * W, rBB = ((size_t)zB * H + yB) * W;
        const float4 a00 = xp[(rAA + x0) * 4 + q], b00 = xp[(rAA + x1) * 4 + q];
        const float4 a01 = xp[(rAB + x0) * 4 + q], b01 = xp[(rAB + x1) * 4 + q];
        const float4 a10 = xp[(rBA + x0) * 4 + q], b10 = xp[(rBA + x1) * 4 + q];
        const float4 a11 = xp[(rBB + x0) * 4 + q], b11 = xp[(rBB + x1) * 4 + q];
        float4 e00, e01, e10, e11;                        // x-interpolated corners [z corner][y corner]
#define RU_UPX(c) e00.c = lx0 * a00.c + lx1 * b00.c; e01.c = lx0 * a01.c + lx1 * b01.c; e10.c = lx0 * a10.c + lx1 * b10.c; e11.c = lx0 * a11.c + lx1 * b11.c
        RU_UPX(x); RU_UPX(y); RU_UPX(z); RU_UPX(w);
#undef RU_UPX
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int zo = 2 * kz + 1 + dz;
            if (zo < 0 || zo >= Do) continue;
            int z0, z1; float lz0, lz1;
            up2_src(zo, D, z0, z1, lz0, lz1);             // z0 == zA; z1 == zB wherever lz1 != 0
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int yo = 2 * ky + 1 + dy;
                if (yo < 0 || yo >= Ho) continue;
                int y0, y1; float ly0, ly1;
                up2_src(yo, H, y0, y1, ly0, ly1);
                float4 o;
#define RU_UP2(c) o.c = lrelu(lz0 * (ly0 * e00.c + ly1 * e01.c) + lz1 * (ly0 * e10.c + ly1 * e11.c), slope)
                RU_UP2(x); RU_UP2(y); RU_UP2(z); RU_UP2(w);   // slope 1: no activation
#undef RU_UP2
                if (nt_out) __builtin_nontemporal_store(f32x4_c16{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_c16*>(yp + (((size_t)zo * Ho + yo) * Wo + xo) * 4 + q));
                else yp[(((size_t)zo * Ho + yo) * Wo + xo) * 4 + q] = o;
            }
        }
    }
}
int up2_fwd16_launch(const float* x, float* y, int N, int C, int D, int H, int W, float slope, hipStream_t s) {
    RU_REQUIRE(C % 16 == 0, "up2_fwd16: C must be a multiple of 16");
    const size_t total = (size_t)(D + 1) * (H + 1) * (2 * W) * 4;
    size_t bx = (total + 255) / 256;
    if (bx > 16384) bx = 16384;
    hipLaunchKernelGGL(up2_fwd16_kernel, dim3((unsigned)bx, (unsigned)(N * (C / 16))), dim3(256), 0, s, x, y, D, H, W, slope);
    RU_CHECK_LAUNCH("up2_fwd16_kernel");
    return RU_OK;
}

// transpose in gather form: coarse voxel k collects from fine 2k-1 .. 2k+2 on each axis.  A thread owns a coarse (ky, kx) column
// (and channel quad) and marches through a chunk of UP2B_ZC coarse planes: the x/y-reduced value of fine plane z feeds the two
// coarse planes it belongs to, so a coarse output costs 2 fine planes x 16 loads instead of 4 x 16.
#ifndef RU_UP2B_ZC
#define RU_UP2B_ZC 8
#endif
constexpr int UP2B_ZC = RU_UP2B_ZC;
__global__ __launch_bounds__(256) void up2_bwd16_kernel(const float* __restrict__ dy, float* __restrict__ dx, int D, int H, int W) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const size_t nb = blockIdx.y;
    const size_t Vi = (size_t)D * H * W, Vo = Vi * 8;
    const float4* dp = reinterpret_cast<const float4*>(dy) + nb * Vo * 4;
    float4* op = reinterpret_cast<float4*>(dx) + nb * Vi * 4;
    const int nch = (D + UP2B_ZC - 1) / UP2B_ZC;
    const size_t total = (size_t)nch * H * W * 4;
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (size_t)gridDim.x * 256) {
        const int q = (int)(f & 3);
        size_t r = f >> 2;
        const int kx = (int)(r % W); r /= W;
        const int ky = (int)(r % H);
        const int k0 = (int)(r / H) * UP2B_ZC;
        const int k1 = k0 + UP2B_ZC < D ? k0 + UP2B_ZC : D;
        float wy[4], wx[4];
        size_t oy[4], ox[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int y = 2 * ky - 1 + t, xx = 2 * kx - 1 + t;
            wy[t] = (y >= 0 && y < Ho) ? up2_coef(y, H, ky) : 0.f;
            wx[t] = (xx >= 0 && xx < Wo) ? up2_coef(xx, W, kx) : 0.f;
            oy[t] = (size_t)(y < 0 ? 0 : (y >= Ho ? Ho - 1 : y)) * Wo;           // clamped: the weight is 0 outside
            ox[t] = (size_t)(xx < 0 ? 0 : (xx >= Wo ? Wo - 1 : xx)) * 4 + q;
        }
        float4 accP = make_float4(0.f, 0.f, 0.f, 0.f), accC = accP;             // coarse planes m-1 and m
        for (int m = k0; m <= k1; ++m) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int z = 2 * m - 1 + h;
                const bool in = z >= 0 && z < Do;
                const size_t zrow = (size_t)(z < 0 ? 0 : (z >= Do ? Do - 1 : z)) * Ho * Wo;
                float4 pz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4* row = dp + (zrow + oy[b]) * 4;
                    const float4 p0 = row[ox[0]], p1 = row[ox[1]], p2 = row[ox[2]], p3 = row[ox[3]];
                    pz.x += wy[b] * (wx[0] * p0.x + wx[1] * p1.x + wx[2] * p2.x + wx[3] * p3.x);
                    pz.y += wy[b] * (wx[0] * p0.y + wx[1] * p1.y + wx[2] * p2.y + wx[3] * p3.y);
                    pz.z += wy[b] * (wx[0] * p0.z + wx[1] * p1.z + wx[2] * p2.z + wx[3] * p3.z);
                    pz.w += wy[b] * (wx[0] * p0.w + wx[1] * p1.w + wx[2] * p2.w + wx[3] * p3.w);
                }
                const float cP = (in && m - 1 >= k0) ? up2_coef(z, D, m - 1) : 0.f;
                const float cC = (in && m < k1) ? up2_coef(z, D, m) : 0.f;
                accP.x += cP * pz.x; accP.y += cP * pz.y; accP.z += cP * pz.z; accP.w += cP * pz.w;
                accC.x += cC * pz.x; accC.y += cC * pz.y; accC.z += cC * pz.z; accC.w += cC * pz.w;
            }
            if (m - 1 >= k0) op[(((size_t)(m - 1) * H + ky) * W + kx) * 4 + q] = accP;
            accP = accC;
            accC = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
int up2_bwd16_launch(const float* dy, float* dx, int N, int C, int D, int H, int W, hipStream_t s) {
    RU_REQUIRE(C % 16 == 0, "up2_bwd16: C must be a multiple of 16");
    const size_t total = (size_t)((D + UP2B_ZC - 1) / UP2B_ZC) * H * W * 4;
    size_t bx = (total + 255) / 256;
    if (bx > 16384) bx = 16384;
    hipLaunchKernelGGL(up2_bwd16_kernel, dim3((unsigned)bx, (unsigned)(N * (C / 16))), dim3(256), 0, s, dy, dx, D, H, W);
    RU_CHECK_LAUNCH("up2_bwd16_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ 2x2x2 stride-2 conv (model.py:361-363): weight packs
// The conv itself is conv1_16_kernel in gather mode (channel index of the gathered operand = tap*Cin + c, tap = i*4 + j*2 + k).
// weights of the 2x2x2 stride-2 conv [Cout][Cin][8] -> wd[Cout][tap*Cin + c] (forward) and wdT[tap*Cin + c][Cout] (data gradient)
__global__ void pack_down16_kernel(const float* __restrict__ w, float* __restrict__ wd, float* __restrict__ wdT, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * Cin * 8) return;
    const int tap = i & 7, c = (i >> 3) % Cin, o = i / (8 * Cin);
    const float v = w[i];
    wd[(size_t)o * 8 * Cin + tap * Cin + c] = v;
    wdT[(size_t)(tap * Cin + c) * Cout + o] = v;
}
int pack_down16_launch(const float* w, float* wd, float* wdT, int Cout, int Cin, hipStream_t s) {
    const int total = Cout * Cin * 8;
    hipLaunchKernelGGL(pack_down16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, wd, wdT, Cout, Cin);
    RU_CHECK_LAUNCH("pack_down16_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ 1x1x1 convolution on C16 tensors (model.py:393,401; concat of model.py:424)
// y[v][o] = act( sum_c wm[o][c] * xcat[v][c] ) (+ add) on v_mfma_f32_16x16x4_f32 (exact f32): M = 16 output channels,
// N = 16 voxels, K = input channels.  Both operands are aligned float4 loads -- lane (r = l&15, g = l>>4) reads channels
// 4g..4g+3 of a 16-channel block of voxel r (B) / of weight row o0 + r (A) -- and MFMA step e consumes element e, i.e. the
// K index of lane group g in step e is channel 4g + e on both sides.  D: lane holds output channels 4g..4g+3 of voxel r:
// one aligned float4 of the C16 output.  A wave owns 64 voxels x COB*16 output channels.
// S2D: 0 plain, 1 gather (stride-2 conv), 2 scatter (its transpose) -- compile-time, so the K loop has no branch around its loads
// NSLOT > 0: fused GroupNorm-backward statistics of the stored output (Conv1Args::bst_*).  NSLOT = distinct 16-channel blocks of the OUTPUT
// tensor a workgroup touches: COB in the plain mode; in scatter mode the COB blocks of a workgroup are (tap, fine block) pairs and block
// cb carries fine block cb % NSLOT (NSLOT = fine channel blocks, 1 / 2 / 4) -- the sums and constants are kept per SLOT, not per block
// (registers: the kernel is a latency-bound stream and lives on occupancy).  The four waves' sums meet in LDS: one partial per
// (workgroup, channel).
// PAIR (scatter mode with statistics, COB = 4, NSLOT = fine channel blocks = 1 / 2; round 6): the two x taps of a coarse voxel are stored by ONE instruction.
// A lane's accumulators hold (coarse voxel r, tap k) for k = 0 and 1 in two different blocks, and the plain epilogue stores each block by itself: 16 pieces of 64
// bytes, 128 bytes apart, per instruction -- half of every line it touches, like the loads of the residual and of the GroupNorm-backward operand at the same
// addresses (the level-0 launch ran at 3.7 TB/s where its whole-line siblings stream at 5.4-5.9).  Here the lanes change roles for the epilogue: lane (r', g) of
// half h takes, through ds_bpermute, tap r' & 1 of coarse voxel 8 h + (r' >> 1) -- 16 consecutive FINE voxels x 64 bytes = 1 KB of whole lines per instruction.
// Same values at the same addresses (bit-identical output); the statistics are summed in another lane order.
template <int COB, int S2D, int NSLOT, bool PAIR = false>
__global__ __launch_bounds__(256, (PAIR || NSLOT > 0) ? 2 : 1) void conv1_16_kernel(const Conv1Args a, int nvt) {
    static_assert(!PAIR || (S2D == 2 && COB == 4 && (NSLOT == 1 || NSLOT == 2)), "paired stores: the scatter mode with statistics, four blocks per workgroup");
    constexpr bool BST = NSLOT > 0;
    constexpr int NSL = BST ? NSLOT : 1;
    __shared__ float red[BST ? 4 * NSL * 16 * 2 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.z, cog = blockIdx.y;
    const size_t V = a.V;
    const int r = lane & 15, g = lane >> 4;
    const int nkb0 = a.C0 >> 4, nkb1 = a.C1 >> 4, nkb = nkb0 + nkb1, CBo = a.Cout >> 4;
    // fine-grid geometry of the 2x2x2 stride-2 modes
    const int Hf = 2 * a.Hc, Wf = 2 * a.Wc;
    const size_t Vf = V * 8;
    const int CBf_out = CBo >> 3;                        // scatter: channel blocks of the fine output tensor
    const int Cstat = S2D == 2 ? (a.Cout >> 3) : a.Cout; // BST: channels of the output tensor
    f32x4_c16 bs1[NSL], bs2[NSL], bk[NSL][3];
    if constexpr (BST) {
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            bs1[sl] = f32x4_c16{0.f, 0.f, 0.f, 0.f}; bs2[sl] = f32x4_c16{0.f, 0.f, 0.f, 0.f};
            const int cob = cog * COB + sl, chb = S2D == 2 ? cob % CBf_out : cob;
            const float* kp = a.bst_k + (size_t)n * 3 * Cstat + (cob < CBo ? chb : 0) * 16 + 4 * g;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) { const float4 q = *reinterpret_cast<const float4*>(kp + (size_t)t3 * Cstat); bk[sl][t3] = f32x4_c16{q.x, q.y, q.z, q.w}; }
        }
    }
    // nontemporal stores where the output outlasts the caches (>= 128 MB: the 16-channel level; profiles/r05_notes.txt, section 20)
    const bool nt_out = (size_t)a.N * a.Cout * V * sizeof(float) >= ((size_t)128 << 20);
    const int vt = blockIdx.x * 4 + wave;
    const bool live = vt < nvt;
    if (!live) { if constexpr (!BST) return; }
    if (live) {
    const size_t v0 = (size_t)vt * 64;
    size_t vb[4], fv[4];                                 // float offset of this lane's voxel (coarse) / its fine corner voxel index
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        size_t v = v0 + 16 * t + r;
        if (v >= V) v = V - 1;
        vb[t] = v * 16 + 4 * g;
        fv[t] = 0;
        if constexpr (S2D != 0) {
            const int xc = (int)(v % a.Wc);
            const size_t rr = v / a.Wc;
            const int yc = (int)(rr % a.Hc), zc = (int)(rr / a.Hc);
            fv[t] = ((size_t)(2 * zc) * Hf + 2 * yc) * Wf + 2 * xc;
        }
    }
    const float* wrow[COB];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb) {
        const int cob = cog * COB + cb;
        wrow[cb] = a.wT + (size_t)((cob < CBo ? cob : CBo - 1) * 16 + r) * a.ldw + 4 * g;
    }
    f32x4_c16 acc[4][COB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int cb = 0; cb < COB; ++cb) acc[t][cb] = f32x4_c16{0.f, 0.f, 0.f, 0.f};
    const int CBf_in = nkb0 >> 3;                        // gather: channel blocks of the fine input tensor
    // two K-blocks per trip (written out: `#pragma unroll 2` on the runtime-bound loop was refused by the optimizer): the loads of the second
    // block are in flight under the first block's MFMAs
    auto kstep = [&](int kb) __attribute__((always_inline)) {
        float4 xb[4];
        if constexpr (S2D == 1) {
            const int tap = kb / CBf_in, cbf = kb - tap * CBf_in;
            const size_t toff = ((size_t)(tap >> 2) * Hf + ((tap >> 1) & 1)) * Wf + (tap & 1);
            const float* src = a.x0 + ((size_t)(n * CBf_in + cbf) * Vf) * 16 + 4 * g;
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[t] = *reinterpret_cast<const float4*>(src + (fv[t] + toff) * 16);
        } else {
            const float* src = kb < nkb0 ? a.x0 + ((size_t)(n * nkb0 + kb) * V) * 16 : a.x1 + ((size_t)(n * nkb1 + kb - nkb0) * V) * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[t] = *reinterpret_cast<const float4*>(src + vb[t]);
        }
#pragma unroll
        for (int cb = 0; cb < COB; ++cb) {
            const float4 wv = *reinterpret_cast<const float4*>(wrow[cb] + kb * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xb[t].x, acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xb[t].y, acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xb[t].z, acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xb[t].w, acc[t][cb], 0, 0, 0);
            }
        }
    };
    int kb = 0;
    for (; kb + 1 < nkb; kb += 2) { kstep(kb); kstep(kb + 1); }
    if (kb < nkb) kstep(kb);
    if constexpr (PAIR) {
        // Two phases per 16-voxel tile t, so that the epilogue's loads travel TOGETHER: with the operand loads inside `if (a.add)` / behind a store, every
        // load of the plain epilogue was followed by s_waitcnt vmcnt(0) -- 48 dependent memory round trips per wave.  Phase A: roles, addresses, and all eight
        // operand loads of the tile (residual + GroupNorm-backward operand of the four (half, tap pair) roles), unconditional from valid addresses; phase B:
        // the lane exchange, the arithmetic, the stores.  (A launch with a LeakyReLU mask takes the plain epilogue: not a launch of the network in this mode.)
        const int k = r & 1;
        constexpr int CBf = NSLOT;                                           // fine channel blocks (the launch's nslot)
        const bool has_add = a.add != nullptr;
        const float* addp = has_add ? a.add : a.bst_y;                       // (absent: any valid tensor of the same extents, dropped by a select)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            size_t idx[2][2];
            bool ok[2];
            int srcl[2];
            float4 dadd[2][2], dyq[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                srcl[h] = 16 * g + 8 * h + (r >> 1);                         // the lane that computed this role's coarse voxel for channel quad g
                ok[h] = v0 + 16 * t + 8 * h + (r >> 1) < V;
                const unsigned flo = __shfl((unsigned)(fv[t] & 0xffffffffu), srcl[h]), fhi = __shfl((unsigned)(fv[t] >> 32), srcl[h]);
                const size_t fvs = ((size_t)fhi << 32) | flo;               // (of the source lane's CLAMPED voxel: always a valid address)
#pragma unroll
                for (int pa = 0; pa < 2; ++pa) {
                    const int cbA = CBf == 1 ? 2 * pa : pa;
                    const int cobA = cog * COB + cbA, tapA = cobA / CBf, cbf = cobA - tapA * CBf;      // tapA is even: its x tap is 0, the partner block's 1
                    const size_t toff = ((size_t)(tapA >> 2) * Hf + ((tapA >> 1) & 1)) * Wf + k;
                    idx[h][pa] = ((size_t)(n * CBf + cbf) * Vf + fvs + toff) * 16 + 4 * g;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int pa = 0; pa < 2; ++pa) {
                    dadd[h][pa] = *reinterpret_cast<const float4*>(addp + idx[h][pa]);
                    dyq[h][pa] = *reinterpret_cast<const float4*>(a.bst_y + idx[h][pa]);
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int pa = 0; pa < 2; ++pa) {
                    const int cbA = CBf == 1 ? 2 * pa : pa, cbB = CBf == 1 ? 2 * pa + 1 : pa + 2;
                    f32x4_c16 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float va = __shfl(acc[t][cbA][e], srcl[h]), vb = __shfl(acc[t][cbB][e], srcl[h]);
                        o4[e] = k ? vb : va;
                    }
                    float4 o = make_float4(lrelu(o4[0], a.out_slope), lrelu(o4[1], a.out_slope), lrelu(o4[2], a.out_slope), lrelu(o4[3], a.out_slope));
                    const float4 d = dadd[h][pa];
                    if (has_add) { o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
                    if (ok[h]) {
                        if (nt_out) __builtin_nontemporal_store(f32x4_c16{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_c16*>(a.y + idx[h][pa]));
                        else *reinterpret_cast<float4*>(a.y + idx[h][pa]) = o;
                    }
                    const float4 yq = dyq[h][pa];
                    const float okf = ok[h] ? 1.f : 0.f;                     // (roles beyond the tensor contribute zeros to the sums)
                    const f32x4_c16 ov = f32x4_c16{o.x, o.y, o.z, o.w} * okf, os = ov * a.bst_slope;
                    const int sl_ = CBf == 1 ? 0 : pa;                       // slot = fine channel block
                    static_for_c1<NSL>([&](auto SL) {
                        constexpr int s_ = decltype(SL)::value;
                        if (sl_ == s_) {
                            const f32x4_c16 u = f32x4_c16{yq.x, yq.y, yq.z, yq.w} * bk[s_][0] + bk[s_][1];
                            f32x4_c16 dh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) dh[e] = u[e] > bk[s_][2][e] ? ov[e] : os[e];
                            bs1[s_] += dh;
                            bs2[s_] += dh * u;
                        }
                    });
                }
            }
        }
    } else if constexpr (BST) {
    // two phases per tile, like the paired epilogue above: addresses and ALL operand loads of the tile's COB blocks first (one wave-uniform test per operand
    // kind, not one per load -- a load inside `if (a.add)` behind a store was followed by s_waitcnt vmcnt(0): up to 3 x COB x 4 dependent round trips per
    // wave), then the arithmetic and the stores
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const size_t v = v0 + 16 * t + r;
        const bool vok = v < V;
        const size_t vc = vok ? v : V - 1;
        size_t idx[COB];
        float* ydst[COB];
        bool second[COB];
#pragma unroll
        for (int cb = 0; cb < COB; ++cb) {
            int cob = cog * COB + cb;
            if (cob >= CBo) cob = CBo - 1;                  // (clamped: loaded and dropped)
            ydst[cb] = a.y;
            second[cb] = false;
            if constexpr (S2D == 2) {
                const int tap = cob / CBf_out, cbf = cob - tap * CBf_out;
                const size_t toff = ((size_t)(tap >> 2) * Hf + ((tap >> 1) & 1)) * Wf + (tap & 1);
                idx[cb] = ((size_t)(n * CBf_out + cbf) * Vf + fv[t] + toff) * 16 + 4 * g;
            } else {
                int cbt = CBo, cl = cob;
                if (a.y1) {                              // split output: channel blocks >= Cout0/16 belong to the second tensor
                    const int CB0o = a.Cout0 >> 4;
                    second[cb] = cob >= CB0o;
                    cbt = second[cb] ? CBo - CB0o : CB0o;
                    cl = second[cb] ? cob - CB0o : cob;
                    ydst[cb] = second[cb] ? a.y1 : a.y;
                }
                idx[cb] = ((size_t)(n * cbt + cl) * V + vc) * 16 + 4 * g;
            }
        }
        float4 mq[COB], dq[COB], yq[BST ? COB : 1];
        if (a.mask) {
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) mq[cb] = *reinterpret_cast<const float4*>(a.mask + ((second[cb] || !a.y1) ? idx[cb] : 0));
        }
        if (a.add) {
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) dq[cb] = *reinterpret_cast<const float4*>(a.add + idx[cb]);
        }
        if constexpr (BST) {
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) yq[cb] = *reinterpret_cast<const float4*>(a.bst_y + idx[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < COB; ++cb) {
            if (!vok || cog * COB + cb >= CBo) continue;
            float4 o = make_float4(lrelu(acc[t][cb][0], a.out_slope), lrelu(acc[t][cb][1], a.out_slope),
                                   lrelu(acc[t][cb][2], a.out_slope), lrelu(acc[t][cb][3], a.out_slope));
            if (a.mask && (second[cb] || !a.y1)) {
                const float4 m = mq[cb];
                o.x = m.x > 0.f ? o.x : o.x * a.mask_slope; o.y = m.y > 0.f ? o.y : o.y * a.mask_slope;
                o.z = m.z > 0.f ? o.z : o.z * a.mask_slope; o.w = m.w > 0.f ? o.w : o.w * a.mask_slope;
            }
            if (a.add) { const float4 d = dq[cb]; o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
            if (nt_out) __builtin_nontemporal_store(f32x4_c16{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_c16*>(ydst[cb] + idx[cb]));
            else *reinterpret_cast<float4*>(ydst[cb] + idx[cb]) = o;
            if constexpr (BST) {                         // sums of the STORED gradient (sb_out_tile_bst's arithmetic)
                const f32x4_c16 ov = f32x4_c16{o.x, o.y, o.z, o.w}, os = ov * a.bst_slope;
                static_for_c1<NSL>([&](auto SL) {        // (compile-time slot: no register array is indexed by a run-time value)
                    constexpr int s_ = decltype(SL)::value;
                    if (cb % NSL == s_) {
                        const f32x4_c16 u = f32x4_c16{yq[cb].x, yq[cb].y, yq[cb].z, yq[cb].w} * bk[s_][0] + bk[s_][1];
                        f32x4_c16 dh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dh[e] = u[e] > bk[s_][2][e] ? ov[e] : os[e];
                        bs1[s_] += dh;
                        bs2[s_] += dh * u;
                    }
                });
            }
        }
    }
    } else {
    // no statistics: the forward launches, which have no epilogue operand at all -- they live on occupancy, and the batched form costs 24-28 registers
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const size_t v = v0 + 16 * t + r;
        if (v >= V) continue;
#pragma unroll
        for (int cb = 0; cb < COB; ++cb) {
            const int cob = cog * COB + cb;
            if (cob >= CBo) continue;
            size_t idx;
            float* ydst = a.y;
            bool second = false;
            if constexpr (S2D == 2) {
                const int tap = cob / CBf_out, cbf = cob - tap * CBf_out;
                const size_t toff = ((size_t)(tap >> 2) * Hf + ((tap >> 1) & 1)) * Wf + (tap & 1);
                idx = ((size_t)(n * CBf_out + cbf) * Vf + fv[t] + toff) * 16 + 4 * g;
            } else {
                int cbt = CBo, cl = cob;
                if (a.y1) {                              // split output: channel blocks >= Cout0/16 belong to the second tensor
                    const int CB0o = a.Cout0 >> 4;
                    second = cob >= CB0o;
                    cbt = second ? CBo - CB0o : CB0o;
                    cl = second ? cob - CB0o : cob;
                    ydst = second ? a.y1 : a.y;
                }
                idx = ((size_t)(n * cbt + cl) * V + v) * 16 + 4 * g;
            }
            float4 o = make_float4(lrelu(acc[t][cb][0], a.out_slope), lrelu(acc[t][cb][1], a.out_slope),
                                   lrelu(acc[t][cb][2], a.out_slope), lrelu(acc[t][cb][3], a.out_slope));
            if (a.mask && (second || !a.y1)) {
                const float4 m = *reinterpret_cast<const float4*>(a.mask + idx);
                o.x = m.x > 0.f ? o.x : o.x * a.mask_slope; o.y = m.y > 0.f ? o.y : o.y * a.mask_slope;
                o.z = m.z > 0.f ? o.z : o.z * a.mask_slope; o.w = m.w > 0.f ? o.w : o.w * a.mask_slope;
            }
            if (a.add) { const float4 d = *reinterpret_cast<const float4*>(a.add + idx); o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
            if (nt_out) __builtin_nontemporal_store(f32x4_c16{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4_c16*>(ydst + idx));
            else *reinterpret_cast<float4*>(ydst + idx) = o;
            if constexpr (BST) {                         // sums of the STORED gradient (sb_out_tile_bst's arithmetic)
                constexpr int sl = 0;
                (void)sl;
                const float4 yq = *reinterpret_cast<const float4*>(a.bst_y + idx);
                const f32x4_c16 ov = f32x4_c16{o.x, o.y, o.z, o.w}, os = ov * a.bst_slope;
                static_for_c1<NSL>([&](auto SL) {        // (compile-time slot: no register array is indexed by a run-time value)
                    constexpr int s_ = decltype(SL)::value;
                    if (cb % NSL == s_) {
                        const f32x4_c16 u = f32x4_c16{yq.x, yq.y, yq.z, yq.w} * bk[s_][0] + bk[s_][1];
                        f32x4_c16 dh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dh[e] = u[e] > bk[s_][2][e] ? ov[e] : os[e];
                        bs1[s_] += dh;
                        bs2[s_] += dh * u;
                    }
                });
            }
        }
    }
    }                                                    // !PAIR
    }                                                    // live
    if constexpr (BST) {
        // lanes sharing g hold the same channel quad: fold the 16 voxel lanes, then the four waves -- one partial per (workgroup, channel)
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { bs1[sl][e] += __shfl_xor(bs1[sl][e], o); bs2[sl][e] += __shfl_xor(bs2[sl][e], o); }
            }
            if (r == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { red[((wave * NSL + sl) * 16 + 4 * g + e) * 2] = bs1[sl][e]; red[((wave * NSL + sl) * 16 + 4 * g + e) * 2 + 1] = bs2[sl][e]; }
            }
        }
        __syncthreads();
        const int nblk = S2D == 2 ? gridDim.x * gridDim.y : gridDim.x;
        const int blk = S2D == 2 ? blockIdx.y * gridDim.x + blockIdx.x : blockIdx.x;
        for (int i = threadIdx.x; i < NSL * 16; i += 256) {
            const int slot = i >> 4, c = i & 15;
            const int cob0 = cog * COB + slot;
            if (cob0 >= CBo) continue;
            const int chb = S2D == 2 ? cob0 % CBf_out : cob0;
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u1 += red[((w * NSL + slot) * 16 + c) * 2]; u2 += red[((w * NSL + slot) * 16 + c) * 2 + 1]; }
            float* p = a.stat_partials + (((size_t)n * Cstat + chb * 16 + c) * nblk + blk) * 2;
            p[0] = u1; p[1] = u2;
        }
    }
}
// output channel blocks per workgroup.  Scatter mode (the transposed stride-2 conv: a streaming kernel whose 16-byte stores land 128 bytes
// apart): RU_C1_SCATTER_COB (env, tools only) overrides the default for A/B runs
static int conv1_16_cob(const Conv1Args& a) {
    const int CBo = a.Cout / 16;
    int cob = CBo >= 4 ? 4 : (CBo >= 2 ? 2 : 1);
    if (a.s2d == 2) {
        static const int forced = [] { const char* e = getenv("RU_C1_SCATTER_COB"); return e ? atoi(e) : 0; }();
        if ((forced == 2 || forced == 4 || forced == 1) && forced <= CBo) cob = forced;
    }
    return cob;
}
int conv1_16_bst_nblk(const Conv1Args& a) {
    if (a.s2d == 1 || a.y1 || a.Cout % 16 || a.V == 0) return 0;
    const int nvt = (int)((a.V + 63) / 64), CBo = a.Cout / 16;
    const int cob = conv1_16_cob(a);
    const int gx = cdiv(nvt, 4), gy = cdiv(CBo, cob);
    if (a.s2d == 2) {
        const int cbf = CBo >> 3;                        // every workgroup must carry whole groups of taps of the same fine channel blocks
        if (a.Cout % 128 || (cbf != 1 && cbf != 2 && cbf != 4) || !(cob % cbf == 0 || cbf % cob == 0)) return 0;
        if (cbf > cob) return 0;                         // (a workgroup would carry a part of the fine blocks only: not a launch of the network)
        return gx * gy;
    }
    return gx;
}
// a.wT is read as wm[Cout][C0 + C1] with row pitch a.ldw (row-major OUT x IN: the reference layout of a 1x1x1 weight)
int conv1_16_launch(const Conv1Args& a, hipStream_t s) {
    RU_REQUIRE(a.N > 0 && a.C0 > 0 && a.Cout > 0 && a.V > 0, "conv1_16: bad shape");
    RU_REQUIRE(a.C0 % 16 == 0 && a.C1 % 16 == 0 && a.Cout % 16 == 0 && a.ldw >= a.C0 + a.C1 && a.ldw % 4 == 0, "conv1_16: channels must be multiples of 16");
    RU_REQUIRE(!a.y1 || (!a.s2d && !a.add && a.Cout0 > 0 && a.Cout0 < a.Cout && a.Cout0 % 16 == 0), "conv1_16: a split output needs the plain mode, no residual and a split at a multiple of 16");
    if (a.s2d) {
        RU_REQUIRE(a.Dc > 0 && a.Hc > 0 && a.Wc > 0 && (size_t)a.Dc * a.Hc * a.Wc == a.V && a.C1 == 0, "conv1_16: bad stride-2 geometry");
        RU_REQUIRE(a.s2d == 1 ? a.C0 % 128 == 0 : a.Cout % 128 == 0, "conv1_16: stride-2 modes need 8 x (multiple of 16) channels");
    }
    const int nvt = (int)((a.V + 63) / 64), CBo = a.Cout / 16;
    const int cob = conv1_16_cob(a);
    const bool bst = a.bst_y != nullptr;
    RU_REQUIRE(!bst || (a.bst_k && a.stat_partials && a.s2d != 1 && !a.y1 && conv1_16_bst_nblk(a) > 0), "conv1_16: fused GroupNorm-backward statistics need the plain or scatter mode with whole channel blocks per workgroup");
    dim3 grid((unsigned)cdiv(nvt, 4), (unsigned)cdiv(CBo, cob), (unsigned)a.N);
    const int nslot = !bst ? 0 : (a.s2d == 2 ? (CBo >> 3) : cob);          // distinct output channel blocks per workgroup (conv1_16_bst_nblk checked the shape)
    // paired stores (the scatter mode's two x taps in one whole-line instruction): 8 consecutive coarse voxels of a 64-voxel wave tile must share a row; RU_C1_PAIR=0: off (A/B)
    static const bool pair_off = [] { const char* e = getenv("RU_C1_PAIR"); return e && *e == '0'; }();
    const bool pair = a.s2d == 2 && bst && !pair_off && a.Wc % 8 == 0 && cob == 4 && !a.mask;
#define RU_C1_LAUNCH(COB_)                                                                                          \
    do {                                                                                                           \
        if (a.s2d == 1) hipLaunchKernelGGL((conv1_16_kernel<COB_, 1, 0>), grid, dim3(256), 0, s, a, nvt);           \
        else if (a.s2d == 2 && nslot == 1 && pair && COB_ == 4) hipLaunchKernelGGL((conv1_16_kernel<4, 2, 1, true>), grid, dim3(256), 0, s, a, nvt);  \
        else if (a.s2d == 2 && nslot == 2 && pair && COB_ == 4) hipLaunchKernelGGL((conv1_16_kernel<4, 2, 2, true>), grid, dim3(256), 0, s, a, nvt);  \
        else if (a.s2d == 2 && nslot == 1) hipLaunchKernelGGL((conv1_16_kernel<COB_, 2, 1>), grid, dim3(256), 0, s, a, nvt);  \
        else if (a.s2d == 2 && nslot == 2) hipLaunchKernelGGL((conv1_16_kernel<COB_, 2, (COB_ >= 2 ? 2 : 1)>), grid, dim3(256), 0, s, a, nvt);  \
        else if (a.s2d == 2 && nslot == 4) hipLaunchKernelGGL((conv1_16_kernel<COB_, 2, (COB_ >= 4 ? 4 : 1)>), grid, dim3(256), 0, s, a, nvt);  \
        else if (a.s2d == 2) hipLaunchKernelGGL((conv1_16_kernel<COB_, 2, 0>), grid, dim3(256), 0, s, a, nvt);      \
        else if (bst) hipLaunchKernelGGL((conv1_16_kernel<COB_, 0, COB_>), grid, dim3(256), 0, s, a, nvt);          \
        else hipLaunchKernelGGL((conv1_16_kernel<COB_, 0, 0>), grid, dim3(256), 0, s, a, nvt);                      \
    } while (0)
    if (cob == 4) RU_C1_LAUNCH(4);
    else if (cob == 2) RU_C1_LAUNCH(2);
    else RU_C1_LAUNCH(1);
#undef RU_C1_LAUNCH
    RU_CHECK_LAUNCH("conv1_16_kernel");
    return RU_OK;
}

}  // namespace ru
