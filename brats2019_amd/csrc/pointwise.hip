// pointwise.hip -- the HBM-bound kernels of the ResUNet path (gfx950): 1x1x1 convolution, space-to-depth for the
// 2x2x2 stride-2 convolution, GroupNorm statistics / apply / backward, LeakyReLU, sigmoid, trilinear x2 up-sampling
// and its transpose, the Dice+BCE criterion, and Adam(amsgrad).  All are streaming kernels: 16-byte coalesced
// accesses along W (NCDHW), one pass per tensor, two-stage deterministic reductions (no float atomics).
#include "pw_helpers.hpp"

namespace ru {

// ------------------------------------------------------------------ 1x1x1 convolution (model.py:393,401), VALU with scalar weights
// Each thread owns VEC consecutive voxels and 16 output channels; the weight of (c, o) is wave-uniform, so it is
// fetched by scalar loads and used as the SGPR operand of v_fmac_f32.  The channel concat of model.py:424 is the
// two-pointer input (x0 then x1) -- the cat is never materialised.
template <int VEC, int COT>
__global__ __launch_bounds__(256) void conv1_kernel(const Conv1Args a) {
    const int n = blockIdx.z;
    const int co0 = blockIdx.y * COT;
    const size_t V = a.V;
    const size_t v = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (v >= V) return;
    float acc[COT][VEC];
#pragma unroll
    for (int o = 0; o < COT; ++o)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[o][k] = 0.f;
    int widx[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) widx[o] = (co0 + o < a.Cout) ? co0 + o : a.Cout - 1;

    const float* xp = a.x0 + (size_t)n * a.C0 * V + v;
    const float* wr = a.wT;
#pragma unroll 4
    for (int c = 0; c < a.C0; ++c) {
        float xv[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(xp + (size_t)c * V);
            xv[0] = t.x; xv[1 % VEC] = t.y; xv[2 % VEC] = t.z; xv[3 % VEC] = t.w;
        } else {
            xv[0] = xp[(size_t)c * V];
        }
#pragma unroll
        for (int o = 0; o < COT; ++o) {
            const float w = wr[(size_t)c * a.ldw + widx[o]];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[o][k] += w * xv[k];
        }
    }
    if (a.x1) {
        const float* xq = a.x1 + (size_t)n * a.C1 * V + v;
        const float* wq = a.wT + (size_t)a.C0 * a.ldw;
#pragma unroll 4
        for (int c = 0; c < a.C1; ++c) {
            float xv[VEC];
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xq + (size_t)c * V);
                xv[0] = t.x; xv[1 % VEC] = t.y; xv[2 % VEC] = t.z; xv[3 % VEC] = t.w;
            } else {
                xv[0] = xq[(size_t)c * V];
            }
#pragma unroll
            for (int o = 0; o < COT; ++o) {
                const float w = wq[(size_t)c * a.ldw + widx[o]];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[o][k] += w * xv[k];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COT; ++o) {
        if (co0 + o >= a.Cout) continue;
        const size_t idx = ((size_t)n * a.Cout + co0 + o) * V + v;
        float r[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = lrelu(acc[o][k], a.out_slope);
        if (a.add) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) r[k] += a.add[idx + k];
        }
        if (VEC == 4) *reinterpret_cast<float4*>(a.y + idx) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
        else a.y[idx] = r[0];
    }
}

// Deep levels (few voxels, hundreds of channels): the 1x1x1 conv is a GEMM  y[o][v] = sum_c wT[c][o] * x[c][v]  with
// M = 16 output channels, N = 64 voxels (4 tiles) per wave and K = Cin walked 4 at a time on v_mfma_f32_16x16x4_f32
// (exact f32, k-ordered like the scalar kernel).  Operands come straight from global / L2: A lane (o = l&15, k = l>>4),
// B lane (v = l&15, k = l>>4); D lane holds 4 consecutive output channels of one voxel.
typedef float f32x4_pw __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv1_mfma_kernel(const Conv1Args a, int nvt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.z;
    const int o0 = blockIdx.y * 16;
    const int vt = blockIdx.x * 4 + wave;                 // 64-voxel tile of this wave
    if (vt >= nvt) return;
    const size_t V = a.V;
    const size_t v0 = (size_t)vt * 64;
    const int r = lane & 15, k = lane >> 4;
    const int Ct = a.C0 + a.C1;
    const int oa = o0 + r < a.Cout ? o0 + r : a.Cout - 1;
    size_t vb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const size_t v = v0 + 16 * t + r; vb[t] = v < V ? v : V - 1; }
    f32x4_pw acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_pw{0.f, 0.f, 0.f, 0.f};
    const float* x0 = a.x0 + (size_t)n * a.C0 * V;
    const float* x1 = a.x1 ? a.x1 + (size_t)n * a.C1 * V : a.x0;
    auto cstep = [&](int c0) __attribute__((always_inline)) {
        const int c = c0 + k;
        const bool cok = c < Ct;
        const int cc = cok ? c : Ct - 1;
        const float av = cok ? a.wT[(size_t)cc * a.ldw + oa] : 0.f;
        const float* xp = cc < a.C0 ? x0 + (size_t)cc * V : x1 + (size_t)(cc - a.C0) * V;
        float bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = xp[vb[t]];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
    };
    int c0 = 0;                                           // four K-groups per trip, written out (the runtime-bound loop refused `#pragma unroll 4`)
    for (; c0 + 12 < Ct; c0 += 16) { cstep(c0); cstep(c0 + 4); cstep(c0 + 8); cstep(c0 + 12); }
    for (; c0 < Ct; c0 += 4) cstep(c0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const size_t v = v0 + 16 * t + r;
        if (v >= V) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = o0 + k * 4 + q;
            if (o >= a.Cout) continue;
            const size_t idx = ((size_t)n * a.Cout + o) * V + v;
            float y = lrelu(acc[t][q], a.out_slope);
            if (a.add) y += a.add[idx];
            a.y[idx] = y;
        }
    }
}

int conv1_launch(const Conv1Args& a, hipStream_t s) {
    RU_REQUIRE(a.N > 0 && a.C0 > 0 && a.Cout > 0 && a.V > 0 && a.ldw >= a.Cout, "conv1: bad shape");
    const bool vec = (a.V % 4) == 0;
    // few voxels x many channels (deep levels): one voxel and 4 outputs per thread, otherwise the launch is a handful of
    // workgroups each walking hundreds of input channels serially (latency-bound)
    const long wg_big = (long)((a.V / 4 + 255) / 256) * cdiv(a.Cout, 16) * a.N;
    if (wg_big < 256 && a.C0 + a.C1 >= 32) {
        const int nvt = (int)((a.V + 63) / 64);
        dim3 grid((unsigned)cdiv(nvt, 4), (unsigned)cdiv(a.Cout, 16), (unsigned)a.N);
        hipLaunchKernelGGL(conv1_mfma_kernel, grid, dim3(256), 0, s, a, nvt);
    } else if (!vec || wg_big < 256) {
        dim3 grid((unsigned)((a.V + 255) / 256), (unsigned)cdiv(a.Cout, 4), (unsigned)a.N);
        hipLaunchKernelGGL((conv1_kernel<1, 4>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((unsigned)((a.V / 4 + 255) / 256), (unsigned)cdiv(a.Cout, 16), (unsigned)a.N);
        hipLaunchKernelGGL((conv1_kernel<4, 16>), grid, dim3(256), 0, s, a);
    }
    RU_CHECK_LAUNCH("conv1_kernel");
    return RU_OK;
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int c = i / rows, r = i % rows;      // dst index i = c*rows + r
    dst[i] = src[(size_t)r * cols + c];
}
int transpose_launch(const float* src, float* dst, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, s, src, dst, rows, cols);
    RU_CHECK_LAUNCH("transpose_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ space-to-depth / depth-to-space (2x2x2 stride-2 conv, model.py:361-363)
// The non-overlapping 2^3 patches make the down-sampling conv a 1x1x1 conv over 8*Cin channels;
// channel order c*8 + i*4 + j*2 + k matches the flattening of the reference weight [Cout][Cin][2][2][2].
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int D, int H, int W) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)NC * Do * Ho * Wo;
    const size_t Vo = (size_t)Do * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        size_t r = i / Wo;
        const int yo = (int)(r % Ho); r /= Ho;
        const int zo = (int)(r % Do);
        const size_t nc = r / Do;
        const float* xp = x + ((nc * D + 2 * zo) * H + 2 * yo) * (size_t)W + 2 * xo;
        float* yp = y + nc * 8 * Vo + ((size_t)zo * Ho + yo) * Wo + xo;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            const float2 t = *reinterpret_cast<const float2*>(xp + ((size_t)(ij >> 1) * H + (ij & 1)) * W);
            yp[(size_t)(ij * 2) * Vo] = t.x;
            yp[(size_t)(ij * 2 + 1) * Vo] = t.y;
        }
    }
}
__global__ __launch_bounds__(256) void d2s_kernel(const float* __restrict__ y, float* __restrict__ x, int NC, int D, int H, int W) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)NC * Do * Ho * Wo;
    const size_t Vo = (size_t)Do * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        size_t r = i / Wo;
        const int yo = (int)(r % Ho); r /= Ho;
        const int zo = (int)(r % Do);
        const size_t nc = r / Do;
        float* xp = x + ((nc * D + 2 * zo) * H + 2 * yo) * (size_t)W + 2 * xo;
        const float* yp = y + nc * 8 * Vo + ((size_t)zo * Ho + yo) * Wo + xo;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            float2 t;
            t.x = yp[(size_t)(ij * 2) * Vo];
            t.y = yp[(size_t)(ij * 2 + 1) * Vo];
            *reinterpret_cast<float2*>(xp + ((size_t)(ij >> 1) * H + (ij & 1)) * W) = t;
        }
    }
}
int s2d_launch(const float* x, float* y, int N, int C, int D, int H, int W, hipStream_t s) {
    RU_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "s2d: extents must be even");
    const size_t total = (size_t)N * C * (D / 2) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(s2d_kernel, dim3(grid1d(total, 256)), dim3(256), 0, s, x, y, N * C, D, H, W);
    RU_CHECK_LAUNCH("s2d_kernel");
    return RU_OK;
}
int d2s_launch(const float* y, float* x, int N, int C, int D, int H, int W, hipStream_t s) {
    RU_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "d2s: extents must be even");
    const size_t total = (size_t)N * C * (D / 2) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(d2s_kernel, dim3(grid1d(total, 256)), dim3(256), 0, s, y, x, N * C, D, H, W);
    RU_CHECK_LAUNCH("d2s_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ GroupNorm (model.py:95-96,338)
constexpr int GN_CHUNK = 8192;   // elements of one (n,c) row reduced by one workgroup
int gn_stats_tiles(size_t V) { return (int)((V + GN_CHUNK - 1) / GN_CHUNK); }
int gn_bwd_tiles(size_t V) { return gn_stats_tiles(V); }

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ partials, size_t V, int nblk) {
    __shared__ float buf[4];
    const size_t row = blockIdx.y;              // n*C + c
    const size_t v0 = (size_t)blockIdx.x * GN_CHUNK;
    const size_t v1 = v0 + GN_CHUNK < V ? v0 + GN_CHUNK : V;
    const float* xp = x + row * V;
    float s1 = 0.f, s2 = 0.f;
    if ((V & 3) == 0) {
        for (size_t v = v0 + threadIdx.x * 4; v < v1; v += 1024) {
            const float4 t = *reinterpret_cast<const float4*>(xp + v);
            s1 += (t.x + t.y) + (t.z + t.w);
            s2 += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
    } else {
        for (size_t v = v0 + threadIdx.x; v < v1; v += 256) { const float t = xp[v]; s1 += t; s2 += t * t; }
    }
    s1 = block_sum(s1, buf);
    s2 = block_sum(s2, buf);
    if (threadIdx.x == 0) {
        float* p = partials + (row * nblk + blockIdx.x) * 2;
        p[0] = s1; p[1] = s2;
    }
}
int gn_stats_launch(const float* x, float* partials, int N, int C, size_t V, hipStream_t s) {
    const int nblk = gn_stats_tiles(V);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, N * C), dim3(256), 0, s, x, partials, V, nblk);
    RU_CHECK_LAUNCH("gn_stats_kernel");
    return RU_OK;
}

// one workgroup per (n, g): float64 combine of the per-tile float32 (sum, sumsq) partials
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partials, int nblk, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ scale, float* __restrict__ shift, int C, size_t V, int G, float eps,
                                                          float* __restrict__ bst_k) {
    __shared__ double buf[4][2];
    __shared__ float sh[2];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int cpg = C / G;
    const float* p = partials + ((size_t)n * C + (size_t)g * cpg) * nblk * 2;   // cpg*nblk contiguous (sum,sumsq) pairs
    const int cnt = cpg * nblk;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < cnt; i += 256) { s1 += (double)p[2 * i]; s2 += (double)p[2 * i + 1]; }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { buf[threadIdx.x >> 6][0] = s1; buf[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = (buf[0][0] + buf[1][0]) + (buf[2][0] + buf[3][0]);
        s2 = (buf[0][1] + buf[1][1]) + (buf[2][1] + buf[3][1]);
        const double m = (double)cpg * (double)V;
        const double mu = s1 / m;
        double var = s2 / m - mu * mu;
        if (var < 0.0) var = 0.0;
        const double rs = 1.0 / sqrt(var + (double)eps);
        sh[0] = (float)mu; sh[1] = (float)rs;
        mean[blockIdx.x] = (float)mu;
        rstd[blockIdx.x] = (float)rs;
    }
    __syncthreads();
    if (threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        const float a = gamma[c] * sh[1];
        scale[n * C + c] = a;
        shift[n * C + c] = beta[c] - sh[0] * a;
        if (bst_k) {                             // constants of the fused GroupNorm-backward statistics (Conv3Args::bst_k): u = y*k1 + k2 = sign(gamma)*xhat, mask <=> u > thr = -beta/|gamma|
            const float gm = gamma[c], bt = beta[c], sg = gm < 0.f ? -1.f : 1.f;
            float* kn = bst_k + (size_t)n * 3 * C;
            kn[c] = sg * sh[1];
            kn[C + c] = -sg * sh[0] * sh[1];
            kn[2 * C + c] = gm == 0.f ? (bt > 0.f ? -INFINITY : INFINITY) : -bt / fabsf(gm);
        }
    }
}
int gn_finalize_launch(const float* partials, int nblk, const float* gamma, const float* beta, float* mean, float* rstd,
                       float* scale, float* shift, int N, int C, size_t V, int G, float eps, hipStream_t s, float* bst_k) {
    RU_REQUIRE(C % G == 0 && C / G <= 256, "groupnorm: C must be divisible by G (and C/G <= 256)");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(256), 0, s, partials, nblk, gamma, beta, mean, rstd, scale, shift, C, V, G, eps, bst_k);
    RU_CHECK_LAUNCH("gn_finalize_kernel");
    return RU_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ res, float* __restrict__ y, size_t V, float slope) {
    const size_t row = blockIdx.y;
    const float a = scale[row], b = shift[row];
    const size_t base = row * V;
    for (size_t v = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; v < V; v += (size_t)gridDim.x * 256 * VEC) {
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + base + v);
            float4 o;
            o.x = lrelu(t.x * a + b, slope); o.y = lrelu(t.y * a + b, slope);
            o.z = lrelu(t.z * a + b, slope); o.w = lrelu(t.w * a + b, slope);
            if (res) {
                const float4 r = *reinterpret_cast<const float4*>(res + base + v);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            *reinterpret_cast<float4*>(y + base + v) = o;
        } else {
            float o = lrelu(x[base + v] * a + b, slope);
            if (res) o += res[base + v];
            y[base + v] = o;
        }
    }
}
static inline dim3 rows_grid(size_t V, int vec, int rows) {
    size_t bx = (V / vec + 255) / 256;
    if (bx > 4096) bx = 4096;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)rows);
}
int gn_apply_launch(const float* x, const float* scale, const float* shift, const float* res, float* y,
                    int N, int C, size_t V, float slope, hipStream_t s) {
    if (V % 4 == 0) hipLaunchKernelGGL(gn_apply_kernel<4>, rows_grid(V, 4, N * C), dim3(256), 0, s, x, scale, shift, res, y, V, slope);
    else hipLaunchKernelGGL(gn_apply_kernel<1>, rows_grid(V, 1, N * C), dim3(256), 0, s, x, scale, shift, res, y, V, slope);
    RU_CHECK_LAUNCH("gn_apply_kernel");
    return RU_OK;
}

// backward: S1 = sum dyh, S2 = sum dyh * xhat per (n,c) tile, dyh = dy * lrelu'(pre), pre = x*scale+shift
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float slope, float* __restrict__ partials, int C, size_t V, int G, int nblk) {
    __shared__ float buf[4];
    const size_t row = blockIdx.y;
    const int n = (int)(row / C), c = (int)(row % C);
    const int g = c / (C / G);
    const float a = scale[row], b = shift[row];
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    const size_t v0 = (size_t)blockIdx.x * GN_CHUNK;
    const size_t v1 = v0 + GN_CHUNK < V ? v0 + GN_CHUNK : V;
    const float* xp = x + row * V;
    const float* dp = dy + row * V;
    float s1 = 0.f, s2 = 0.f;
    if ((V & 3) == 0) {
        for (size_t v = v0 + threadIdx.x * 4; v < v1; v += 1024) {
            const float4 t = *reinterpret_cast<const float4*>(xp + v);
            const float4 d = *reinterpret_cast<const float4*>(dp + v);
            const float tx[4] = {t.x, t.y, t.z, t.w};
            const float dx[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dh = (tx[k] * a + b) > 0.f ? dx[k] : dx[k] * slope;
                s1 += dh;
                s2 += dh * ((tx[k] - mu) * rs);
            }
        }
    } else {
        for (size_t v = v0 + threadIdx.x; v < v1; v += 256) {
            const float t = xp[v], d = dp[v];
            const float dh = (t * a + b) > 0.f ? d : d * slope;
            s1 += dh;
            s2 += dh * ((t - mu) * rs);
        }
    }
    s1 = block_sum(s1, buf);
    s2 = block_sum(s2, buf);
    if (threadIdx.x == 0) {
        float* p = partials + (row * nblk + blockIdx.x) * 2;
        p[0] = s1; p[1] = s2;
    }
}
int gn_bwd_reduce_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* mean,
                         const float* rstd, float slope, float* partials, int N, int C, size_t V, int G, hipStream_t s) {
    const int nblk = gn_bwd_tiles(V);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(nblk, N * C), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, slope, partials, C, V, G, nblk);
    RU_CHECK_LAUNCH("gn_bwd_reduce_kernel");
    return RU_OK;
}

// one workgroup per group g.  Every (sample, channel-of-group) pair is reduced by ONE wave with shuffles only (fixed order:
// deterministic), one barrier, then the per-sample coefficients and the batch-ordered dgamma/dbeta sums.
// LPI lanes share one (sample, channel) item: 64 for long partial lists, 16 / 4 when a deep level has only a few partials per
// item but many items (then a wave finishes 4 / 16 items per pass instead of one).
template <int LPI>
__global__ __launch_bounds__(512) void gn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ coef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, size_t V, int G, int s2_sign) {
    extern __shared__ double S[];            // [N][cpg][2]
    const int g = blockIdx.x;
    const int cpg = C / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int IPW = 64 / LPI;
    const int sub = lane / LPI, sl = lane % LPI;
    const int nwave = blockDim.x >> 6;
    for (int item0 = wave * IPW; item0 < N * cpg; item0 += nwave * IPW) {
        const int item = item0 + sub;
        const bool ok = item < N * cpg;
        const int it = ok ? item : 0;
        const int n = it / cpg, j = it - n * cpg;
        const float2* p = reinterpret_cast<const float2*>(partials + ((size_t)n * C + g * cpg + j) * nblk * 2);
        double s1 = 0.0, s2 = 0.0;
        int i = sl;
        for (; i + 7 * LPI < nblk; i += 8 * LPI) {        // eight loads in flight: the kernel is a latency chain
            const float2 a = p[i], b = p[i + LPI], c = p[i + 2 * LPI], d = p[i + 3 * LPI];
            const float2 e = p[i + 4 * LPI], f = p[i + 5 * LPI], g2 = p[i + 6 * LPI], h = p[i + 7 * LPI];
            s1 += (((double)a.x + (double)b.x) + ((double)c.x + (double)d.x)) + (((double)e.x + (double)f.x) + ((double)g2.x + (double)h.x));
            s2 += (((double)a.y + (double)b.y) + ((double)c.y + (double)d.y)) + (((double)e.y + (double)f.y) + ((double)g2.y + (double)h.y));
        }
        for (; i + 3 * LPI < nblk; i += 4 * LPI) {
            const float2 a = p[i], b = p[i + LPI], c = p[i + 2 * LPI], d = p[i + 3 * LPI];
            s1 += ((double)a.x + (double)b.x) + ((double)c.x + (double)d.x);
            s2 += ((double)a.y + (double)b.y) + ((double)c.y + (double)d.y);
        }
        for (; i < nblk; i += LPI) { const float2 a = p[i]; s1 += (double)a.x; s2 += (double)a.y; }
#pragma unroll
        for (int o = LPI / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if (s2_sign && ok && gamma[g * cpg + j] < 0.f) s2 = -s2;       // fused conv statistics carry sum dh * sign(gamma) * xhat
        if (sl == 0 && ok) { S[item * 2] = s1; S[item * 2 + 1] = s2; }
    }
    __syncthreads();
    const double m = (double)cpg * (double)V;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        double m1 = 0.0, m2 = 0.0;
        for (int j = 0; j < cpg; ++j) {
            const double gm = (double)gamma[g * cpg + j];
            m1 += gm * S[(n * cpg + j) * 2];
            m2 += gm * S[(n * cpg + j) * 2 + 1];
        }
        m1 /= m; m2 /= m;
        const double mu = (double)mean[n * G + g], rs = (double)rstd[n * G + g];
        for (int j = 0; j < cpg; ++j) {
            const int c = g * cpg + j;
            float* q = coef + ((size_t)n * C + c) * 3;
            q[0] = (float)(rs * (double)gamma[c]);
            q[1] = (float)(-rs * rs * m2);
            q[2] = (float)(rs * rs * m2 * mu - rs * m1);
        }
    }
    for (int j = threadIdx.x; j < cpg; j += blockDim.x) {
        double dg = 0.0, db = 0.0;
        for (int n = 0; n < N; ++n) { db += S[(n * cpg + j) * 2]; dg += S[(n * cpg + j) * 2 + 1]; }
        if (dgamma) dgamma[g * cpg + j] = (float)dg;
        if (dbeta) dbeta[g * cpg + j] = (float)db;
    }
}
int gn_bwd_finalize_launch(const float* partials, int nblk, const float* gamma, const float* mean, const float* rstd,
                           float* coef, float* dgamma, float* dbeta, int N, int C, size_t V, int G, hipStream_t s, int s2_sign) {
    RU_REQUIRE(C % G == 0 && C / G <= 256, "groupnorm: C must be divisible by G (and C/G <= 256)");
    const size_t shm = (size_t)N * (C / G) * 2 * sizeof(double);
    RU_REQUIRE(shm <= 60000, "groupnorm backward: batch x channels-per-group too large for the finalize kernel");
    if (nblk > 32) hipLaunchKernelGGL(gn_bwd_finalize_kernel<64>, dim3(G), dim3(N * (C / G) > 4 ? 512 : 256), shm, s, partials, nblk, gamma, mean, rstd, coef, dgamma, dbeta, N, C, V, G, s2_sign);
    else if (nblk > 4) hipLaunchKernelGGL(gn_bwd_finalize_kernel<16>, dim3(G), dim3(256), shm, s, partials, nblk, gamma, mean, rstd, coef, dgamma, dbeta, N, C, V, G, s2_sign);
    else hipLaunchKernelGGL(gn_bwd_finalize_kernel<4>, dim3(G), dim3(256), shm, s, partials, nblk, gamma, mean, rstd, coef, dgamma, dbeta, N, C, V, G, s2_sign);
    RU_CHECK_LAUNCH("gn_bwd_finalize_kernel");
    return RU_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ coef, float slope,
                                                           float* __restrict__ dx, size_t V) {
    const size_t row = blockIdx.y;
    const float a = scale[row], b = shift[row];
    const float cA = coef[row * 3], cB = coef[row * 3 + 1], cC = coef[row * 3 + 2];
    const size_t base = row * V;
    for (size_t v = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; v < V; v += (size_t)gridDim.x * 256 * VEC) {
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + base + v);
            const float4 d = *reinterpret_cast<const float4*>(dy + base + v);
            float4 o;
            o.x = cA * ((t.x * a + b) > 0.f ? d.x : d.x * slope) + (cB * t.x + cC);
            o.y = cA * ((t.y * a + b) > 0.f ? d.y : d.y * slope) + (cB * t.y + cC);
            o.z = cA * ((t.z * a + b) > 0.f ? d.z : d.z * slope) + (cB * t.z + cC);
            o.w = cA * ((t.w * a + b) > 0.f ? d.w : d.w * slope) + (cB * t.w + cC);
            *reinterpret_cast<float4*>(dx + base + v) = o;
        } else {
            const float t = x[base + v], d = dy[base + v];
            dx[base + v] = cA * ((t * a + b) > 0.f ? d : d * slope) + (cB * t + cC);
        }
    }
}
int gn_bwd_apply_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* coef,
                        float slope, float* dx, int N, int C, size_t V, hipStream_t s) {
    if (V % 4 == 0) hipLaunchKernelGGL(gn_bwd_apply_kernel<4>, rows_grid(V, 4, N * C), dim3(256), 0, s, x, dy, scale, shift, coef, slope, dx, V);
    else hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, rows_grid(V, 1, N * C), dim3(256), 0, s, x, dy, scale, shift, coef, slope, dx, V);
    RU_CHECK_LAUNCH("gn_bwd_apply_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ flat elementwise
enum { EW_LRELU = 0, EW_LRELU_BWD = 1, EW_SIGMOID = 2, EW_SIGMOID_BWD = 3, EW_ADD = 4, EW_FILL = 5 };
template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n, float p) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float r;
        if (OP == EW_LRELU) r = lrelu(a[i], p);
        else if (OP == EW_LRELU_BWD) r = a[i] > 0.f ? b[i] : b[i] * p;            // a = OUTPUT of the LeakyReLU, b = dy
        else if (OP == EW_SIGMOID) r = 1.f / (1.f + expf(-a[i]));
        else if (OP == EW_SIGMOID_BWD) { const float q = a[i]; r = b[i] * q * (1.f - q); }   // a = p, b = dp
        else if (OP == EW_ADD) r = a[i] + b[i];
        else r = p;
        y[i] = r;
    }
}
template <int OP>
static int ew_launch(const float* a, const float* b, float* y, size_t n, float p, hipStream_t s, const char* name) {
    if (n == 0) return RU_OK;
    hipLaunchKernelGGL(ew_kernel<OP>, dim3(grid1d(n, 256 * 4, 8192)), dim3(256), 0, s, a, b, y, n, p);
    RU_CHECK_LAUNCH(name);
    return RU_OK;
}
int lrelu_fwd_launch(const float* x, float* y, size_t n, float slope, hipStream_t s) { return ew_launch<EW_LRELU>(x, nullptr, y, n, slope, s, "lrelu"); }
int lrelu_bwd_launch(const float* yv, const float* dy, float* dx, size_t n, float slope, hipStream_t s) { return ew_launch<EW_LRELU_BWD>(yv, dy, dx, n, slope, s, "lrelu_bwd"); }
int sigmoid_launch(const float* x, float* y, size_t n, hipStream_t s) { return ew_launch<EW_SIGMOID>(x, nullptr, y, n, 0.f, s, "sigmoid"); }
int sigmoid_bwd_launch(const float* p, const float* dp, float* dz, size_t n, hipStream_t s) { return ew_launch<EW_SIGMOID_BWD>(p, dp, dz, n, 0.f, s, "sigmoid_bwd"); }
int add_launch(const float* a, const float* b, float* y, size_t n, hipStream_t s) { return ew_launch<EW_ADD>(a, b, y, n, 0.f, s, "add"); }
// split-K partial tensors -> y, summed in z order (deterministic)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int ksplit, size_t n, float* __restrict__ y) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 acc = reinterpret_cast<const float4*>(part)[i];
        for (int z = 1; z < ksplit; ++z) {
            const float4 t = reinterpret_cast<const float4*>(part + (size_t)z * n)[i];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        reinterpret_cast<float4*>(y)[i] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                  // ragged tail
        const size_t i = (n4 << 2) + threadIdx.x;
        float acc = part[i];
        for (int z = 1; z < ksplit; ++z) acc += part[(size_t)z * n + i];
        y[i] = acc;
    }
}
int sum_partials_launch(const float* part, int ksplit, size_t n, float* y, hipStream_t s) {
    RU_REQUIRE(part && y && ksplit >= 1, "sum_partials: bad argument");
    RU_REQUIRE(ksplit == 1 || (n & 3) == 0, "sum_partials: partial tensors of %zu floats are not 16-byte aligned to each other", n);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(grid1d(n, 256 * 4, 2048)), dim3(256), 0, s, part, ksplit, n, y);
    RU_CHECK_LAUNCH("sum_partials_kernel");
    return RU_OK;
}
int fill_launch(float* p, float v, size_t n, hipStream_t s) { return ew_launch<EW_FILL>(nullptr, nullptr, p, n, v, s, "fill"); }

// ------------------------------------------------------------------ trilinear x2, align_corners=False (model.py:12-14)
__global__ __launch_bounds__(256) void up2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int D, int H, int W) {
    const int Do = 2 * D, Ho = 2 * H;
    const size_t total = (size_t)NC * Do * Ho * W;     // one thread per output x-PAIR (2k, 2k+1)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % W);
        size_t r = i / W;
        const int yo = (int)(r % Ho); r /= Ho;
        const int zo = (int)(r % Do);
        const size_t nc = r / Do;
        int z0, z1, y0, y1; float lz0, lz1, ly0, ly1;
        up2_src(zo, D, z0, z1, lz0, lz1);
        up2_src(yo, H, y0, y1, ly0, ly1);
        const int km = k > 0 ? k - 1 : 0, kp = k < W - 1 ? k + 1 : k;
        const float* p00 = x + ((nc * D + z0) * H + y0) * (size_t)W;
        const float* p01 = x + ((nc * D + z0) * H + y1) * (size_t)W;
        const float* p10 = x + ((nc * D + z1) * H + y0) * (size_t)W;
        const float* p11 = x + ((nc * D + z1) * H + y1) * (size_t)W;
        // even output 2k: src = k - 0.25 -> (k-1: 0.25, k: 0.75), except k == 0 -> (0: 1, 1: 0)
        // odd output 2k+1: src = k + 0.25 -> (k: 0.75, k+1 clamped: 0.25)
        const float e0 = k > 0 ? 0.25f : 1.f, e1 = k > 0 ? 0.75f : 0.f;
        const int ei0 = k > 0 ? km : 0, ei1 = k > 0 ? k : (W > 1 ? 1 : 0);
        const float ev = lz0 * (ly0 * (e0 * p00[ei0] + e1 * p00[ei1]) + ly1 * (e0 * p01[ei0] + e1 * p01[ei1])) +
                         lz1 * (ly0 * (e0 * p10[ei0] + e1 * p10[ei1]) + ly1 * (e0 * p11[ei0] + e1 * p11[ei1]));
        const float ov = lz0 * (ly0 * (0.75f * p00[k] + 0.25f * p00[kp]) + ly1 * (0.75f * p01[k] + 0.25f * p01[kp])) +
                         lz1 * (ly0 * (0.75f * p10[k] + 0.25f * p10[kp]) + ly1 * (0.75f * p11[k] + 0.25f * p11[kp]));
        float* yp = y + ((nc * Do + zo) * Ho + yo) * (size_t)(2 * W) + 2 * k;
        *reinterpret_cast<float2*>(yp) = make_float2(ev, ov);
    }
}
// W even: one thread per FOUR consecutive outputs (coarse k = 2m, 2m+1): 16 cached loads, one 16-byte store
__global__ __launch_bounds__(256) void up2_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int D, int H, int W) {
    const int Do = 2 * D, Ho = 2 * H, Wh = W / 2;
    const size_t total = (size_t)NC * Do * Ho * Wh;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i % Wh);
        size_t r = i / Wh;
        const int yo = (int)(r % Ho); r /= Ho;
        const int zo = (int)(r % Do);
        const size_t nc = r / Do;
        int z0, z1, y0, y1; float lz0, lz1, ly0, ly1;
        up2_src(zo, D, z0, z1, lz0, lz1);
        up2_src(yo, H, y0, y1, ly0, ly1);
        const int k = 2 * m;
        const int km = k > 0 ? k - 1 : 0, kpp = k + 2 < W ? k + 2 : W - 1;
        const float* rows[4] = {x + ((nc * D + z0) * H + y0) * (size_t)W, x + ((nc * D + z0) * H + y1) * (size_t)W,
                                x + ((nc * D + z1) * H + y0) * (size_t)W, x + ((nc * D + z1) * H + y1) * (size_t)W};
        const float wgt[4] = {lz0 * ly0, lz0 * ly1, lz1 * ly0, lz1 * ly1};
        // same nesting as the scalar kernel: z (y (x)))
        float o[4];
        float rowv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a0 = rows[q][km], a1 = rows[q][k], a2 = rows[q][k + 1], a3 = rows[q][kpp];
            rowv[q][0] = k > 0 ? 0.25f * a0 + 0.75f * a1 : 1.f * a1 + 0.f * a2;      // output 2k   (k == 0: x[0] with weight 1)
            rowv[q][1] = 0.75f * a1 + 0.25f * a2;                                     // output 2k+1
            rowv[q][2] = 0.25f * a1 + 0.75f * a2;                                     // output 2k+2 = 2(k+1)
            rowv[q][3] = 0.75f * a2 + 0.25f * a3;                                     // output 2k+3 (a3 clamped at the end)
        }
        (void)wgt;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            o[t] = lz0 * (ly0 * rowv[0][t] + ly1 * rowv[1][t]) + lz1 * (ly0 * rowv[2][t] + ly1 * rowv[3][t]);
        *reinterpret_cast<float4*>(y + ((nc * Do + zo) * Ho + yo) * (size_t)(2 * W) + 4 * m) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
int up2_fwd_launch(const float* x, float* y, int N, int C, int D, int H, int W, hipStream_t s) {
    if ((W & 1) == 0) {
        const size_t total = (size_t)N * C * 2 * D * 2 * H * (W / 2);
        hipLaunchKernelGGL(up2_fwd_vec_kernel, dim3(grid1d(total, 256)), dim3(256), 0, s, x, y, N * C, D, H, W);
        RU_CHECK_LAUNCH("up2_fwd_vec_kernel");
        return RU_OK;
    }
    const size_t total = (size_t)N * C * 2 * D * 2 * H * W;
    hipLaunchKernelGGL(up2_fwd_kernel, dim3(grid1d(total, 256)), dim3(256), 0, s, x, y, N * C, D, H, W);
    RU_CHECK_LAUNCH("up2_fwd_kernel");
    return RU_OK;
}

__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int D, int H, int W) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)NC * D * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int kx = (int)(i % W);
        size_t r = i / W;
        const int ky = (int)(r % H); r /= H;
        const int kz = (int)(r % D);
        const size_t nc = r / D;
        float wz[4], wy[4], wx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oz = 2 * kz - 1 + t, oy = 2 * ky - 1 + t, ox = 2 * kx - 1 + t;
            wz[t] = (oz >= 0 && oz < Do) ? up2_coef(oz, D, kz) : 0.f;
            wy[t] = (oy >= 0 && oy < Ho) ? up2_coef(oy, H, ky) : 0.f;
            wx[t] = (ox >= 0 && ox < Wo) ? up2_coef(ox, W, kx) : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (wz[a] == 0.f) continue;
            const int oz = 2 * kz - 1 + a;
            float accy = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (wy[b] == 0.f) continue;
                const int oy = 2 * ky - 1 + b;
                const float* p = dy + ((nc * Do + oz) * Ho + oy) * (size_t)Wo;
                float accx = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ox = 2 * kx - 1 + c;
                    if (wx[c] != 0.f) accx += wx[c] * p[ox];
                }
                accy += wy[b] * accx;
            }
            acc += wz[a] * accy;
        }
        dx[i] = acc;
    }
}
// W even: one thread per TWO coarse voxels (kx = 2m, 2m+1): per fine row one aligned float4 (ox 4m..4m+3) plus the two
// neighbours 4m-1 and 4m+4 instead of eight scalar loads
__global__ __launch_bounds__(256) void up2_bwd_vec_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int D, int H, int W) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W, Wh = W / 2;
    const size_t total = (size_t)NC * D * H * Wh;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i % Wh);
        size_t r = i / Wh;
        const int ky = (int)(r % H); r /= H;
        const int kz = (int)(r % D);
        const size_t nc = r / D;
        const int k0 = 2 * m, k1 = 2 * m + 1;
        float wz[4], wy[4], wa[4], wb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oz = 2 * kz - 1 + t, oy = 2 * ky - 1 + t, oa = 2 * k0 - 1 + t, ob = 2 * k1 - 1 + t;
            wz[t] = (oz >= 0 && oz < Do) ? up2_coef(oz, D, kz) : 0.f;
            wy[t] = (oy >= 0 && oy < Ho) ? up2_coef(oy, H, ky) : 0.f;
            wa[t] = (oa >= 0 && oa < Wo) ? up2_coef(oa, W, k0) : 0.f;
            wb[t] = (ob >= 0 && ob < Wo) ? up2_coef(ob, W, k1) : 0.f;
        }
        const int oxl = 4 * m - 1 >= 0 ? 4 * m - 1 : 0, oxr = 4 * m + 4 < Wo ? 4 * m + 4 : Wo - 1;   // clamped (their weights are 0 when outside)
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oz = 2 * kz - 1 + a;
            const int ozc = oz < 0 ? 0 : (oz >= Do ? Do - 1 : oz);
            float ay0 = 0.f, ay1 = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int oy = 2 * ky - 1 + b;
                const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy);
                const float* p = dy + ((nc * Do + ozc) * Ho + oyc) * (size_t)Wo;
                const float4 f = *reinterpret_cast<const float4*>(p + 4 * m);     // fine 4m .. 4m+3
                const float fl = p[oxl], fr = p[oxr];
                // k0 = 2m uses fine 4m-1 .. 4m+2 ; k1 = 2m+1 uses fine 4m+1 .. 4m+4
                ay0 += wy[b] * (wa[0] * fl + wa[1] * f.x + wa[2] * f.y + wa[3] * f.z);
                ay1 += wy[b] * (wb[0] * f.y + wb[1] * f.z + wb[2] * f.w + wb[3] * fr);
            }
            acc0 += wz[a] * ay0;
            acc1 += wz[a] * ay1;
        }
        *reinterpret_cast<float2*>(dx + ((nc * D + kz) * H + ky) * (size_t)W + 2 * m) = make_float2(acc0, acc1);
    }
}
int up2_bwd_launch(const float* dy, float* dx, int N, int C, int D, int H, int W, hipStream_t s) {
    if ((W & 1) == 0) {
        const size_t tv = (size_t)N * C * D * H * (W / 2);
        hipLaunchKernelGGL(up2_bwd_vec_kernel, dim3(grid1d(tv, 256)), dim3(256), 0, s, dy, dx, N * C, D, H, W);
        RU_CHECK_LAUNCH("up2_bwd_vec_kernel");
        return RU_OK;
    }
    const size_t total = (size_t)N * C * D * H * W;
    hipLaunchKernelGGL(up2_bwd_kernel, dim3(grid1d(total, 256)), dim3(256), 0, s, dy, dx, N * C, D, H, W);
    RU_CHECK_LAUNCH("up2_bwd_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ bias gradient (conv_output.bias, model.py:348)
constexpr int BG_CHUNK = 16384;
size_t bias_grad_workspace_bytes(int N, int C, size_t V) { return (size_t)N * C * ((V + BG_CHUNK - 1) / BG_CHUNK) * sizeof(float); }
__global__ __launch_bounds__(256) void bias_partial_kernel(const float* __restrict__ dy, float* __restrict__ part, size_t V, int nblk) {
    __shared__ float buf[4];
    const size_t row = blockIdx.y;
    const size_t v0 = (size_t)blockIdx.x * BG_CHUNK;
    const size_t v1 = v0 + BG_CHUNK < V ? v0 + BG_CHUNK : V;
    float s1 = 0.f;
    for (size_t v = v0 + threadIdx.x; v < v1; v += 256) s1 += dy[row * V + v];
    s1 = block_sum(s1, buf);
    if (threadIdx.x == 0) part[row * nblk + blockIdx.x] = s1;
}
__global__ __launch_bounds__(256) void bias_final_kernel(const float* __restrict__ part, float* __restrict__ db, int N, int C, int nblk) {
    __shared__ double buf[4];
    const int c = blockIdx.x;
    double s1 = 0.0;
    for (int i = threadIdx.x; i < N * nblk; i += 256) {
        const int n = i / nblk, b = i % nblk;
        s1 += (double)part[((size_t)n * C + c) * nblk + b];
    }
    s1 = block_sum_d(s1, buf);
    if (threadIdx.x == 0) db[c] = (float)s1;
}
int bias_grad_launch(const float* dy, float* db, int N, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < bias_grad_workspace_bytes(N, C, V)) { set_error("bias_grad: workspace too small"); return RU_ENOMEM; }
    const int nblk = (int)((V + BG_CHUNK - 1) / BG_CHUNK);
    hipLaunchKernelGGL(bias_partial_kernel, dim3(nblk, N * C), dim3(256), 0, s, dy, (float*)ws, V, nblk);
    RU_CHECK_LAUNCH("bias_partial_kernel");
    hipLaunchKernelGGL(bias_final_kernel, dim3(C), dim3(256), 0, s, (const float*)ws, db, N, C, nblk);
    RU_CHECK_LAUNCH("bias_final_kernel");
    return RU_OK;
}

// Head of the backward pass on the 4-channel path, one pass instead of three (sigmoid backward, 4-channel copy, bias partials):
// dz = dp*p*(1-p) (model.py:431) goes straight into the zero-padded voxel-major copy [N][V][4] that the head's weight- and
// data-gradient kernels read, and the bias gradient's partial sums (model.py:348) are taken on the way.  V % 4 == 0.
// CRIT: `dp` is the TARGET tensor and the incoming gradient is the criterion's (crit_grad_kernel's expression, same float operations in
// the same order -> the same bits), formed from (p, target) and the per-class constants of `cg`
template <bool CRIT>
__global__ __launch_bounds__(256) void head_grad_c4_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ d4,
                                                           float* __restrict__ part, int C, size_t V, int nblk, const CritGradArgs cg) {
    __shared__ float buf[4];
    const size_t n = blockIdx.y;
    const size_t v0 = (size_t)blockIdx.x * BG_CHUNK;
    const size_t v1 = v0 + BG_CHUNK < V ? v0 + BG_CHUNK : V;
    const float* pp = p + n * C * V;
    const float* dpp = dp + n * C * V;
    float4* op = reinterpret_cast<float4*>(d4) + n * V;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    float kg[4] = {0.f, 0.f, 0.f, 0.f}, kp[4] = {0.f, 0.f, 0.f, 0.f}, cb = 0.f;
    const float one_eps = (float)(1.0 + 1e-6);
    if constexpr (CRIT) {
        for (int c = 0; c < C && c < 4; ++c) {           // crit_grad_kernel's constants
            const double I = cg.sums[c] + 1e-6, U = cg.sums[C + c] + 2e-6;
            const double k = (double)cg.priority * (2.0 / (double)C) * (double)cg.w_dice;
            kg[c] = (float)(-k / U);
            kp[c] = (float)(2.0 * k * I / (U * U));
        }
        cb = (float)((double)cg.w_bce / cg.count);
    }
    for (size_t v = v0 + (size_t)threadIdx.x * 4; v < v1; v += 1024) {
        float d[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = c < C;
            const float4 q = *reinterpret_cast<const float4*>(pp + (ok ? c : 0) * V + v);
            float4 g = *reinterpret_cast<const float4*>(dpp + (ok ? c : 0) * V + v);
            if constexpr (CRIT) {
                const float pv[4] = {q.x, q.y, q.z, q.w}, gv[4] = {g.x, g.y, g.z, g.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bce = -(gv[e] / (pv[e] + 1e-6f) - cg.bgw * (1.f - gv[e]) / (one_eps - pv[e]));
                    o[e] = (kg[c] * gv[e] + kp[c] * pv[e]) + cb * bce;
                }
                g = make_float4(o[0], o[1], o[2], o[3]);
            }
            d[c][0] = ok ? g.x * q.x * (1.f - q.x) : 0.f;
            d[c][1] = ok ? g.y * q.y * (1.f - q.y) : 0.f;
            d[c][2] = ok ? g.z * q.z * (1.f - q.z) : 0.f;
            d[c][3] = ok ? g.w * q.w * (1.f - q.w) : 0.f;
            sum[c] += (d[c][0] + d[c][1]) + (d[c][2] + d[c][3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) op[v + e] = make_float4(d[0][e], d[1][e], d[2][e], d[3][e]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float t = block_sum(sum[c], buf);
        if (threadIdx.x == 0 && c < C) part[(n * C + c) * nblk + blockIdx.x] = t;
    }
}
int head_grad_c4_launch(const float* p, const float* dp, float* d4, float* db, int N, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s) {
    RU_REQUIRE(C > 0 && C <= 4 && V % 4 == 0, "head_grad_c4: 1..4 channels, V % 4 == 0");
    if (!ws || ws_bytes < bias_grad_workspace_bytes(N, C, V)) { set_error("head_grad_c4: workspace too small"); return RU_ENOMEM; }
    const int nblk = (int)((V + BG_CHUNK - 1) / BG_CHUNK);
    hipLaunchKernelGGL(head_grad_c4_kernel<false>, dim3(nblk, N), dim3(256), 0, s, p, dp, d4, (float*)ws, C, V, nblk, CritGradArgs{});
    RU_CHECK_LAUNCH("head_grad_c4_kernel");
    hipLaunchKernelGGL(bias_final_kernel, dim3(C), dim3(256), 0, s, (const float*)ws, db, N, C, nblk);
    RU_CHECK_LAUNCH("bias_final_kernel");
    return RU_OK;
}
int head_grad_c4_crit_launch(const float* p, const CritGradArgs& cg, float* d4, float* db, int N, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s) {
    RU_REQUIRE(C > 0 && C <= 4 && V % 4 == 0 && cg.target && cg.sums, "head_grad_c4_crit: 1..4 channels, V % 4 == 0, target and sums");
    if (!ws || ws_bytes < bias_grad_workspace_bytes(N, C, V)) { set_error("head_grad_c4_crit: workspace too small"); return RU_ENOMEM; }
    const int nblk = (int)((V + BG_CHUNK - 1) / BG_CHUNK);
    hipLaunchKernelGGL(head_grad_c4_kernel<true>, dim3(nblk, N), dim3(256), 0, s, p, cg.target, d4, (float*)ws, C, V, nblk, cg);
    RU_CHECK_LAUNCH("head_grad_c4_kernel<crit>");
    hipLaunchKernelGGL(bias_final_kernel, dim3(C), dim3(256), 0, s, (const float*)ws, db, N, C, nblk);
    RU_CHECK_LAUNCH("bias_final_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ criterion: Dice_loss_joint + BCE_Loss (loss.py:64-79,98-122)
constexpr int CR_CHUNK = 8192;
int crit_tiles(size_t V) { return (int)((V + CR_CHUNK - 1) / CR_CHUNK); }
__global__ __launch_bounds__(256) void crit_partial_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ part,
                                                           size_t V, int nblk, float bgw) {
    __shared__ float buf[4];
    const size_t row = blockIdx.y;             // n*C + c
    const size_t v0 = (size_t)blockIdx.x * CR_CHUNK;
    const size_t v1 = v0 + CR_CHUNK < V ? v0 + CR_CHUNK : V;
    const float one_eps = (float)(1.0 + 1e-6);   // the float32 value of loss.py:77's (1.+1e-6)
    float si = 0.f, su = 0.f, sb = 0.f;
    for (size_t v = v0 + threadIdx.x; v < v1; v += 256) {
        const float pv = p[row * V + v], gv = g[row * V + v];
        si += pv * gv;
        su += pv * pv + gv;
        sb += gv * logf(pv + 1e-6f) + bgw * (1.f - gv) * logf(one_eps - pv);
    }
    si = block_sum(si, buf);
    su = block_sum(su, buf);
    sb = block_sum(sb, buf);
    if (threadIdx.x == 0) {
        float* q = part + (row * nblk + blockIdx.x) * 3;
        q[0] = si; q[1] = su; q[2] = sb;
    }
}
__global__ __launch_bounds__(256) void crit_final_kernel(const float* __restrict__ part, double* __restrict__ sums, int N, int C, int nblk) {
    __shared__ double buf[4];
    double bce = 0.0;
    for (int c = 0; c < C; ++c) {
        double si = 0.0, su = 0.0, sb = 0.0;
        for (int i = threadIdx.x; i < N * nblk; i += 256) {
            const int n = i / nblk, b = i % nblk;
            const float* q = part + (((size_t)n * C + c) * nblk + b) * 3;
            si += (double)q[0]; su += (double)q[1]; sb += (double)q[2];
        }
        si = block_sum_d(si, buf);
        su = block_sum_d(su, buf);
        sb = block_sum_d(sb, buf);
        if (threadIdx.x == 0) { sums[c] = si; sums[C + c] = su; }
        bce += sb;
    }
    if (threadIdx.x == 0) sums[2 * C] = bce;
}
int crit_sums_launch(const float* p, const float* g, double* sums, int N, int C, size_t V, float bgw, void* ws, size_t ws_bytes, hipStream_t s) {
    const int nblk = crit_tiles(V);
    if (!ws || ws_bytes < (size_t)N * C * nblk * 3 * sizeof(float)) { set_error("criterion: workspace too small"); return RU_ENOMEM; }
    hipLaunchKernelGGL(crit_partial_kernel, dim3(nblk, N * C), dim3(256), 0, s, p, g, (float*)ws, V, nblk, bgw);
    RU_CHECK_LAUNCH("crit_partial_kernel");
    hipLaunchKernelGGL(crit_final_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, sums, N, C, nblk);
    RU_CHECK_LAUNCH("crit_final_kernel");
    return RU_OK;
}
// loss scalars on the device (one wave): out = (w_dice*dice + w_bce*bce, dice, bce), loss.py:114-122,79
__global__ void crit_value_kernel(const double* __restrict__ sums, int C, double count, double priority, double w_dice, double w_bce,
                                  double* __restrict__ out) {
    double q = threadIdx.x < (unsigned)C ? 2.0 * (sums[threadIdx.x] + 1e-6) / (sums[C + threadIdx.x] + 2e-6) : 0.0;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (threadIdx.x == 0) {
        const double dice = priority * (1.0 - q / C), bce = -sums[2 * C] / count;
        out[0] = w_dice * dice + w_bce * bce; out[1] = dice; out[2] = bce;
    }
}
int crit_value_launch(const double* sums, int C, double count, double priority, double w_dice, double w_bce, double* out3, hipStream_t s) {
    hipLaunchKernelGGL(crit_value_kernel, dim3(1), dim3(64), 0, s, sums, C, count, priority, w_dice, w_bce, out3);
    RU_CHECK_LAUNCH("crit_value_kernel");
    return RU_OK;
}
__global__ __launch_bounds__(256) void crit_grad_kernel(const float* __restrict__ p, const float* __restrict__ g, const double* __restrict__ sums,
                                                        double count, float w_dice, float w_bce, float bgw, float priority,
                                                        float* __restrict__ dp, int C, size_t V) {
    const size_t row = blockIdx.y;
    const int c = (int)(row % C);
    const double I = sums[c] + 1e-6, U = sums[C + c] + 2e-6;
    const double k = (double)priority * (2.0 / (double)C) * (double)w_dice;
    const float cg = (float)(-k / U), cp = (float)(2.0 * k * I / (U * U));
    const float cb = (float)((double)w_bce / count);
    const float one_eps = (float)(1.0 + 1e-6);
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const float pv = p[row * V + v], gv = g[row * V + v];
        const float bce = -(gv / (pv + 1e-6f) - bgw * (1.f - gv) / (one_eps - pv));
        dp[row * V + v] = (cg * gv + cp * pv) + cb * bce;
    }
}
int crit_grad_launch(const float* p, const float* g, const double* sums, double count, float w_dice, float w_bce,
                     float bgw, float priority, float* dp, int N, int C, size_t V, hipStream_t s) {
    hipLaunchKernelGGL(crit_grad_kernel, rows_grid(V, 1, N * C), dim3(256), 0, s, p, g, sums, count, w_dice, w_bce, bgw, priority, dp, C, V);
    RU_CHECK_LAUNCH("crit_grad_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ inference post-processing (test.py:134-159)
// TTA merge: K predictions p[k] of flipped inputs (flip mask bit0 = D, bit1 = H, bit2 = W reversed) are un-flipped and
// averaged in the reference's order ((p0 + p1) + p2 + ...) / K (float32, bit-exact vs numpy), thresholded at 0.5 and counted.
__global__ __launch_bounds__(256) void tta_merge_kernel(const float* __restrict__ p, int K, unsigned flips, float* __restrict__ mean_out,
                                                        unsigned char* __restrict__ mask_out, unsigned long long* __restrict__ counts,
                                                        int C, int D, int H, int W) {
    __shared__ unsigned int cnt[4];
    const size_t V = (size_t)D * H * W;
    const int c = blockIdx.y;
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned int local = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W);
        const int y = (int)((v / W) % H);
        const int z = (int)(v / ((size_t)W * H));
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const unsigned f = (flips >> (3 * k)) & 7u;
            const int zz = (f & 1u) ? D - 1 - z : z, yy = (f & 2u) ? H - 1 - y : y, xx = (f & 4u) ? W - 1 - x : x;
            const float t = p[(((size_t)k * C + c) * D + zz) * H * W + (size_t)yy * W + xx];
            acc = k == 0 ? t : acc + t;
        }
        const float m = acc / (float)K;
        if (mean_out) mean_out[(size_t)c * V + v] = m;
        const bool on = m > 0.5f;
        mask_out[(size_t)c * V + v] = on ? 1 : 0;
        local += on ? 1u : 0u;
    }
    atomicAdd(&cnt[threadIdx.x >> 6], local);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&counts[c], (unsigned long long)cnt[0] + cnt[1] + cnt[2] + cnt[3]);   // integer: order independent
}
// labels: 2 where WT, then 1 where TC, then 4 where ET if the ET count exceeds `et_min` (test.py:153-159)
__global__ __launch_bounds__(256) void compose_labels_kernel(const unsigned char* __restrict__ mask, const unsigned long long* __restrict__ counts,
                                                             unsigned long long et_min, unsigned char* __restrict__ labels, size_t V) {
    const bool et_on = counts[2] > et_min;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        unsigned char l = 0;
        if (mask[v]) l = 2;
        if (mask[V + v]) l = 1;
        if (et_on && mask[2 * V + v]) l = 4;
        labels[v] = l;
    }
}
int tta_merge_launch(const float* p, int K, unsigned flips, float* mean_out, unsigned char* mask, unsigned long long* counts, int C, int D, int H, int W, hipStream_t s) {
    RU_REQUIRE(K >= 1 && K <= 8 && C >= 1, "tta_merge: 1..8 predictions");
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(unsigned long long) * C, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(counts)");
    const size_t V = (size_t)D * H * W;
    hipLaunchKernelGGL(tta_merge_kernel, dim3(grid1d(V, 256 * 8, 512), C), dim3(256), 0, s, p, K, flips, mean_out, mask, counts, C, D, H, W);
    RU_CHECK_LAUNCH("tta_merge_kernel");
    return RU_OK;
}
int compose_labels_launch(const unsigned char* mask, const unsigned long long* counts, unsigned long long et_min, unsigned char* labels, size_t V, hipStream_t s) {
    hipLaunchKernelGGL(compose_labels_kernel, dim3(grid1d(V, 256, 4096)), dim3(256), 0, s, mask, counts, et_min, labels, V);
    RU_CHECK_LAUNCH("compose_labels_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ layout change NCDHW <-> C16
// one workgroup = 64 voxels x 16 channels through LDS; both global sides coalesced
__global__ __launch_bounds__(256) void layout_convert_kernel(const float* __restrict__ src, float* __restrict__ dst, int CB, size_t V, int to_c16) {
    __shared__ float tile[16][65];
    const size_t v0 = (size_t)blockIdx.x * 64;
    const size_t nb = blockIdx.y;                          // n * CB + cb
    const int t = threadIdx.x;
    const float* s_nc = src + nb * 16 * V;                 // both layouts: 16 * V floats per (n, cb)
    float* d_nc = dst + nb * 16 * V;
    if (to_c16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (t >> 6) + 4 * j, v = t & 63;
            tile[c][v] = v0 + v < V ? s_nc[(size_t)c * V + v0 + v] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = t + 256 * j, v = e >> 4, c = e & 15;
            if (v0 + v < V) d_nc[(v0 + v) * 16 + c] = tile[c][v];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = t + 256 * j, v = e >> 4, c = e & 15;
            tile[c][v] = v0 + v < V ? s_nc[(v0 + v) * 16 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (t >> 6) + 4 * j, v = t & 63;
            if (v0 + v < V) d_nc[(size_t)c * V + v0 + v] = tile[c][v];
        }
    }
}
int layout_convert_launch(const float* src, float* dst, int N, int C, size_t V, int to_c16, hipStream_t s) {
    RU_REQUIRE(C > 0 && C % 16 == 0, "layout_convert: C must be a multiple of 16");
    hipLaunchKernelGGL(layout_convert_kernel, dim3((unsigned)((V + 63) / 64), (unsigned)(N * (C / 16))), dim3(256), 0, s, src, dst, C / 16, V, to_c16);
    RU_CHECK_LAUNCH("layout_convert_kernel");
    return RU_OK;
}

// few-channel NCDHW tensor (network input: 4 modalities; head gradient: 3 classes) -> one zero-padded voxel-major block, so the
// transpose-read weight-gradient kernel can take it
__global__ __launch_bounds__(256) void pad_to_c16_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, size_t V) {
    const size_t n = blockIdx.y;
    const float* sp = src + n * C * V;
    float4* dp = reinterpret_cast<float4*>(dst + n * 16 * V);
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < V * 4; f += (size_t)gridDim.x * 256) {
        const size_t v = f >> 2;
        const int c0 = (int)(f & 3) * 4;
        float4 o;
        o.x = c0 + 0 < C ? sp[(size_t)(c0 + 0) * V + v] : 0.f;
        o.y = c0 + 1 < C ? sp[(size_t)(c0 + 1) * V + v] : 0.f;
        o.z = c0 + 2 < C ? sp[(size_t)(c0 + 2) * V + v] : 0.f;
        o.w = c0 + 3 < C ? sp[(size_t)(c0 + 3) * V + v] : 0.f;
        dp[f] = o;
    }
}
int pad_to_c16_launch(const float* src, float* dst, int N, int C, size_t V, hipStream_t s) {
    RU_REQUIRE(C > 0 && C <= 16, "pad_to_c16: 1..16 channels");
    hipLaunchKernelGGL(pad_to_c16_kernel, dim3(grid1d(V * 4, 256, 4096), (unsigned)N), dim3(256), 0, s, src, dst, C, V);
    RU_CHECK_LAUNCH("pad_to_c16_kernel");
    return RU_OK;
}

__global__ __launch_bounds__(256) void pad_to_c4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, size_t V) {
    const size_t n = blockIdx.y;
    const float* sp = src + n * C * V;
    float4* dp = reinterpret_cast<float4*>(dst) + n * V;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        float4 o;
        o.x = sp[v];
        o.y = C > 1 ? sp[V + v] : 0.f;
        o.z = C > 2 ? sp[2 * V + v] : 0.f;
        o.w = C > 3 ? sp[3 * V + v] : 0.f;
        dp[v] = o;
    }
}
int pad_to_c4_launch(const float* src, float* dst, int N, int C, size_t V, hipStream_t s) {
    RU_REQUIRE(C > 0 && C <= 4, "pad_to_c4: 1..4 channels");
    hipLaunchKernelGGL(pad_to_c4_kernel, dim3(grid1d(V, 256, 4096), (unsigned)N), dim3(256), 0, s, src, dst, C, V);
    RU_CHECK_LAUNCH("pad_to_c4_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ evaluation metric (metrics.py:108-133)
// per (sample, channel): counts[row] = { sum(p>0.5 & g>0.5), sum(p>0.5) + sum(g>0.5) } as integers (order independent)
// ONE atomic pair per workgroup and at most 64 workgroups per row: the 2 x rows counters share two cache lines, and same-line atomics
// serialise at ~12 ns each -- the first version (one pair per WAVE of a 1024 x rows grid: 98 000 atomics) took 1.0 ms per call, 6 % of a
// training step through Trainer (metrics.Dice is updated every iteration, train.py:224-225)
__global__ __launch_bounds__(256) void dice_counts_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                          unsigned long long* __restrict__ counts, size_t V) {
    __shared__ unsigned int sm[4][2];
    const size_t row = blockIdx.y;
    const float* pp = p + row * V;
    const float* gp = g + row * V;
    unsigned int inter = 0, uni = 0;
    if ((V & 3) == 0) {
        const size_t V4 = V >> 2;
        for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V4; v += (size_t)gridDim.x * 256) {
            const float4 a = reinterpret_cast<const float4*>(pp)[v], b = reinterpret_cast<const float4*>(gp)[v];
            const bool a0 = a.x > 0.5f, a1 = a.y > 0.5f, a2 = a.z > 0.5f, a3 = a.w > 0.5f;
            const bool b0 = b.x > 0.5f, b1 = b.y > 0.5f, b2 = b.z > 0.5f, b3 = b.w > 0.5f;
            inter += (unsigned)(a0 && b0) + (unsigned)(a1 && b1) + (unsigned)(a2 && b2) + (unsigned)(a3 && b3);
            uni += (unsigned)a0 + (unsigned)a1 + (unsigned)a2 + (unsigned)a3 + (unsigned)b0 + (unsigned)b1 + (unsigned)b2 + (unsigned)b3;
        }
    } else {
        for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
            const bool a = pp[v] > 0.5f, b = gp[v] > 0.5f;
            inter += (a && b) ? 1u : 0u;
            uni += (a ? 1u : 0u) + (b ? 1u : 0u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { inter += __shfl_xor(inter, o); uni += __shfl_xor(uni, o); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = inter; sm[threadIdx.x >> 6][1] = uni; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&counts[row * 2], (unsigned long long)sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0]);
        atomicAdd(&counts[row * 2 + 1], (unsigned long long)sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1]);
    }
}
int dice_counts_launch(const float* p, const float* g, unsigned long long* counts, int rows, size_t V, hipStream_t s) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(unsigned long long) * 2 * rows, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dice counts)");
    hipLaunchKernelGGL(dice_counts_kernel, dim3(grid1d(V, 256 * 4 * 8, 64), rows), dim3(256), 0, s, p, g, counts, V);
    RU_CHECK_LAUNCH("dice_counts_kernel");
    return RU_OK;
}

// metrics.py:124-130 on the device: r = 2*num/den in float32 (exact integer sums below 2^24 like the reference's float32 sums), NaN -> 1,
// batch mean and accumulator in float64 -- one thread per class, samples in order
__global__ void dice_accumulate_kernel(const unsigned long long* __restrict__ counts, double* __restrict__ acc, int N, int C, int nacc) {
    const int c = threadIdx.x;
    if (c >= nacc) return;
    double sum = 0.0;
    for (int n = 0; n < N; ++n) {
        const float num = (float)counts[((size_t)n * C + c) * 2], den = (float)counts[((size_t)n * C + c) * 2 + 1];
        float r = 2.f * num / den;
        if (r != r) r = 1.f;
        sum += (double)r;
    }
    acc[c] += sum / (double)N;
}
int dice_accumulate_launch(const unsigned long long* counts, double* acc, int N, int C, int nacc, hipStream_t s) {
    RU_REQUIRE(counts && acc && N > 0 && nacc > 0 && nacc <= C && nacc <= 64, "dice_accumulate: bad argument");
    hipLaunchKernelGGL(dice_accumulate_kernel, dim3(1), dim3(64), 0, s, counts, acc, N, C, nacc);
    RU_CHECK_LAUNCH("dice_accumulate_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ training input pipeline (dataloader.py:100-216)
// z-score statistics: per channel count(x > 0), sum x, sum x^2 over ALL voxels (dataloader.py:124-130), float64, two stages
constexpr int ZS_CHUNK = 16384;
size_t zscore_workspace_bytes(int C, size_t V) { return (size_t)C * ((V + ZS_CHUNK - 1) / ZS_CHUNK) * 3 * sizeof(double); }
__global__ __launch_bounds__(256) void zscore_partial_kernel(const float* __restrict__ x, double* __restrict__ part, size_t V, int nblk) {
    __shared__ double buf[4];
    const size_t c = blockIdx.y;
    const size_t v0 = (size_t)blockIdx.x * ZS_CHUNK, v1 = v0 + ZS_CHUNK < V ? v0 + ZS_CHUNK : V;
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t v = v0 + threadIdx.x; v < v1; v += 256) {
        const double t = (double)x[c * V + v];
        n += t > 0.0 ? 1.0 : 0.0;
        s1 += t;
        s2 += t * t;
    }
    n = block_sum_d(n, buf);
    s1 = block_sum_d(s1, buf);
    s2 = block_sum_d(s2, buf);
    if (threadIdx.x == 0) { double* p = part + (c * nblk + blockIdx.x) * 3; p[0] = n; p[1] = s1; p[2] = s2; }
}
__global__ void zscore_final_kernel(const double* __restrict__ part, double* __restrict__ stats, int nblk) {
    const int c = blockIdx.x, j = threadIdx.x;                    // 3 threads: count, sum, sumsq; fixed order
    if (j >= 3) return;
    double a = 0.0;
    for (int i = 0; i < nblk; ++i) a += part[((size_t)c * nblk + i) * 3 + j];
    stats[c * 3 + j] = a;
}
int zscore_stats_launch(const float* x, double* stats, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s) {
    RU_REQUIRE(ws && ws_bytes >= zscore_workspace_bytes(C, V), "zscore: workspace too small");
    const int nblk = (int)((V + ZS_CHUNK - 1) / ZS_CHUNK);
    hipLaunchKernelGGL(zscore_partial_kernel, dim3(nblk, C), dim3(256), 0, s, x, (double*)ws, V, nblk);
    RU_CHECK_LAUNCH("zscore_partial_kernel");
    hipLaunchKernelGGL(zscore_final_kernel, dim3(C), dim3(64), 0, s, (const double*)ws, stats, nblk);
    RU_CHECK_LAUNCH("zscore_final_kernel");
    return RU_OK;
}

// One fused pass per patch (dataloader.py:147-205): crop -> affine zoom (scipy affine_transform, diagonal matrix, order 1, mode
// 'reflect' = linear interpolation on the half-sample-symmetric extension of the CROP, source coordinate = scale * output index)
// of the z-scored modalities and of the one-hot label -> flips of D, H, W -> optional D<->H transpose -> per-channel gain / bias
// -> WT / TC / ET soft targets.  One thread per output voxel: the 8 corner offsets and weights are shared by all channels.
__device__ __forceinline__ int reflect_idx(int i, int n) {
    int m = i % (2 * n);
    if (m < 0) m += 2 * n;
    return m < n ? m : 2 * n - 1 - m;
}
__global__ __launch_bounds__(256) void augment_patch_kernel(const AugmentArgs a) {
    const int P0 = a.P[0], P1 = a.P[1], P2 = a.P[2];
    const bool tr = (a.flags & 8) != 0;
    const int Q0 = tr ? P1 : P0, Q1 = tr ? P0 : P1;                // output extents
    const size_t total = (size_t)Q0 * Q1 * P2;
    const size_t HW = (size_t)a.H * a.W, DHW = (size_t)a.D * HW;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int k = (int)(o % P2);
        const size_t r = o / P2;
        const int j = (int)(r % Q1), i = (int)(r / Q1);
        const int pa = tr ? j : i, pb = tr ? i : j;                // indices before the transpose
        const int p[3] = {(a.flags & 1) ? P0 - 1 - pa : pa, (a.flags & 2) ? P1 - 1 - pb : pb, (a.flags & 4) ? P2 - 1 - k : k};
        size_t off[3][2];
        float wgt[3][2];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double x = (double)p[ax] * a.scale[ax];
            const double f = floor(x);
            const int i0 = (int)f;
            const float t = (float)(x - f);
            const size_t stride = ax == 0 ? HW : (ax == 1 ? (size_t)a.W : 1);
            off[ax][0] = (size_t)(reflect_idx(i0, a.P[ax]) + a.lo[ax]) * stride;
            off[ax][1] = (size_t)(reflect_idx(i0 + 1, a.P[ax]) + a.lo[ax]) * stride;
            wgt[ax][0] = 1.f - t;
            wgt[ax][1] = t;
        }
        float acc[RU_AUG_MAXC];
#pragma unroll
        for (int c = 0; c < RU_AUG_MAXC; ++c) acc[c] = 0.f;
        float cw1 = 0.f, cw2 = 0.f, cw3 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int qa = q >> 2, qb = (q >> 1) & 1, qc = q & 1;
            const float w = wgt[0][qa] * wgt[1][qb] * wgt[2][qc];
            const size_t v = off[0][qa] + off[1][qb] + off[2][qc];
#pragma unroll
            for (int c = 0; c < RU_AUG_MAXC; ++c)
                if (c < a.C) acc[c] += w * a.image[(size_t)c * DHW + v];
            const int l = a.label[v];
            cw1 += l == 1 ? w : 0.f;
            cw2 += l == 2 ? w : 0.f;
            cw3 += l == 3 ? w : 0.f;
        }
#pragma unroll
        for (int c = 0; c < RU_AUG_MAXC; ++c)
            if (c < a.C) a.data[(size_t)c * total + o] = ((acc[c] - a.mean[c]) * a.istd[c]) * a.gain[c] + a.bias[c];
        a.target[o] = (cw1 + cw2) + cw3;                           // WT = 1 + 2 + 3
        a.target[total + o] = cw1 + cw3;                           // TC = 1 + 3
        a.target[2 * total + o] = cw3;                             // ET = 3
    }
}
int augment_patch_launch(const AugmentArgs& a, hipStream_t s) {
    const size_t total = (size_t)a.P[0] * a.P[1] * a.P[2];
    hipLaunchKernelGGL(augment_patch_kernel, dim3(grid1d(total, 256, 4096)), dim3(256), 0, s, a);
    RU_CHECK_LAUNCH("augment_patch_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ Adam(amsgrad=True, weight_decay) (main.py:133-137)
// AMS = false: plain Adam (torch.optim.Adam(amsgrad=False)), vmax is not touched
template <bool AMS>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, size_t n, float step_size, float b1, float b2, float eps, float wd,
                                                   float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float wi = w[i];
        const float gi = g[i] + wd * wi;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        float vm = vi;
        if (AMS) { vm = fmaxf(vmax[i], vi); vmax[i] = vm; }
        const float denom = sqrtf(vm) / bc2_sqrt + eps;
        m[i] = mi; v[i] = vi;
        w[i] = wi - step_size * (mi / denom);
    }
}
int adam_launch(float* w, const float* g, float* m, float* v, float* vmax, size_t n, float lr, float b1, float b2,
                float eps, float wd, int step, hipStream_t s) {
    if (n == 0) return RU_OK;
    RU_REQUIRE(step >= 1, "adam: step is 1-based");
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    if (vmax) hipLaunchKernelGGL(adam_kernel<true>, dim3(grid1d(n, 256 * 4, 4096)), dim3(256), 0, s, w, g, m, v, vmax, n, (float)((double)lr / bc1), b1, b2, eps, wd,
                                 (float)sqrt(bc2));
    else hipLaunchKernelGGL(adam_kernel<false>, dim3(grid1d(n, 256 * 4, 4096)), dim3(256), 0, s, w, g, m, v, vmax, n, (float)((double)lr / bc1), b1, b2, eps, wd,
                            (float)sqrt(bc2));
    RU_CHECK_LAUNCH("adam_kernel");
    return RU_OK;
}

}  // namespace ru
