// wgrad_f32.hip -- weight gradients of the convolutions (autograd of model.py:72-73,336,348,361-363,393,401;
// SURVEY Appendix A1/A2) on v_mfma_f32_16x16x4_f32.
//
//   dw[o][c][tap] = sum_{n,voxel} dy[n][o][voxel] * xin[n][c][voxel + tap]
//
// GEMM view: M = output channels o (16 per tile), N = input channels c (16 per tile), K = voxels, 4 per MFMA
// (4 consecutive x positions).  A fragment lane l: dy_lds[o = l&15][pos + (l>>4)], B fragment lane l:
// x_lds[c = l&15][pos' + (l>>4)].  LDS rows (one per channel) have a stride == 2 (mod 4) words, which makes the
// 16 channels x 2 positions a half-wave reads land on 32 distinct banks.
//
// 3x3x3 kernel: a workgroup owns a spatial tile (TZ x TY x 16 voxels + halo), its 4 waves split the 27 taps
// (7,7,7,6), each wave keeps its taps' accumulators in registers across ALL tiles it walks (persistent over
// tiles), then writes one partial per workgroup; a second kernel reduces the partials in a fixed order
// (deterministic, no atomics).  1x1x1 kernel: waves split the voxels of a 256-voxel chunk instead.
#include "ru_common.h"

namespace ru {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int pad2mod4(int v) { return v + ((2 - (v & 3)) & 3); }   // smallest s >= v with s % 4 == 2

template <int TZ, int TY, int OT, int CT>
struct W3 {
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = 18;
    static constexpr int HVOL = HZ * HY * HX, TVOL = TZ * TY * 16;
    static constexpr int DS = pad2mod4(TVOL), CS = pad2mod4(HVOL);
    static constexpr int LDS_FLOATS = OT * 16 * DS + CT * 16 * CS;
};

template <int TZ, int TY, int OT, int CT>
__global__ __launch_bounds__(256, 2) void wgrad3_f32_kernel(const Wgrad3Args a, float* __restrict__ partials,
                                                           int ntz, int nty, int ntx, int ncg, int CoP, int CiP) {
    using P = W3<TZ, TY, OT, CT>;
    constexpr int DS = P::DS, CS = P::CS, HY = P::HY, HX = P::HX, HVOL = P::HVOL, TVOL = P::TVOL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;
    float* xs = smem + OT * 16 * DS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = blockIdx.y / ncg, cgp = blockIdx.y % ncg;
    const int o0 = og * OT * 16, c0 = cgp * CT * 16;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    const bool xform = a.in_scale != nullptr;

    // this wave's taps
    int toff[7];
    bool tval[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int tap = wave + 4 * j;
        tval[j] = tap < 27;
        const int t = tval[j] ? tap : 0;
        toff[j] = ((t / 9) * HY + (t / 3) % 3) * HX + t % 3;
    }
    f32x4 acc[7][OT][CT];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int p = 0; p < OT; ++p)
#pragma unroll
            for (int q = 0; q < CT; ++q) acc[j][p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int abase = (lane & 15) * DS + (lane >> 4);
    const int bbase = (lane & 15) * CS + (lane >> 4);
    const int ntile = a.N * ntz * nty * ntx;

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int b = tile;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty; b /= nty;
        const int tz = b % ntz;
        const int n = b / ntz;
        const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16;
        __syncthreads();   // previous tile's MFMA reads are done
        // ---- stage dy tile: OT*16 channels x TVOL voxels (zero outside the volume / beyond Cout)
        for (int e = tid; e < OT * 16 * TVOL; e += 256) {
            const int ch = e / TVOL;
            const int r = e - ch * TVOL;
            const int z = r / (TY * 16), y = (r / 16) % TY, x = r & 15;
            const int gz = z0 + z, gy = y0 + y, gx = x0 + x, o = o0 + ch;
            float v = 0.f;
            if (o < a.Cout && gz < D && gy < H && gx < W) v = a.dy[((size_t)n * a.Cout + o) * DHW + (size_t)gz * HW + (size_t)gy * W + gx];
            dys[ch * DS + r] = v;
        }
        // ---- stage x halo tile: CT*16 channels x HVOL (fused producer transform, zero padding after it)
        for (int e = tid; e < CT * 16 * HVOL; e += 256) {
            const int ch = e / HVOL;
            const int s = e - ch * HVOL;
            const int hz = s / (HY * HX);
            const int r = s - hz * (HY * HX);
            const int hy = r / HX, hx = r - hy * HX;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1, c = c0 + ch;
            float v = 0.f;
            if (c < a.Cin && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                v = a.x[((size_t)n * a.Cin + c) * DHW + (size_t)gz * HW + (size_t)gy * W + gx];
                if (xform) {
                    v = v * a.in_scale[n * a.Cin + c] + a.in_shift[n * a.Cin + c];
                    v = v > 0.f ? v : v * a.in_slope;
                }
            }
            xs[ch * CS + s] = v;
        }
        __syncthreads();
        // ---- K loop over the tile's voxels, 4 per MFMA
#pragma unroll 1
        for (int zy = 0; zy < TZ * TY; ++zy) {
            const int z = zy / TY, y = zy - z * TY;
            const int az = abase + zy * 16;
            const int bz = bbase + (z * HY + y) * HX;
#pragma unroll
            for (int xq = 0; xq < 4; ++xq) {
                float af[OT];
#pragma unroll
                for (int p = 0; p < OT; ++p) af[p] = dys[az + p * 16 * DS + xq * 4];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (!tval[j]) continue;
                    float bf[CT];
#pragma unroll
                    for (int q = 0; q < CT; ++q) bf[q] = xs[bz + toff[j] + q * 16 * CS + xq * 4];
#pragma unroll
                    for (int p = 0; p < OT; ++p)
#pragma unroll
                        for (int q = 0; q < CT; ++q)
                            acc[j][p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[p], bf[q], acc[j][p][q], 0, 0, 0);
                }
            }
        }
    }
    // ---- write this workgroup's partial: partials[blockIdx.x][tap][o][c]
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (!tval[j]) continue;
        const int tap = wave + 4 * j;
#pragma unroll
        for (int p = 0; p < OT; ++p)
#pragma unroll
            for (int q = 0; q < CT; ++q) {
                const int c = c0 + q * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = o0 + p * 16 + (lane >> 4) * 4 + r;
                    if (o < CoP && c < CiP)
                        partials[(((size_t)blockIdx.x * 27 + tap) * CoP + o) * CiP + c] = acc[j][p][q][r];
                }
            }
    }
}

// dw[o*so + c*sc + tap] = sum_parts partials[part][tap][o][c]
__global__ void wgrad_reduce_kernel(const float* __restrict__ partials, int nparts, int taps, int CoP, int CiP,
                                    int Cout, int Cin, float* __restrict__ dw, int so, int sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = taps * Cout * Cin;
    if (i >= total) return;
    const int c = i % Cin;
    const int o = (i / Cin) % Cout;
    const int tap = i / (Cin * Cout);
    const size_t stride = (size_t)taps * CoP * CiP;
    const float* p = partials + ((size_t)tap * CoP + o) * CiP + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // fixed-order 4-way interleaved sum
    int k = 0;
    for (; k + 3 < nparts; k += 4) {
        s0 += p[(size_t)k * stride];
        s1 += p[(size_t)(k + 1) * stride];
        s2 += p[(size_t)(k + 2) * stride];
        s3 += p[(size_t)(k + 3) * stride];
    }
    for (; k < nparts; ++k) s0 += p[(size_t)k * stride];
    dw[(size_t)o * so + (size_t)c * sc + tap] = (s0 + s1) + (s2 + s3);
}

struct W3Choice { int tz, ty, ot, ct, nbx, ngroups, ncg; };

static W3Choice wgrad3_choose(int N, int Cin, int Cout, int D, int H, int W) {
    W3Choice c;
    const int CoP = round_up(Cout, 16), CiP = round_up(Cin, 16);
    if (CoP >= 32 && CiP >= 32) { c.tz = 2; c.ty = 4; c.ot = 2; c.ct = 2; }
    else { c.tz = 2; c.ty = 8; c.ot = 1; c.ct = 1; }
    const int nog = cdiv(CoP, 16 * c.ot);
    c.ncg = cdiv(CiP, 16 * c.ct);
    c.ngroups = nog * c.ncg;
    const long ntile = (long)N * cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, 16);
    long nbx = 512 / c.ngroups;
    if (nbx < 1) nbx = 1;
    if (nbx > ntile) nbx = ntile;
    c.nbx = (int)nbx;
    return c;
}

size_t wgrad3_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    const W3Choice c = wgrad3_choose(N, Cin, Cout, D, H, W);
    return (size_t)c.nbx * 27 * round_up(Cout, 16) * round_up(Cin, 16) * sizeof(float);
}

template <int TZ, int TY, int OT, int CT>
static int wgrad3_cfg(const Wgrad3Args& a, const W3Choice& c, hipStream_t s) {
    using P = W3<TZ, TY, OT, CT>;
    static bool attr_done = false;
    const size_t lds = (size_t)P::LDS_FLOATS * sizeof(float);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_f32_kernel<TZ, TY, OT, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad3)");
        attr_done = true;
    }
    const int CoP = round_up(a.Cout, 16), CiP = round_up(a.Cin, 16);
    dim3 grid(c.nbx, c.ngroups);
    hipLaunchKernelGGL((wgrad3_f32_kernel<TZ, TY, OT, CT>), grid, dim3(256), lds, s, a, (float*)a.ws,
                       cdiv(a.D, TZ), cdiv(a.H, TY), cdiv(a.W, 16), c.ncg, CoP, CiP);
    RU_CHECK_LAUNCH("wgrad3_f32_kernel");
    const int total = 27 * a.Cout * a.Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)a.ws, c.nbx, 27, CoP, CiP,
                       a.Cout, a.Cin, a.dw, a.Cin * 27, 27);
    RU_CHECK_LAUNCH("wgrad_reduce_kernel");
    return RU_OK;
}

int wgrad3_launch(const Wgrad3Args& a, hipStream_t s) {
    RU_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cout > 0 && a.D > 0 && a.H > 0 && a.W > 0, "wgrad3: bad shape");
    const W3Choice c = wgrad3_choose(a.N, a.Cin, a.Cout, a.D, a.H, a.W);
    if (a.ws_bytes < wgrad3_workspace_bytes(a.N, a.Cin, a.Cout, a.D, a.H, a.W) || !a.ws) {
        set_error("wgrad3: workspace too small");
        return RU_ENOMEM;
    }
    // a.db (bias gradient) is produced by bias_grad_launch at the call sites
    if (c.ot == 2) return wgrad3_cfg<2, 4, 2, 2>(a, c, s);
    return wgrad3_cfg<2, 8, 1, 1>(a, c, s);
}

// ------------------------------------------------------------------ 1x1x1
constexpr int W1_VC = 256;                       // voxels per chunk
constexpr int W1_RS = pad2mod4(W1_VC);           // LDS row stride (258)

template <int OT, int CT>
__global__ __launch_bounds__(256, 2) void wgrad1_f32_kernel(const Wgrad1Args a, float* __restrict__ partials,
                                                           int nchunk, int ncg, int CoP, int CiP) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;
    float* xs = smem + OT * 16 * W1_RS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = blockIdx.y / ncg, cgp = blockIdx.y % ncg;
    const int o0 = og * OT * 16, c0 = cgp * CT * 16;
    const size_t V = a.V;
    f32x4 acc[OT][CT];
#pragma unroll
    for (int p = 0; p < OT; ++p)
#pragma unroll
        for (int q = 0; q < CT; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int abase = (lane & 15) * W1_RS + (lane >> 4) + wave * 64;
    const long ntot = (long)a.N * nchunk;
    for (long t = blockIdx.x; t < ntot; t += gridDim.x) {
        const int n = (int)(t / nchunk);
        const size_t v0 = (size_t)(t % nchunk) * W1_VC;
        __syncthreads();
        for (int e = tid; e < OT * 16 * W1_VC; e += 256) {
            const int ch = e / W1_VC, r = e % W1_VC;
            const int o = o0 + ch;
            float v = 0.f;
            if (o < a.Cout && v0 + r < V) v = a.dy[((size_t)n * a.Cout + o) * V + v0 + r];
            dys[ch * W1_RS + r] = v;
        }
        for (int e = tid; e < CT * 16 * W1_VC; e += 256) {
            const int ch = e / W1_VC, r = e % W1_VC;
            const int c = c0 + ch;
            float v = 0.f;
            if (c < a.Cin && v0 + r < V) v = a.x[((size_t)n * a.Cin + c) * V + v0 + r];
            xs[ch * W1_RS + r] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float af[OT], bf[CT];
#pragma unroll
            for (int p = 0; p < OT; ++p) af[p] = dys[abase + p * 16 * W1_RS + ks * 4];
#pragma unroll
            for (int q = 0; q < CT; ++q) bf[q] = xs[abase + q * 16 * W1_RS + ks * 4];
#pragma unroll
            for (int p = 0; p < OT; ++p)
#pragma unroll
                for (int q = 0; q < CT; ++q)
                    acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[p], bf[q], acc[p][q], 0, 0, 0);
        }
    }
    const int part = blockIdx.x * 4 + wave;
#pragma unroll
    for (int p = 0; p < OT; ++p)
#pragma unroll
        for (int q = 0; q < CT; ++q) {
            const int c = c0 + q * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + p * 16 + (lane >> 4) * 4 + r;
                if (o < CoP && c < CiP) partials[((size_t)part * CoP + o) * CiP + c] = acc[p][q][r];
            }
        }
}

struct W1Choice { int ot, ct, nbx, ngroups, ncg, nchunk; };

static W1Choice wgrad1_choose(int N, int Cin, int Cout, size_t V) {
    W1Choice c;
    const int CoP = round_up(Cout, 16), CiP = round_up(Cin, 16);
    c.ot = CoP >= 32 ? 2 : 1;
    c.ct = CiP >= 32 ? 2 : 1;
    c.ncg = cdiv(CiP, 16 * c.ct);
    c.ngroups = cdiv(CoP, 16 * c.ot) * c.ncg;
    c.nchunk = (int)((V + W1_VC - 1) / W1_VC);
    long nbx = 1024 / c.ngroups;
    if (nbx < 1) nbx = 1;
    const long ntot = (long)N * c.nchunk;
    if (nbx > ntot) nbx = ntot;
    c.nbx = (int)nbx;
    return c;
}

size_t wgrad1_workspace_bytes(int N, int Cin, int Cout, size_t V) {
    const W1Choice c = wgrad1_choose(N, Cin, Cout, V);
    return (size_t)c.nbx * 4 * round_up(Cout, 16) * round_up(Cin, 16) * sizeof(float);
}

template <int OT, int CT>
static int wgrad1_cfg(const Wgrad1Args& a, const W1Choice& c, hipStream_t s) {
    static bool attr_done = false;
    const size_t lds = (size_t)(OT + CT) * 16 * W1_RS * sizeof(float);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1_f32_kernel<OT, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad1)");
        attr_done = true;
    }
    const int CoP = round_up(a.Cout, 16), CiP = round_up(a.Cin, 16);
    hipLaunchKernelGGL((wgrad1_f32_kernel<OT, CT>), dim3(c.nbx, c.ngroups), dim3(256), lds, s, a, (float*)a.ws, c.nchunk, c.ncg, CoP, CiP);
    RU_CHECK_LAUNCH("wgrad1_f32_kernel");
    const int total = a.Cout * a.Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)a.ws, c.nbx * 4, 1, CoP, CiP,
                       a.Cout, a.Cin, a.dw, a.ldw, 1);
    RU_CHECK_LAUNCH("wgrad_reduce_kernel");
    return RU_OK;
}

int wgrad1_launch(const Wgrad1Args& a, hipStream_t s) {
    RU_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cout > 0 && a.V > 0 && a.ldw >= a.Cin, "wgrad1: bad shape");
    const W1Choice c = wgrad1_choose(a.N, a.Cin, a.Cout, a.V);
    if (!a.ws || a.ws_bytes < wgrad1_workspace_bytes(a.N, a.Cin, a.Cout, a.V)) {
        set_error("wgrad1: workspace too small");
        return RU_ENOMEM;
    }
    if (c.ot == 2 && c.ct == 2) return wgrad1_cfg<2, 2>(a, c, s);
    if (c.ot == 2) return wgrad1_cfg<2, 1>(a, c, s);
    if (c.ct == 2) return wgrad1_cfg<1, 2>(a, c, s);
    return wgrad1_cfg<1, 1>(a, c, s);
}

}  // namespace ru
