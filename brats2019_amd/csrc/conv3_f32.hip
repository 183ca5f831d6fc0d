// conv3_f32.hip -- 3x3x3 stride-1 pad-1 convolution as an implicit GEMM on the exact-f32 matrix core
// instruction v_mfma_f32_16x16x4_f32 (gfx950).  Replaces nn.Conv3d at model.py:72-73,336,348 (forward) and,
// with flipped/transposed packed weights, its data gradient.
//
// Mapping (per workgroup of 4 waves):
//   GEMM M = output voxels   : a TZ x TY x 16 tile of one sample; one MFMA M-tile = 16 consecutive x voxels
//   GEMM N = output channels : NT tiles of 16
//   GEMM K = 27 taps x Cin   : walked in chunks of KC input channels staged in LDS
//   LDS  : input halo tile  xs[KC][(TZ+2)(TY+2)(18)]  NCDHW order, channel stride == 16 (mod 32) words so the
//          A-fragment read (lanes 0-15: channel c, x..x+15; lanes 16-31: channel c+1) is bank-conflict free;
//          weight chunk     ws[27][KC][WS], WS == 16 (mod 32) for the same reason on the B fragment.
//   A fragment lane l: xs[c0 + (l>>4)][pos + (l&15)]   (one ds_read_b32, immediate offset per tap)
//   B fragment lane l: ws[tap][c0 + (l>>4)][o0 + (l&15)]
//   C/D        lane l: rows (l>>4)*4 + r = 4 consecutive x voxels, column l&15 = output channel
//                      -> one 16-byte store per accumulator into NCDHW.
// Fusions: optional per-(n,c) affine + LeakyReLU on the INPUT while staging (GroupNorm-apply + activation of
// the producer, model.py:92-94, never materialised); optional per-tile (sum, sumsq) of the OUTPUT for the
// consumer GroupNorm's statistics; optional bias, residual add and sigmoid (model.py:431) in the epilogue.
#include "conv3_epilogue.hpp"

namespace ru {

template <int TZ, int TY, int KC, int NT>
struct C3 {
    static constexpr int TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    static constexpr int HVOL = HZ * HY * HX;
    static constexpr int CS = ((HVOL - 16 + 31) / 32) * 32 + 16;       // >= HVOL, == 16 (mod 32)
    static constexpr int WS = (NT % 2 == 1) ? NT * 16 : NT * 16 + 16;  // == 16 (mod 32)
    static constexpr int MT = TZ * TY / 4;                              // M-tiles per wave
    static constexpr int LDS_FLOATS = KC * CS + 27 * KC * WS;
    static_assert(MT >= 1 && TY % MT == 0, "a wave's M-tiles must lie in one z-slab");
    static_assert(CS >= HVOL && CS % 32 == 16 && WS % 32 == 16, "bank layout");
    static_assert(KC % 4 == 0, "K chunk is a multiple of the MFMA K");
};

template <int TZ, int TY, int KC, int NT>
__global__ __launch_bounds__(256, 2) void conv3_f32_kernel(const Conv3Args a0, int ntz, int nty, int ntx) {
    using P = C3<TZ, TY, KC, NT>;
    // split-K (Conv3Args::ksplit = gridDim.z): this workgroup sums input channels [c_begin, c_end) into partial tensor blockIdx.z
    Conv3Args a = a0;
    const int c_span = a0.CinP / (int)gridDim.z, c_begin = (int)blockIdx.z * c_span, c_end = c_begin + c_span;
    a.y = a0.y + (size_t)blockIdx.z * a0.N * a0.Cout * ((size_t)a0.D * a0.H * a0.W);
    constexpr int MT = P::MT, CS = P::CS, WS = P::WS, HY = P::HY, HX = P::HX, HVOL = P::HVOL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + KC * CS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = b % ntx; b /= ntx;
    const int ty = b % nty; b /= nty;
    const int tz = b % ntz;
    const int n = b / ntz;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16;
    const int co0 = blockIdx.y * (NT * 16);
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;

    // ---- per-thread staging slots, computed once (they do not depend on the channel).
    // Vector path (W % 4 == 0): a halo row [x0-1, x0+17) is covered by six 16-byte aligned segments [x0-4+4q, +4);
    // one slot = one float4 load; all KC channels' loads of a thread are issued before the first LDS store.
    // Scalar path (ragged W): one slot = one element.
    constexpr int NROW = P::HZ * HY;
    constexpr int NSV = (NROW * 6 + 255) / 256;
    constexpr int NS = (HVOL + 255) / 256;
    const bool vec = (W & 3) == 0;
    int goff[NS];     // scalar path: global offset inside one channel, -1 = zero fill
    int gv[NSV];      // vector path: global offset of the float4, -1 = zero fill, -2 = no slot
    int lv[NSV];      // vector path: LDS offset of element 0 of the float4 (may point 3 before the row start)
    if (vec) {
#pragma unroll
        for (int j = 0; j < NSV; ++j) {
            const int item = tid + j * 256;
            const int row = item / 6, q = item - row * 6;
            const int hz = row / HY, hy = row - hz * HY;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 - 4 + 4 * q;
            const bool slot = item < NROW * 6;
            const bool ok = slot && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && gx >= 0 && gx < W;
            gv[j] = !slot ? -2 : (ok ? (gz * H + gy) * W + gx : -1);
            lv[j] = row * HX + 4 * q - 3;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int s = tid + j * 256;
            const int hz = s / (HY * HX);
            const int r = s - hz * (HY * HX);
            const int hy = r / HX;
            const int hx = r - hy * HX;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = (s < HVOL) && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            goff[j] = ok ? (gz * H + gy) * W + gx : -1;
        }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int mz = (wave * MT) / TY, my0 = (wave * MT) % TY;
    const int abase = (lane >> 4) * CS + (lane & 15) + mz * (HY * HX) + my0 * HX;
    const int bbase = (lane >> 4) * WS + (lane & 15);
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;    // neutral constants (1, 0, 1) make the fused transform branch-free and exact

    for (int c0 = c_begin; c0 < c_end; c0 += KC) {
        if (c0 != c_begin) __syncthreads();
        // ---- stage the input halo tile (zero padding AFTER the fused transform: the reference pads the activated tensor)
        if (vec) {
            float4 v[KC][NSV];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                // UNCONDITIONAL loads from clamped (always valid) addresses, zeroed by select below: a conditional load makes
                // hipcc branch around it and wait vmcnt(0) per load (one load in flight per thread)
                const int cg = c0 + c;
                const float* xp = a.x + ((size_t)n * a.Cin + (cg < a.Cin ? cg : 0)) * DHW;
#pragma unroll
                for (int j = 0; j < NSV; ++j) v[c][j] = *reinterpret_cast<const float4*>(xp + (gv[j] > 0 ? gv[j] : 0));
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int cg = c0 + c;
                float sc = 1.f, sh = 0.f;
                if (xform && cg < a.Cin) { sc = a.in_scale[n * a.Cin + cg]; sh = a.in_shift[n * a.Cin + cg]; }
#pragma unroll
                for (int j = 0; j < NSV; ++j) {
                    if (gv[j] == -2) continue;
                    float t[4] = {v[c][j].x, v[c][j].y, v[c][j].z, v[c][j].w};
                    const bool live = gv[j] >= 0 && cg < a.Cin;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t[e] = fmaf(t[e], sc, sh);
                        t[e] = fmaxf(t[e], t[e] * slope);       // LeakyReLU for 0 < slope <= 1
                        t[e] = live ? t[e] : 0.f;
                    }
                    const int q = (tid + j * 256) % 6;
                    float* dst = xs + c * CS + lv[j];
                    if (q == 0) dst[3] = t[3];                 // only x0-1 of the first segment is inside the halo row
                    else if (q == 5) dst[0] = t[0];            // only x0+16 of the last one
                    else { dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3]; }
                }
            }
        } else {
#pragma unroll 2
            for (int c = 0; c < KC; ++c) {
                const int cg = c0 + c;
                const bool cok = cg < a.Cin;
                const float* xp = a.x + ((size_t)n * a.Cin + (cok ? cg : 0)) * DHW;
                float sc = 1.f, sh = 0.f;
                if (xform && cok) { sc = a.in_scale[n * a.Cin + cg]; sh = a.in_shift[n * a.Cin + cg]; }
                float v[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) v[j] = xp[goff[j] > 0 ? goff[j] : 0];       // unconditional, clamped (see above)
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    float t = fmaf(v[j], sc, sh);
                    t = fmaxf(t, t * slope);
                    t = (cok && goff[j] >= 0) ? t : 0.f;
                    if (tid + j * 256 < HVOL) xs[c * CS + tid + j * 256] = t;
                }
            }
        }
        // ---- stage the weight chunk  wp[tap][c0+kc][co0 ..]  ->  ws[tap][kc][WS]
        constexpr int ROW4 = NT * 4;
        for (int i = tid; i < 27 * KC * ROW4; i += 256) {
            const int o4 = i % ROW4;
            const int r = i / ROW4;
            const int kc = r % KC;
            const int tap = r / KC;
            const float4 w4 = *reinterpret_cast<const float4*>(a.wp + ((size_t)(tap * a.CinP + c0 + kc) * a.CoutP + co0 + o4 * 4));
            *reinterpret_cast<float4*>(ws + (tap * KC + kc) * WS + o4 * 4) = w4;
        }
        __syncthreads();
        // ---- 27 taps x KC/4 k-steps of MT x NT MFMAs, every LDS offset an immediate
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
            for (int cg = 0; cg < KC / 4; ++cg) {
                float bf[NT], af[MT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bf[t] = ws[bbase + (tap * KC + cg * 4) * WS + t * 16];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = xs[abase + cg * 4 * CS + (dz * HY + dy + i) * HX + dx];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[t], acc[i][t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue (bias / residual / GN tile statistics / sigmoid / store)
    conv3_epilogue<MT, NT>(a, acc, smem, n, z0, y0, x0, mz, my0, co0, tz, ty, tx, ntz, nty, ntx);
}

// ------------------------------------------------------------------ host side
struct C3Choice { int tz, ty, kc, nt; };

int conv3_cin_pad(int Cin) { return Cin <= 4 ? 4 : round_up(Cin, 8); }

static C3Choice conv3_choose(int N, int Cin, int Cout, int D, int H, int W) {
    const int CoutP = conv3_cout_pad(Cout);
    const int kc = conv3_cin_pad(Cin) == 4 ? 4 : 8;
    if (kc == 4) return {4, 8, 4, 1};
    auto blocks = [&](int tz, int ty, int nt) {
        return (long)N * cdiv(D, tz) * cdiv(H, ty) * cdiv(W, 16) * cdiv(CoutP, 16 * nt);
    };
    const int nt = CoutP >= 32 ? 2 : 1;
    const int cand[3][2] = {{4, 8}, {2, 8}, {2, 4}};
    for (int i = 0; i < 3; ++i)
        if (blocks(cand[i][0], cand[i][1], nt) >= 768) return {cand[i][0], cand[i][1], kc, nt};
    // small problem: smallest tile; prefer more workgroups over register blocking on N
    if (nt == 2 && blocks(2, 4, 2) < 512) return {2, 4, kc, 1};
    return {2, 4, kc, nt};
}

// Split-K factor of an exact-f32 conv: shapes with fewer than two workgroups per CU walk all their input-channel chunks in sequence with
// every chunk's global-load latency exposed (one wave per SIMD, no double buffer: the 128-channel level of a batch-1 forward ran at 26 %
// of the f32 MFMA peak) -- the chunks are spread over up to four co-resident workgroups per CU.
int conv3_f32_ksplit(int N, int Cin, int Cout, int D, int H, int W) {
    const C3Choice c = conv3_choose(N, Cin, Cout, D, H, W);
    if (c.kc != 8) return 1;
    const long blocks = (long)N * cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, 16) * cdiv(conv3_cout_pad(Cout), 16 * c.nt);
    const int nchunk = conv3_cin_pad(Cin) / 8;
    int ks = 1;
    while (nchunk % (ks * 2) == 0 && nchunk / (ks * 2) >= 2 && blocks * (ks * 2) <= 1024) ks *= 2;
    return ks;
}

int conv3_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W, int mode) {
    if (conv3_effective_mode(mode, W) == RU_PREC_BF16X3) return conv3_sb_tiles_per_sample(N, Cin, Cout, D, H, W);
    if (conv3_f32_ksplit(N, Cin, Cout, D, H, W) > 1) return gn_stats_tiles((size_t)D * H * W);      // (the engine's split path: gn_stats_launch takes them)
    const C3Choice c = conv3_choose(N, Cin, Cout, D, H, W);
    return cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, 16);
}

size_t conv3_packed_floats(int Cin_conv, int Cout_conv) {
    return (size_t)27 * conv3_cin_pad(Cin_conv) * conv3_cout_pad(Cout_conv);
}

template <int TZ, int TY, int KC, int NT>
static int launch_cfg(const Conv3Args& a, hipStream_t s) {
    using P = C3<TZ, TY, KC, NT>;
    static PerDevice attr_done;
    const size_t lds = (size_t)P::LDS_FLOATS * sizeof(float);
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_f32_kernel<TZ, TY, KC, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3)");
        attr_done.set();
    }
    const int ntz = cdiv(a.D, TZ), nty = cdiv(a.H, TY), ntx = cdiv(a.W, 16);
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    RU_REQUIRE(ks == 1 || ((a.CinP / KC) % ks == 0 && !a.stat_partials && !a.bias && !a.add && !a.sigmoid),
               "conv3_f32: split-K divides the input-channel chunks and has a plain epilogue");
    dim3 grid((unsigned)((long)a.N * ntz * nty * ntx), (unsigned)cdiv(a.CoutP, NT * 16), (unsigned)ks);
    hipLaunchKernelGGL((conv3_f32_kernel<TZ, TY, KC, NT>), grid, dim3(256), lds, s, a, ntz, nty, ntx);
    RU_CHECK_LAUNCH("conv3_f32_kernel");
    return RU_OK;
}

int conv3_launch(const Conv3Args& a_in, hipStream_t s) {
    Conv3Args a = a_in;
    a.CinP = conv3_cin_pad(a.Cin);
    a.CoutP = conv3_cout_pad(a.Cout);
    RU_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cout > 0 && a.D > 0 && a.H > 0 && a.W > 0, "conv3: bad shape");
    RU_REQUIRE((size_t)a.D * a.H * a.W < (1u << 31), "conv3: volume too large for 32-bit voxel offsets");
    if ((a.in_c16 || a.out_c16) && !a.in_c4 && a.mode == RU_PREC_F32) return conv3_f32c_launch(a, a.wfrag, s);      // exact-f32 voxel-major flow (inference)
    if (a.in_c16 || a.out_c16 || a.in_c4) RU_REQUIRE(a.mode == RU_PREC_BF16X3, "conv3: voxel-major tensors need the split-bf16 kernel");
    if (a.in_c16 || a.out_c16 || a.in_c4 || conv3_effective_mode(a.mode, a.W) == RU_PREC_BF16X3) {
        RU_REQUIRE(a.wfrag != nullptr, "conv3: bf16x3 mode needs packed weight fragments");
        return conv3_sb_launch(a, s);
    }
    RU_REQUIRE(a.wp != nullptr, "conv3: f32 mode needs packed weights");
    const C3Choice c = conv3_choose(a.N, a.Cin, a.Cout, a.D, a.H, a.W);
#define RU_C3_CASE(TZ, TY, KC, NT) \
    if (c.tz == TZ && c.ty == TY && c.kc == KC && c.nt == NT) return launch_cfg<TZ, TY, KC, NT>(a, s);
    RU_C3_CASE(4, 8, 4, 1)
    RU_C3_CASE(4, 8, 8, 1)
    RU_C3_CASE(2, 8, 8, 1)
    RU_C3_CASE(2, 4, 8, 1)
    RU_C3_CASE(4, 8, 8, 2)
    RU_C3_CASE(2, 8, 8, 2)
    RU_C3_CASE(2, 4, 8, 2)
#undef RU_C3_CASE
    set_error("conv3: no kernel for config tz=%d ty=%d kc=%d nt=%d", c.tz, c.ty, c.kc, c.nt);
    return RU_EINVAL;
}

__global__ void conv3_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin_f, int Cout_f, int mode,
                                  int CinP, int CoutP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 27 * CinP * CoutP;
    if (i >= total) return;
    const int o = i % CoutP;
    const int c = (i / CoutP) % CinP;
    const int tap = i / (CoutP * CinP);
    float v = 0.f;
    if (mode == 0) {
        if (c < Cin_f && o < Cout_f) v = w[((size_t)o * Cin_f + c) * 27 + tap];
    } else {   // data gradient: input channels = Cout_f, output channels = Cin_f, taps mirrored
        if (c < Cout_f && o < Cin_f) v = w[((size_t)c * Cin_f + o) * 27 + (26 - tap)];
    }
    wp[i] = v;
}

int conv3_pack_weights(const float* w, float* wp, int Cin_f, int Cout_f, int mode, hipStream_t s) {
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int CinP = conv3_cin_pad(cin_conv), CoutP = conv3_cout_pad(cout_conv);
    const int total = 27 * CinP * CoutP;
    hipLaunchKernelGGL(conv3_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, wp, Cin_f, Cout_f, mode, CinP, CoutP);
    RU_CHECK_LAUNCH("conv3_pack_kernel");
    return RU_OK;
}

}  // namespace ru
