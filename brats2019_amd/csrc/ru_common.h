// Internal declarations shared by the HIP translation units of libresunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <vector>

#include "../../include/resunet_hip.h"

namespace ru {

// ---- split-bf16 operand format: v = hi + lo with hi = bf16_rne(v), lo = bf16_rne(v - hi); two values per call, packed (low half = a).
// 3 VALU per value: v_cvt_pk_bf16_f32, v_perm_b32 / v_and_b32 to re-expand the hi halves, two v_sub_f32, v_cvt_pk_bf16_f32.  The low
// half is re-expanded with a byte permute, not a shift: the optimizer rewrites (cvt_pk(a, b) << 16) as cvt_pk(a, undef) << 16 -- a
// second conversion per pair.  The two remainders must stay two plain v_sub_f32 (the library is built with -fno-slp-vectorize so that
// they are not paired into one v_pk_add_f32): these helpers run in the staging waves that share a SIMD with the MFMA waves, and
// there a packed-f32 instruction starves (tools/coissue_probe.hip: 85 ns per v_pk_* against 7 ns per plain VALU instruction).
typedef __bf16 ru_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hb, unsigned& lb) {
    ru_bf16x2 h;
    h[0] = (__bf16)a;
    h[1] = (__bf16)b;
    hb = __builtin_bit_cast(unsigned, h);
    const float h0 = __builtin_bit_cast(float, __builtin_amdgcn_perm(hb, hb, 0x01000c0cu));     // (hb & 0xffff) << 16
    const float h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
    ru_bf16x2 l;
    l[0] = (__bf16)(a - h0);
    l[1] = (__bf16)(b - h1);
    lb = __builtin_bit_cast(unsigned, l);
}
// ---- gradient operand form of the MX product scheme (conv3_mx_pack.hpp, MXG_*): one exponent per voxel (16 channels), DERIVED from the voxel's bf16 hi values --
// the reader recomputes it from the hi packets it stages (conv3_mx_kernel<GRAD>), so it is stored nowhere.
// amax = the largest |bf16 hi| of the voxel (both channel halves).  byte = E8M0 of 2^e with amax / 2^e in [128, 256); sc = 2^e, sc8 = 2^(e-8) as floats.
__device__ __forceinline__ unsigned mxg_exponent_byte(unsigned biased_exponent_of_amax) {
    const int b = (int)biased_exponent_of_amax - 7;
    return (unsigned)(b < 9 ? 9 : b);                    // (voxels below 2^-118: the scale stops following them -- their values still convert, to smaller codes)
}
typedef short ru_s16x2 __attribute__((ext_vector_type(2)));
// step 1: 8 values of a voxel half -> 4 dwords bf16 hi (RNE, split_pair's), the exact residuals, and the largest |hi| of the half
template <class V>
__device__ __forceinline__ float mxg_hi8(const float (&t)[8], V& hi, float (&lo)[8]) {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ru_bf16x2 h;
        h[0] = (__bf16)t[2 * c];
        h[1] = (__bf16)t[2 * c + 1];
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        hi[c] = hb;
        const float h0 = __builtin_bit_cast(float, __builtin_amdgcn_perm(hb, hb, 0x01000c0cu)), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
        lo[2 * c] = t[2 * c] - h0;
        lo[2 * c + 1] = t[2 * c + 1] - h1;
        m = fmaxf(m, fmaxf(fabsf(h0), fabsf(h1)));
    }
    return m;
}
// step 2 (amax16 = the larger of the two halves' maxima): 2 dwords e4m3(lo / 2^(e-8)), 2 dwords e4m3(g / 2^e).  v_cvt_scalef32_pk_fp8_f32 divides by its scale operand
// (tools/mx_cvt_probe.hip); the caller has set MODE.FP16_OVFL so that the conversions saturate
__device__ __forceinline__ void mxg_cvt8(const float (&t)[8], const float (&lo)[8], float amax16, unsigned (&l8)[2], unsigned (&x8)[2]) {
    const unsigned b = mxg_exponent_byte((__builtin_bit_cast(unsigned, amax16) >> 23) & 0xffu);
    const float sc = __builtin_bit_cast(float, b << 23), sc8 = __builtin_bit_cast(float, (b - 8u) << 23);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        ru_s16x2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(ru_s16x2, lo[4 * d]), lo[4 * d], lo[4 * d + 1], sc8, false);
        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lo[4 * d + 2], lo[4 * d + 3], sc8, true);
        l8[d] = __builtin_bit_cast(unsigned, w);
        ru_s16x2 v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(ru_s16x2, t[4 * d]), t[4 * d], t[4 * d + 1], sc, false);
        v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, t[4 * d + 2], t[4 * d + 3], sc, true);
        x8[d] = __builtin_bit_cast(unsigned, v);
    }
}
template <int NP, class V>
__device__ __forceinline__ void split_n(const float (&t)[2 * NP], V& hi, V& lo) {       // 2*NP floats -> NP packed dwords of hi and of lo
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        unsigned hb, lb;
        split_pair(t[2 * i], t[2 * i + 1], hb, lb);
        hi[i] = hb;
        lo[i] = lb;
    }
}

// thread-local error message (ru_last_error)
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define RU_CHECK_LAUNCH(what)                                  \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return ru::hip_fail(e__, what); \
    } while (0)

#define RU_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            ru::set_error(__VA_ARGS__);  \
            return RU_EINVAL;            \
        }                                \
    } while (0)

// a flag per HIP device: hipFuncSetAttribute (dynamic LDS above 64 KB) is a per-device setting, so a second device used by the same
// process must not inherit the first one's "done"
struct PerDevice {
    bool done[64] = {};
    static int cur() { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : -1; }
    bool get() const { const int d = cur(); return d >= 0 && done[d]; }
    void set() { const int d = cur(); if (d >= 0) done[d] = true; }
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ------------------------------------------------------------------ in-launch finalization of partial sums (fin_tail.hpp)
// RU_FUSE_TAIL_FINALIZE: the kernel that writes the per-workgroup partial sums of a GroupNorm (forward statistics, or the two sums of its
// backward) also finalizes them -- its LAST workgroup to finish (one integer ticket, agent scope, reset by the finisher so that the
// launch can be replayed from a hipGraph) reads all partials back in a fixed order and writes what the separate finalize kernels
// (gn_finalize_kernel / gn_bwd_finalize_kernel) wrote.  No float atomics: the result does not depend on who finishes last.
// Partials are published with 8-byte agent-scope stores and read with agent-scope loads (per-XCD L2s are not coherent with each other).
struct FinTail {
    unsigned* ticket;        // null: no tail (the caller launches the finalize kernel)
    int kind;                // 1: (sum, sumsq) -> mean / rstd / scale / shift (/ bst_k); 2: GroupNorm-backward sums -> coef / dgamma / dbeta
    int nblk, N, C, G;       // partials [N][C][nblk][2]
    size_t V;
    float eps;
    int s2_sign;             // kind 2: the second sum carries sign(gamma) (fused conv statistics, Conv3Args::bst_*)
    const float* gamma;
    const float* beta;       // kind 1
    float* mean;             // kind 1: out; kind 2: in
    float* rstd;
    float* scale;            // kind 1
    float* shift;
    float* bst_k;            // kind 1, optional
    float* coef;             // kind 2
    float* dgamma;
    float* dbeta;
};
// LDS scratch the tail needs at the pointer handed to fin_tail(): the flag, plus S[N][C][2] doubles for kind 2
static inline size_t fin_tail_lds_bytes(const FinTail& f) { return 16 + ((f.ticket && f.kind == 2) ? (size_t)f.N * f.C * 16 : 0); }

// ------------------------------------------------------------------ conv 3x3x3 (conv3_f32.hip)
// Implicit GEMM on v_mfma_f32_16x16x4_f32.  Weights must be PACKED: wp[tap][CinP][CoutP] (K-major).
struct Conv3Args {
    const float* x;          // [N][Cin][D][H][W]
    const float* wp;         // packed weights [27][CinP][CoutP]
    const float* bias;       // [Cout] or null
    float* y;                // [N][Cout][D][H][W]
    const float* add;        // same shape as y, added in the epilogue, or null
    const float* in_scale;   // [N][Cin] fused input transform v -> lrelu(v*scale+shift, in_slope), or null
    const float* in_shift;
    float in_slope;
    // Residual of the INPUT (inference head only: conv3_sb_head_takes_residual): the conv reads lrelu(x*scale+shift, in_slope) + in_res -- the output of
    // the last Residual block (model.py:112-116: x + relu2(norm2(conv2))) formed in the staging, so that block's residual pass never runs.  Voxel-major,
    // same extents as x; zero padding applies to the SUM.  Null everywhere else.
    const float* in_res;
    float* in_sum_out;       // with in_res (training): the staged sum is also WRITTEN here (voxel-major, every voxel exactly once) -- the block output the backward reads
    float* stat_partials;    // [N][Cout][nblk][2] per-tile (sum, sumsq) of y, or null
    int sigmoid;             // apply 1/(1+exp(-v)) in the epilogue
    int N, Cin, Cout, D, H, W;
    int CinP, CoutP;
    int mode;                // RU_PREC_F32: exact-f32 MFMA (wp); RU_PREC_BF16X3: split-bf16, 3 MFMA products (wfrag)
    const void* wfrag;       // packed bf16 hi/lo weight fragments (conv3_sb.hip) when mode == RU_PREC_BF16X3
    // Engine-internal voxel-major layout "C16" (split-bf16 kernels only): [N][C/16][D][H][W][16], C % 16 == 0.
    // in_c16: x (and in_scale/in_shift semantics unchanged); out_c16: y, add.  0 = NCDHW.
    int in_c16, out_c16;
    int in_s16;              // x is C16 in SPLIT form (see gn_bwd_apply16_launch): the staging copies hi/lo packets, no conversion, no transform
    // with in_s16: x is in the GRADIENT OPERAND form of the MX scheme instead (conv3_mx_pack.hpp: [bf16 hi | hi | e4m3 (lo, value) ch 0-7 | ch 8-15] per voxel, same 64 bytes, the
    // voxel's exponent implied by its hi values): conv3_mx_kernel<GRAD> (16 -> 16 channels; written by wgrad3_tz<1,0,3,3> or conv3_mxg_split_launch)
    int in_g16;
    int in_c4;               // x is a [N][D][H][W][4] copy (pad_to_c4) of a tensor with Cin <= 4: conv3_sb2c4_kernel, wfrag from conv3_sb4_pack_weights
    // Fused GroupNorm-BACKWARD statistics (persistent split-bf16 kernel, C16 in and out, no bias / sigmoid; a residual `add` is part of the
    // output and therefore of the sums): this conv's output is
    // the gradient d w.r.t. the activation that followed a GroupNorm of `bst_y` (same shape as y).  The epilogue then writes, instead
    // of (sum, sumsq), the partial sums the GroupNorm backward needs -- S1 = sum dh, S2' = sum dh*u with u = y*k1 + k2 (= sign(gamma) *
    // xhat), dh = u > thr ? d : d*bst_slope -- to stat_partials, saving the separate reduce pass over (y, d).  bst_k: [N][3][Cout] =
    // (k1, k2, thr) written by gn_finalize_launch(..., bst_k); gn_bwd_finalize_launch(..., s2_sign = 1) undoes the sign.
    const float* bst_y;
    const float* bst_k;
    float bst_slope;
    // MFMA products per operand pair in split-bf16 mode: 0 / 3 = hi*hi + lo*hi + hi*lo; 1 = hi*hi only (plain bf16 operands: the engine's
    // gradient precision RU_PREC_BF16).  Honoured by the persistent voxel-major kernel; every other kernel keeps three products.
    // 2 = the input is an ACTIVATION tensor (a forward convolution) and may take the fp16 + MX-fp8 scheme (conv3_mx.hpp: f16*f16 + e4m3 cross terms on
    // v_mfma_scale_f32_16x16x128_f8f6f4) where a kernel for the shape exists; elsewhere it means three products.  Never set for gradient inputs.
    int products;
    // Split-K of the exact-f32 kernel (small shapes: fewer workgroups than two per CU, each walking ALL input-channel chunks with the
    // load latency of every chunk exposed): ksplit > 1 workgroups share a (tile, cout block), each sums CinP / ksplit input channels and
    // writes its partial OUTPUT tensor to y + z * N * Cout * D * H * W (z = 0 .. ksplit-1; plain epilogue: no statistics, bias, residual,
    // activation); sum_partials_launch adds them in z order.  0 / 1 = off.
    int ksplit;
    FinTail fin;             // split-bf16 kernels: finalize stat_partials in the launch (ticket null = off)
};
int conv3_f32_ksplit(int N, int Cin, int Cout, int D, int H, int W);     // split factor the engine uses for an exact-f32 conv of this shape (1 = none)
int sum_partials_launch(const float* part, int ksplit, size_t n, float* y, hipStream_t s);    // y[i] = part[0][i] + part[1][i] + ... (fixed order)
int conv3_cin_pad(int Cin);                       // CinP for a given Cin
static inline int conv3_cout_pad(int Cout) { return round_up(Cout, 16); }
// effective mode for a shape (the split-bf16 kernel needs W % 4 == 0; otherwise the f32 kernel runs)
static inline int conv3_effective_mode(int mode, int W) { return (mode == RU_PREC_BF16X3 && (W & 3) == 0) ? RU_PREC_BF16X3 : RU_PREC_F32; }
// number of spatial tiles per sample the kernel will use (== nblk of stat_partials)
int conv3_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W, int mode);
int conv3_launch(const Conv3Args& a, hipStream_t s);
// exact-f32 convolution on voxel-major tensors (conv3_f32c.hip): forward only; `wfr` = fragments from conv3_f32c_pack_weights (Conv3Args::wfrag when
// mode == RU_PREC_F32 and a voxel-major side is set -- conv3_launch routes there)
size_t conv3_f32c_frag_bytes(int Cin_conv, int Cout_conv);
int conv3_f32c_pack_weights(const float* w, void* wfr, int Cin_f, int Cout_f, int mode, hipStream_t s);
int conv3_f32c_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W);
int conv3_f32c_launch(const Conv3Args& a, const void* wfr, hipStream_t s);
bool conv3_f32c_head_takes_residual(int Cin, int Cout, int W);
// split-bf16 path (conv3_sb.hip)
int conv3_sb_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W, int products = 3);    // products: Conv3Args::products of the launch
bool conv3_sb_head_form_enabled();                // RU_HEAD_FORM=0 keeps the <= 4-output-channel convolutions on the 16-column kernel (A/B runs, parity tests)
bool conv3_sb_uses_wz(int N, int Cin, int Cout, int D, int H, int W, int products);
bool conv3_sb_head_takes_residual(int N, int Cin, int Cout, int D, int H, int W);   // the 16 -> <=4 voxel-major-in / NCDHW-out conv of this shape takes the head-form kernel, which stages Conv3Args::in_res (RU_HEAD_RES=0: never)
int conv3_sb_launch(const Conv3Args& a, hipStream_t s);
size_t conv3_sb_frag_bytes(int Cin_conv, int Cout_conv);          // direct fragments + the Winograd-z fragments behind them (where the channel counts allow)
size_t conv3_sb_frag_bytes_direct(int Cin_conv, int Cout_conv);
size_t conv3_sb4_frag_bytes(int Cout_conv);
int conv3_sb4_pack_weights(const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, hipStream_t s);
bool conv3_sb4_usable(int N, int Cin, int Cout, int D, int H, int W);                                   // shape fits the 4-channel kernel
int conv3_sb_pack_weights(const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, hipStream_t s, bool grad_operand = false);   // mode 0 fwd, 1 data-gradient; grad_operand: the MX fragments in the gradient-operand variant
// the same for many weights in one launch: add entries, then flush (add flushes by itself when the table is full)
constexpr int RU_PACK_BATCH = 64;
struct SbPackEntry { const float* w; void* wfrag; int Cin_f, Cout_f, mode, nchunk, ncog, forms; };
struct SbPackBatch { SbPackEntry e[RU_PACK_BATCH]; int n; };
// all_forms false: only the forms a training step launches (sb_pack_forms); skip_direct (training, forward weights): every launch that reads this pack takes the
// Winograd-z / fp16 + MX-fp8 kernel of its shape, so the direct three-product fragments -- a third of a deep-level weight's bytes -- are not written
int conv3_sb_pack_add(SbPackBatch& b, const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, bool all_forms, hipStream_t s, bool skip_direct = false);
int conv3_sb_switch_signature();                   // the kernel-choice switches as the launches read them now (RU_WZ, RU_MX, devtools RU_WZ32): the engine records it with a
                                                  // training forward's packs and refuses a backward under another signature (packs hold only the forms that signature launches)
// the gradient-operand form of the MX product scheme (conv3_mx.hip; conv3_mx_pack.hpp MXG_*)
bool conv3_mxg_enabled();                                // RU_MXG=0: the data-gradient convolutions of the 16-channel level keep three bf16 products
bool conv3_mxg_usable(int N, int Cin, int Cout, int D, int H, int W);      // ... and the shape is one conv3_mx_kernel<GRAD> takes: the producer of the gradient asks before it writes the operand form
// fp32 voxel-major [N][1][D][H][W][16] -> the gradient operand form: per voxel 64 bytes [bf16 hi ch 0-7 | hi ch 8-15 | e4m3(lo * 2^(8-e)), e4m3(g * 2^-e) ch 0-7 | the same ch 8-15],
// e from the voxel's largest |hi| (what wgrad3_tz<1,0,3,3> publishes in the step; this launch serves the op-level entry and the tests)
int conv3_mxg_split_launch(const float* x, void* g16, size_t nvox, hipStream_t s);
bool conv3_sb_forward_skips_direct(int N, int Cin, int Cout, int D, int H, int W);     // a forward launch of this shape on activations (products == 2) takes a non-direct kernel under the current switches
int conv3_sb_pack_batch(SbPackBatch& b, hipStream_t s);
// pack [Cout][Cin][27] -> wp.  mode 0: forward; mode 1: data-gradient (taps flipped, in/out swapped:
// the packed conv maps Cout_f input channels to Cin_f output channels).
int conv3_pack_weights(const float* w, float* wp, int Cin_f, int Cout_f, int mode, hipStream_t s);
size_t conv3_packed_floats(int Cin_conv, int Cout_conv);   // for the conv as launched (after any swap)

// ------------------------------------------------------------------ weight gradients (wgrad_f32.hip)
struct Wgrad3Args {
    const float* x;          // [N][Cin][D][H][W]  (conv input; fused transform like Conv3Args)
    const float* dy;         // [N][Cout][D][H][W]
    const float* in_scale;
    const float* in_shift;
    float in_slope;
    float* dw;               // [Cout][Cin][27]  (overwritten)
    float* db;               // [Cout] or null (sum of dy)
    void* ws;                // partials
    size_t ws_bytes;
    int N, Cin, Cout, D, H, W;
    int mode;                // RU_PREC_F32 / RU_PREC_BF16X3 (split-bf16 kernel, wgrad_sb.hip; needs W % 4 == 0)
    int x_c16, dy_c16;       // voxel-major x / dy (split-bf16 kernel only); 0 = NCDHW
    int dw_cin, dw_cout;     // wgrad_tr only: real channel counts of dw when x / dy are zero-padded to 16 channels (0 = Cin / Cout)
    int dy_s16;              // wgrad_tr only: dy is C16 in split form: its staging is a plain copy
    int x_c4, dy_c4;         // wgrad_tr only: that operand is a [N][D][H][W][4] copy (pad_to_c4) standing for a 16-channel block whose channels 4..15 are zero
    // wgrad_tr only, Cout == 16: the dy operand is the GroupNorm-backward APPLY computed on the fly (`dy` is ignored):
    //     dy = cA * ((y*scale + shift) > 0 ? d : d*slope) + (cB*y + cC)
    // from the forward tensor gb_y, the gradient w.r.t. the activation gb_d (both voxel-major fp32), the GroupNorm's (scale, shift)
    // [N][Cout], the finalize coefficients gb_coef [N][Cout][3] and gb_slope -- the arithmetic of gn_bwd_apply16_launch -- and the
    // kernel also writes it to gb_out in split form for the data-gradient conv: no separate apply pass over (y, d).
    const float* gb_y;
    const float* gb_d;
    const float* gb_scale;
    const float* gb_shift;
    const float* gb_coef;
    float gb_slope;
    float* gb_out;
    int gb_g16;              // gb_out is written in the gradient-operand form of the MX scheme (mxg_hi8 / mxg_cvt8) instead of the split form: its reader is conv3_mx_kernel<GRAD>
    int products;            // wgrad_tr only: 0 / 3 = three split-bf16 products, 1 = hi*hi only (gradient precision RU_PREC_BF16), where such a variant exists
    // wgrad_tr only: the operands are EXCHANGED -- `x` (with its halo) is the convolution's OUTPUT gradient, `dy` (tile centres) its input:
    //     T[t][o'][c'] = sum_v dy[v][o'] * x[v + t][c'] = dW[26 - t][cout = c'][cin = o']
    // so the result is stored transposed with the taps mirrored (dw_cout / dw_cin then name the real channel counts of dy / x).  Used
    // for the head, whose output gradient has 3 channels: as the 4-channel x operand it takes the packed-tap form (wgrad3_tz XS == 2).
    int swapped;
    struct WgradRedList* defer;   // non-null: the reduction of the partials is queued there instead of launched (wgrad_reduce_flush)
};
size_t wgrad3_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W);     // max over both precisions
int wgrad3_launch(const Wgrad3Args& a, hipStream_t s);
// sum of `nparts` partial gradients [nparts][taps][CoP][CiP] into dw (fixed order; wgrad_f32.hip)
// Deferred form: nobody reads a weight gradient before the optimizer, so the engine's backward queues the ~35 reductions of a step
// (`defer` non-null) and runs them as ONE launch behind the last weight-gradient kernel (wgrad_reduce_flush): same kernel body, same
// summation order, bit-identical results -- 35 latency-bound launches of 5-30 us each become one that fills the chip.
struct WgradRedEntry { const float* partials; float* dw; int nparts, taps, CoP, CiP, Cout, Cin, so, sc, split, flip, lo, blk0; };
struct WgradRedList { std::vector<WgradRedEntry> e; };
int wgrad_reduce_launch(const float* partials, int nparts, int taps, int CoP, int CiP, int Cout, int Cin, float* dw, int so, int sc, int split, hipStream_t s,
                        int flip_taps = 0, WgradRedList* defer = nullptr);
int wgrad_reduce_flush(WgradRedList& l, hipStream_t s);
size_t wgrad3_sb_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W);
int wgrad3_sb_launch(const Wgrad3Args& a, hipStream_t s);
// both tensors voxel-major: transpose-read kernel (wgrad_tr.hip); workspace 0 if the channel counts do not fit
size_t wgrad3_tr_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W);
int wgrad3_tr_launch(const Wgrad3Args& a, hipStream_t s);

// 1x1x1: dw[o][c] = sum_{n,v} dy[n][o][v] * x[n][c][v]; result written to dw[o*ldw + c] (ldw >= Cin)
struct Wgrad1Args {
    const float* x;          // [N][Cin][V]
    const float* dy;         // [N][Cout][V]
    float* dw;
    int ldw;
    void* ws;
    size_t ws_bytes;
    int N, Cin, Cout;
    size_t V;
    int c16;                 // x and dy are voxel-major (C16); Cin, Cout multiples of 16
    int s2d, Dc, Hc, Wc;     // c16 only, s2d = 1: x is the FINE tensor of a 2x2x2 stride-2 conv (Cin/8 channels, extents 2Dc x 2Hc x 2Wc);
                             // input channel tap*(Cin/8) + c lives at fine voxel (2z+i, 2y+j, 2x+k); V = Dc*Hc*Wc
    int tap_split;           // > 0: the Cin index is tap*tap_split + c of a 2x2x2 conv: written to dw[o*ldw + c*8 + tap]
    const float* x1;         // c16 only, optional: input channels C0 .. Cin-1 live in this second tensor (a channel concat that was never made)
    int C0;                  // channels of x when x1 is set (multiple of 16)
    struct WgradRedList* defer;   // as Wgrad3Args::defer
    // c16 only, Cout <= 32: the DATA gradient of the same 1x1x1 conv in the same pass over dy (the dy tile of a voxel chunk is in LDS for
    // the weight gradient anyway): dx[v][c] = sum_o dg_w[o*dg_ldw + c] * dy[v][o], input channels < C0 (or all, without x1) written to
    // dg_y0, the others to dg_y1 -- the latter through the LeakyReLU-backward mask of the x1 values staged for the weight gradient
    // (dx * (x1 > 0 ? 1 : dg_mask_slope)): one read of dy and x1 instead of two.  Null dg_w: off.
    const float* dg_w;
    int dg_ldw;
    float* dg_y0;
    float* dg_y1;
    float dg_mask_slope;
};
size_t wgrad1_workspace_bytes(int N, int Cin, int Cout, size_t V);
int wgrad1_launch(const Wgrad1Args& a, hipStream_t s);

// ------------------------------------------------------------------ pointwise / small kernels (pointwise.hip)
// y[n][o][v] = act( sum_c wT[c*ldw + o] * xcat[n][c][v] ) (+ add), xcat = concat(x0[C0], x1[C1]) on channels.
struct Conv1Args {
    const float* x0; int C0;
    const float* x1; int C1;     // may be null / 0
    const float* wT;             // [C0+C1][ldw] (input-channel major)
    int ldw;
    float* y;                    // [N][Cout][V]
    const float* add;            // or null
    float out_slope;             // LeakyReLU slope on the output (1 = none)
    int N, Cout;
    size_t V;
    // conv1_16_launch only: the 2x2x2 stride-2 conv and its transpose without a space-to-depth tensor.  V = COARSE voxels
    // (Dc*Hc*Wc).  s2d = 1 (gather): x0 is the FINE tensor with C0/8 channels, input channel tap*(C0/8) + c is read from fine
    // voxel (2z+i, 2y+j, 2x+k), tap = i*4 + j*2 + k.  s2d = 2 (scatter): y is the FINE tensor with Cout/8 channels, output
    // channel tap*(Cout/8) + c is written to that fine voxel.
    int s2d, Dc, Hc, Wc;
    // conv1_16_launch only: LeakyReLU backward fused into the store: y = acc * (mask[idx] > 0 ? 1 : mask_slope), mask laid out like y
    const float* mask;
    float mask_slope;
    // conv1_16_launch only, plain mode: output channels Cout0 .. Cout-1 go to a second tensor y1 ([N][(Cout-Cout0)/16][V][16]); `mask`
    // then applies to (and is laid out like) y1 only.  Two 1x1 convs of one input in one pass.
    float* y1;
    int Cout0;
    // conv1_16_launch only (plain and scatter modes, no split output): fused GroupNorm-BACKWARD statistics, as Conv3Args::bst_* -- the
    // stored output d (mask / residual included) is the gradient w.r.t. the activation after a GroupNorm of bst_y (laid out like the
    // output tensor); the partial sums S1 = sum dh, S2' = sum dh*u (u = y*k1 + k2, dh = u > thr ? d : d*bst_slope) go to
    // stat_partials [N][C][nblk][2], nblk = conv1_16_bst_nblk(a); gn_bwd_finalize_launch(..., s2_sign = 1) takes them.
    const float* bst_y;
    const float* bst_k;          // [N][3][C] (k1, k2, thr), C = channels of the output tensor
    float bst_slope;
    float* stat_partials;
};
int conv1_16_bst_nblk(const Conv1Args& a);      // partials per (sample, channel) the launch writes; 0: this shape has no fused-statistics form
int conv1_launch(const Conv1Args& a, hipStream_t s);
int transpose_launch(const float* src, float* dst, int rows, int cols, hipStream_t s);   // dst[c][r] = src[r][c]
// NCDHW [N][C][V] <-> C16 [N][C/16][V][16] (C % 16 == 0); to_c16 = 1: src NCDHW -> dst C16, 0: the inverse
int layout_convert_launch(const float* src, float* dst, int N, int C, size_t V, int to_c16, hipStream_t s);
// NCDHW [N][C][V] with C < 16 -> one zero-padded C16 block [N][1][V][16]
int pad_to_c16_launch(const float* src, float* dst, int N, int C, size_t V, hipStream_t s);
// NCDHW [N][C][V] with C <= 4 -> zero-padded [N][V][4]
int pad_to_c4_launch(const float* src, float* dst, int N, int C, size_t V, hipStream_t s);

// space-to-depth for the 2x2x2 stride-2 conv: y[n][c*8 + (i*4+j*2+k)][z][y][x] = x[n][c][2z+i][2y+j][2x+k]
int s2d_launch(const float* x, float* y, int N, int C, int D, int H, int W, hipStream_t s);   // D,H,W = input (even)
int d2s_launch(const float* y, float* x, int N, int C, int D, int H, int W, hipStream_t s);   // inverse (x overwritten)

// GroupNorm pieces
int gn_stats_tiles(size_t V);                                                    // nblk used by gn_stats_launch
int gn_stats_launch(const float* x, float* partials, int N, int C, size_t V, hipStream_t s);   // [N][C][nblk][2]
// partials -> mean,rstd [N][G]; scale,shift [N][C] (y = x*scale+shift)
int gn_finalize_launch(const float* partials, int nblk, const float* gamma, const float* beta, float* mean, float* rstd,
                       float* scale, float* shift, int N, int C, size_t V, int G, float eps, hipStream_t s, float* bst_k = nullptr);   // bst_k: optional [N][3][C] constants for Conv3Args::bst_k
// y = (res ? res : 0) + lrelu(x*scale[n,c]+shift[n,c], slope)
int gn_apply_launch(const float* x, const float* scale, const float* shift, const float* res, float* y,
                    int N, int C, size_t V, float slope, hipStream_t s);
// backward.  reduce: per (n,c) S1 = sum dyh, S2 = sum dyh*xhat (dyh = dy * lrelu'(x*scale+shift)).
int gn_bwd_tiles(size_t V);
int gn_bwd_reduce_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* mean,
                         const float* rstd, float slope, float* partials, int N, int C, size_t V, int G, hipStream_t s);
// partials -> coefficient triples coef[N][C][3] (dx = cA*dyh + cB*x + cC) and dgamma/dbeta (overwritten or accumulated)
int gn_bwd_finalize_launch(const float* partials, int nblk, const float* gamma, const float* mean, const float* rstd,
                           float* coef, float* dgamma, float* dbeta, int N, int C, size_t V, int G, hipStream_t s, int s2_sign = 0);
// the fused statistics need the persistent kernel: true when conv3_sb_launch will use it for this shape
bool conv3_sb_bst_usable(int N, int Cout, int D, int H, int W);
int gn_bwd_apply_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* coef,
                        float slope, float* dx, int N, int C, size_t V, hipStream_t s);

int lrelu_fwd_launch(const float* x, float* y, size_t n, float slope, hipStream_t s);
int lrelu_bwd_launch(const float* y, const float* dy, float* dx, size_t n, float slope, hipStream_t s);
int sigmoid_launch(const float* x, float* y, size_t n, hipStream_t s);
int sigmoid_bwd_launch(const float* p, const float* dp, float* dz, size_t n, hipStream_t s);   // dz = dp*p*(1-p)
int add_launch(const float* a, const float* b, float* y, size_t n, hipStream_t s);
int fill_launch(float* p, float v, size_t n, hipStream_t s);
int up2_fwd_launch(const float* x, float* y, int N, int C, int D, int H, int W, hipStream_t s);
int up2_bwd_launch(const float* dy, float* dx, int N, int C, int D, int H, int W, hipStream_t s);
int bias_grad_launch(const float* dy, float* db, int N, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s);
size_t bias_grad_workspace_bytes(int N, int C, size_t V);
// dz = dp*p*(1-p) written as the zero-padded [N][V][4] copy + bias gradient, one pass (workspace as bias_grad)
int head_grad_c4_launch(const float* p, const float* dp, float* d4, float* db, int N, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s);
// the same with dp = the criterion's gradient (crit_grad_launch's arithmetic) formed in registers from (p, target, sums): dp is never written
struct CritGradArgs { const float* target; const double* sums; double count; float w_dice, w_bce, bgw, priority; };
int head_grad_c4_crit_launch(const float* p, const CritGradArgs& cg, float* d4, float* db, int N, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s);

// ------------------------------------------------------------------ the same on the voxel-major layout C16 (pointwise_c16.hip)
int gn_apply16_launch(const float* x, const float* scale, const float* shift, const float* res, float* y, int N, int C, size_t V, float slope, hipStream_t s,
                      const float* rscale = nullptr, const float* rshift = nullptr, float rslope = 1.f);   // optional: residual = lrelu(res*rscale + rshift)
int gn_bwd_tiles16(size_t V);
int gn_bwd_reduce16_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd,
                           float slope, float* partials, int N, int C, size_t V, int G, hipStream_t s, const FinTail* fin = nullptr);
// split = 1: dx is written in SPLIT form -- per voxel and 16-channel block 64 bytes = [hi bf16 ch 0-7 | hi ch 8-15 | lo ch 0-7 | lo ch 8-15]
// (hi = bf16(v), lo = bf16(v - hi)): exactly the packets the split-bf16 conv and weight-gradient kernels stage, so their producer
// waves copy instead of converting (the conversion VALU work was what bounded the weight gradient).  Only MFMA kernels read it.
int gn_bwd_apply16_launch(const float* x, const float* dy, const float* scale, const float* shift, const float* coef, float slope, float* dx,
                          int N, int C, size_t V, int split, hipStream_t s);
int up2_fwd16_launch(const float* x, float* y, int N, int C, int D, int H, int W, float slope, hipStream_t s);   // D,H,W = coarse extents; LeakyReLU(slope) on the output (1 = none)
int up2_bwd16_launch(const float* dy, float* dx, int N, int C, int D, int H, int W, hipStream_t s);
int pack_down16_launch(const float* w, float* wd, float* wdT, int Cout, int Cin, hipStream_t s);       // [Cout][Cin][8] -> [Cout][8*Cin], [8*Cin][Cout]
// Conv1Args on C16 tensors; a.wT is read as wm[Cout][C0 + C1] (row-major out x in, pitch a.ldw)
int conv1_16_launch(const Conv1Args& a, hipStream_t s);

int crit_tiles(size_t total);
int crit_sums_launch(const float* p, const float* g, double* sums, int N, int C, size_t V, float bgw, void* ws, size_t ws_bytes, hipStream_t s);
int crit_grad_launch(const float* p, const float* g, const double* sums, double count, float w_dice, float w_bce,
                     float bgw, float priority, float* dp, int N, int C, size_t V, hipStream_t s);
int crit_value_launch(const double* sums, int C, double count, double priority, double w_dice, double w_bce, double* out3, hipStream_t s);
int tta_merge_launch(const float* p, int K, unsigned flips, float* mean_out, unsigned char* mask, unsigned long long* counts, int C, int D, int H, int W, hipStream_t s);
int compose_labels_launch(const unsigned char* mask, const unsigned long long* counts, unsigned long long et_min, unsigned char* labels, size_t V, hipStream_t s);
int dice_counts_launch(const float* p, const float* g, unsigned long long* counts, int rows, size_t V, hipStream_t s);
int dice_accumulate_launch(const unsigned long long* counts, double* acc, int N, int C, int nacc, hipStream_t s);
// training input pipeline (dataloader.py:100-216)
constexpr int RU_AUG_MAXC = 8;
struct AugmentArgs {
    const float* image;          // [C][D][H][W] raw modalities
    const unsigned char* label;  // [D][H][W] values 0..3
    float* data;                 // [C][Q0][Q1][P2]
    float* target;               // [3][Q0][Q1][P2]
    int C, D, H, W;
    int lo[3], P[3];
    double scale[3];
    int flags;                   // bit 0-2: flip D, H, W; bit 3: transpose D <-> H
    float mean[RU_AUG_MAXC], istd[RU_AUG_MAXC], gain[RU_AUG_MAXC], bias[RU_AUG_MAXC];
};
size_t zscore_workspace_bytes(int C, size_t V);
int zscore_stats_launch(const float* x, double* stats, int C, size_t V, void* ws, size_t ws_bytes, hipStream_t s);
int augment_patch_launch(const AugmentArgs& a, hipStream_t s);
int adam_launch(float* w, const float* g, float* m, float* v, float* vmax, size_t n, float lr, float b1, float b2,
                float eps, float wd, int step, hipStream_t s);

}  // namespace ru
