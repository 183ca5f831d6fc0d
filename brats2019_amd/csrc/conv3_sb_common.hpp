// conv3_sb_common.hpp -- shared by the split-bf16 3x3x3 convolution units: fragment / tap conventions, LDS image geometry, the row
// epilogues, and the persistent producer/consumer kernel template conv3_sb2_kernel with its launch helpers.  The kernel is
// instantiated in two translation units (conv3_sb2_c16.hip: voxel-major in and out, the engine's variants; conv3_sb2_mixed.hip: the
// NCDHW-side variants of the op-level C-ABI and of the stem / head) so that the 20 variants compile in parallel.
#pragma once
#include "conv3_epilogue.hpp"
#include "fin_tail.hpp"
#include <stdlib.h>
#include <utility>

namespace ru {


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_KSTEPS = 14;       // ceil(27 taps / 2)

// Tap (dz*9 + dy*3 + dx, or -1 = phantom with zero weights) in K-slot `slot` (k-groups 0-1 / 2-3 of the A fragment) of K-step ks.
// The 27 taps are 9 chains (dz, dx) of three dy taps each.  K-step 3f + dy (f < 4) pairs the dy-th taps of chains 2f and 2f+1, so the
// A fragment of (K-step 3f + dy, output row i) is the fragment of (K-step 3f + dy', output row i + dy - dy'): one LDS read serves the
// three output rows that share a halo row (conv3_sb2_kernel's consumer).  The ninth chain fills K-steps 12 (dy 0, 1) and 13 (dy 2, -).
__host__ __device__ constexpr int sb_tap(int ks, int slot) {
    if (ks < 12) {
        const int ch = 2 * (ks / 3) + slot, dy = ks % 3;
        return (ch / 3) * 9 + dy * 3 + ch % 3;
    }
    const int dy = (ks - 12) * 2 + slot;
    return dy < 3 ? 2 * 9 + dy * 3 + 2 : -1;
}

// HEAD form (at most 4 output channels, one input chunk, NCDHW output: conv_output, model.py:348).  A 16-wide N tile for 3 channels wastes 13/16 of the
// matrix work, so the N columns carry (dy, cout) pairs: column n = 4 dy + co.  The fragment F(f, r) of halo row r then feeds, in ONE MFMA per product,
// the three output rows r, r-1, r-2 -- 5 fragments x 10 halo rows x 3 products = 150 MFMAs per wave and item instead of 336 -- into the accumulator of
// HALO row r; an output row is the sum of three accumulators' column groups, taken across lanes in the epilogue.  K-step f (f < 4) holds chains 2f and
// 2f+1 as in sb_tap; K-step 4 the ninth chain in slot 0 (slot 1 zero).  The fragments sit behind the direct (and Winograd-z) ones of the same weight.
constexpr int SB_HEAD_KSTEPS = 5;
// lane l takes the value of lane l + SH of its 16-lane row (zero past the row's end): a DPP row shift -- one VALU move, no LDS.  The value goes through a
// scalar argument on purpose: __builtin_bit_cast applied to a vector ELEMENT (v[e]) reads element 0 whatever the index (hipcc 7.2).
template <int SH>
__device__ __forceinline__ float sb_row_shl(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + SH, 0xf, 0xf, true));
}
__host__ __device__ constexpr bool sb_head_shape(int Cin_conv, int Cout_conv) { return Cout_conv <= 4 && Cin_conv <= 16; }
__host__ __device__ constexpr int sb_head_tap(int f, int slot, int dy) {
    if (dy > 2) return -1;
    if (f < 4) return sb_tap(3 * f + dy, slot);
    return slot == 0 ? 2 * 9 + dy * 3 + 2 : -1;
}

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

// v = hi + lo with hi = bf16_rne(v), lo = bf16_rne(v - hi) (split_pair in ru_common.h: 3 VALU per value)
__device__ __forceinline__ void split8(const float (&t)[8], u32x4& hi, u32x4& lo) {
    split_n<4>(t, hi, lo);
}

// Epilogue of ONE M-tile (output row yy of plane zz), issued as soon as the tile's last MFMA is queued so the stores
// trickle out under the next tiles' matrix work instead of in one burst per item (a burst of 32 KB per CU from every CU
// at once is HBM-write bound and used to block the wave at store issue).
//   NCDHW output (OUT16 = false): lane = (cout co0 + (l&15), 4 consecutive x at x0 + 4*(l>>4)); NS = 1 statistics pair
//   C16 output   (OUT16 = true) : the MFMA ran with swapped operands, D[m = cout][n = voxel]: lane = (voxel x0 + (l&15),
//                                 4 consecutive couts co0 + 4*(l>>4)): one aligned float4 of the voxel-major tensor; NS = 4
struct SbOut {
    size_t base;         // float index of this lane's element in row y = 0 of plane zz; row y is base + y*rs (one v_mad_u64_u32: the
    unsigned rs;         // epilogue's VALU instructions sit between the consumer's MFMAs, so they are kept few)
    bool ok;             // lane-level validity (z, x, cout)
    float4 bias;
};
template <bool OUT16>
__device__ __forceinline__ SbOut sb_out_prepare(const Conv3Args& a, int n, int zz, int x0, int cog, int lane) {
    SbOut o;
    const int D = a.D, H = a.H, W = a.W;
    const int zc = zz < D ? zz : 0;
    if constexpr (OUT16) {
        const int xx = x0 + (lane & 15), cq = 4 * (lane >> 4);
        o.base = ((((size_t)(n * (a.Cout >> 4) + cog) * D + zc) * H) * W + xx) * 16 + cq;
        o.rs = (unsigned)W * 16u;
        o.ok = zz < D && xx < W;
        o.bias = a.bias ? *reinterpret_cast<const float4*>(a.bias + cog * 16 + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const int co = cog * 16 + (lane & 15);
        const int xx = x0 + (lane >> 4) * 4;
        const int cc = co < a.Cout ? co : 0;
        o.base = (((size_t)n * a.Cout + cc) * D + zc) * (size_t)H * W + xx;
        o.rs = (unsigned)W;
        o.ok = zz < D && co < a.Cout && xx < W;              // W % 4 == 0: the 4 voxels are in or out together
        const float b = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        o.bias = make_float4(b, b, b, b);
    }
    return o;
}
template <bool OUT16>
__device__ __forceinline__ size_t sb_out_index(const Conv3Args& a, const SbOut& o, int yy) {
    (void)a;
    return o.base + (size_t)(unsigned)yy * o.rs;
}
template <bool OUT16, int NS>
__device__ __forceinline__ void sb_out_tile(const Conv3Args& a, const SbOut& o, int yy, f32x4 v, const float4& radd, f32x4& s1, f32x4& s2) {
    if (!(o.ok && yy < a.H)) return;
    // straight-line: bias and residual are zeros when absent and the statistics are always taken -- each wave-uniform `if (a.x)` here
    // was a branch (plus phi moves) in the consumer's instruction stream between its MFMAs, ~375 cycles per tile row
    v += f32x4{o.bias.x, o.bias.y, o.bias.z, o.bias.w} + f32x4{radd.x, radd.y, radd.z, radd.w};
    if constexpr (OUT16) {
        s1 += v;
        s2 += v * v;
    } else {
        s1[0] += (v[0] + v[1]) + (v[2] + v[3]);
        s2[0] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    if constexpr (!OUT16) {                                 // only the head has an activation, and its output is NCDHW (checked at launch)
        if (a.sigmoid) {
            // hardware exp2 / rcp (1 ulp each): the library expf is ~14 VALU instructions per value, and they sit in the consumer
            // wave's instruction stream between its MFMAs (the head conv was 75 us slower than the same conv without sigmoid)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v[r]));
        }
    }
    *reinterpret_cast<float4*>(a.y + sb_out_index<OUT16>(a, o, yy)) = make_float4(v[0], v[1], v[2], v[3]);
}
// The same for the persistent kernel, with what a row needs decided at COMPILE time: every VALU instruction of the epilogue sits in the
// consumer's stream between its MFMAs (~25 cycles each there), so a voxel-major row without residual is statistics + store and
// nothing else.  The bias exists for NCDHW output only (the head conv; voxel-major output with a bias takes the one-stage kernel).
#ifdef RU_SB2_DBG
constexpr int kSb2RowDbg = RU_SB2_DBG;      // ablation builds only: 256 = no statistics math, 512 = no store instruction
#else
constexpr int kSb2RowDbg = 0;
#endif
// NT: the row goes out with a nontemporal store -- the 16-channel level, whose 0.5 GB tensors nobody reads while they could still sit in L2 / MALL (the
// deep levels' outputs ARE read again from there: profiles/r05_notes.txt, section 20)
template <bool OUT16, bool HAS_R, bool NT = false>
__device__ __forceinline__ void sb2_out_row(const Conv3Args& a, const SbOut& o, int yy, f32x4 v, const float4& radd, f32x4& s1, f32x4& s2) {
    if (!(o.ok && yy < a.H)) return;
    if constexpr (!OUT16) v += f32x4{o.bias.x, o.bias.y, o.bias.z, o.bias.w};
    if constexpr (HAS_R) v += f32x4{radd.x, radd.y, radd.z, radd.w};
    if constexpr (OUT16 && (kSb2RowDbg & 256)) {
        s1[0] += v[0];                            // keeps the accumulator alive with one VALU instruction
    } else if constexpr (OUT16) {
        s1 += v;
        s2 += v * v;
    } else {
        s1[0] += (v[0] + v[1]) + (v[2] + v[3]);
        s2[0] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        if (a.sigmoid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v[r]));
        }
    }
    if constexpr ((kSb2RowDbg & 512) != 0) { s2[0] += v[1] + v[2] + v[3]; return; }
    if constexpr (OUT16 && NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.y + sb_out_index<OUT16>(a, o, yy)));
    else *reinterpret_cast<float4*>(a.y + sb_out_index<OUT16>(a, o, yy)) = make_float4(v[0], v[1], v[2], v[3]);
}
// Conv3Args::bst_*: this conv's output d is the gradient w.r.t. the activation after GroupNorm(y); the row is stored unchanged and the
// GroupNorm-backward sums are taken on the way: u = y*k1 + k2 (= sign(gamma)*xhat), dh = u > thr ? d : d*slope, S1 += dh, S2' += dh*u
template <bool NT = false>
__device__ __forceinline__ void sb_out_tile_bst(const Conv3Args& a, const SbOut& o, int yy, f32x4 v, const float4& yv, const f32x4 (&kc)[3], float slope,
                                                f32x4& s1, f32x4& s2, const float4* radd = nullptr) {
    if (!(o.ok && yy < a.H)) return;
    if (radd) v += f32x4{radd->x, radd->y, radd->z, radd->w};          // residual first: the sums are those of the STORED gradient
    const f32x4 u = f32x4{yv.x, yv.y, yv.z, yv.w} * kc[0] + kc[1];
    const f32x4 vs = v * slope;
    f32x4 dh;
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = u[r] > kc[2][r] ? v[r] : vs[r];
    s1 += dh;
    s2 += dh * u;
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.y + sb_out_index<true>(a, o, yy)));
    else *reinterpret_cast<float4*>(a.y + sb_out_index<true>(a, o, yy)) = make_float4(v[0], v[1], v[2], v[3]);
}
// Statistics partials: ONE per (workgroup, sample), [N][Cout][nblk][2].  A consumer wave folds its lanes and leaves its 16 channels'
// (sum, sum2) in an LDS scratch row (sb_stats_to_lds: sc = this wave's 32 floats); after a workgroup barrier one wave adds the four
// rows in wave order and publishes the pair (sb_stats_commit) -- a quarter of the per-wave partials the finalize used to read.
constexpr int SB_STAT_LDS_FLOATS = 2 * 4 * 32;             // two generations (a flush may follow a flush one item later) x 4 waves x 16 channels x 2
template <bool OUT16>
__device__ __forceinline__ void sb_stats_to_lds(f32x4& s1, f32x4& s2, float* sc, int lane) {
    if constexpr (OUT16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o); s2[r] += __shfl_xor(s2[r], o); }
        }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[(4 * (lane >> 4) + r) * 2] = s1[r]; sc[(4 * (lane >> 4) + r) * 2 + 1] = s2[r]; }
        }
    } else {
        s1[0] += __shfl_xor(s1[0], 16); s2[0] += __shfl_xor(s2[0], 16);
        s1[0] += __shfl_xor(s1[0], 32); s2[0] += __shfl_xor(s2[0], 32);
        if (lane < 16) { sc[lane * 2] = s1[0]; sc[lane * 2 + 1] = s2[0]; }
    }
}
// lanes 0..15 of one wave; sc4 = the four waves' rows of one generation (null: zeros -- a sample this workgroup never touched)
__device__ __forceinline__ void sb_stats_commit(const Conv3Args& a, const float* sc4, int n, int cog, int blk, int nblk, int lane) {
    const int co = cog * 16 + lane;
    if (lane < 16 && co < a.Cout) {
        float u1 = 0.f, u2 = 0.f;
        if (sc4) {
#pragma unroll
            for (int w = 0; w < 4; ++w) { u1 += sc4[w * 32 + lane * 2]; u2 += sc4[w * 32 + lane * 2 + 1]; }
        }
        stat_publish(a.stat_partials + (((size_t)n * a.Cout + co) * nblk + blk) * 2, u1, u2);
    }
}

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {           // f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}


struct SBChoice { int tz, ty; };
static inline int sb_ncu() {                                  // CUs of the CURRENT device (cached per device id)
    static int ncu[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (ncu[dev] == 0) {
        int v = 0;
        ncu[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return ncu[dev];
}
static inline SBChoice sb_choose(int N, int Cout, int D, int H, int W) {
    const int ncog = cdiv(Cout, 16);
    auto blocks = [&](int tz, int ty) { return (long)N * cdiv(D, tz) * cdiv(H, ty) * cdiv(W, 16) * ncog; };
    // the persistent kernel addresses a 16-channel block of a voxel-major input through a buffer descriptor with 32-bit byte offsets
    // (its out-of-range lanes are the zero padding): volumes of 2^25 voxels and more (e.g. 336^3) take the one-stage kernel
    const bool fits32 = (size_t)D * H * W * 64 < ((size_t)1 << 31);
    if (fits32 && blocks(4, 8) >= 256) return {4, 8};   // one persistent producer/consumer workgroup per CU is enough (deep levels: 8 chunks per tile)
    if (blocks(2, 8) >= 1024) return {2, 8};
    // fewer (2,4,16) tiles x cout groups than TWO per CU (the 128-channel level of a batch-1 forward: 256): the one-stage kernel walks its
    // input-channel chunks with every chunk's weight-fragment latency exposed and only a second resident workgroup can hide it -- half-size tiles
    static const bool no22 = [] { const char* e = getenv("RU_SB1_NO22"); return e && *e == '1'; }();      // (tools: A/B of the half-size tile)
    if (!no22 && blocks(2, 4) < 2 * sb_ncu() && blocks(2, 2) >= sb_ncu()) return {2, 2};
    return {2, 4};
}

// v2 (persistent producer/consumer) handles the large-tile case
static inline bool sb_use_v2(const SBChoice& c) { return c.tz == 4 && c.ty == 8; }

// workgroups along x of the persistent kernel: one resident workgroup per CU in total
static inline long sb2_grid_x(int N, int Cout, int D, int H, int W) {
    const int ncu = sb_ncu(), ncog = cdiv(Cout, 16);
    const long ntile = (long)N * cdiv(D, 4) * cdiv(H, 8) * cdiv(W, 16);
    long gx = ncu / (ncog < ncu ? ncog : ncu);
    if (gx < 1) gx = 1;
    return gx > ntile ? ntile : gx;
}


// ------------------------------------------------------------------ v2: persistent producer / consumer workgroups
// 512 threads: waves 0-3 are CONSUMERS (A fragments from LDS, weights in registers, 3 MFMAs per K-step, epilogue),
// waves 4-7 are PRODUCERS (global float4 loads -> fused affine + LeakyReLU -> hi/lo split -> transposed LDS image).
// A workgroup walks a contiguous run of tiles (halo re-reads stay in its XCD's L2); the LDS image is double buffered:
// while the consumers are on item w the producers finish item w+1 in the other buffer and already have the global
// loads of item w+2 in flight.  One __syncthreads per item.  Each SIMD hosts one consumer and one producer wave, so
// the matrix pipe and the VALU/LDS-store work of the staging overlap instead of alternating.
// Per-tile GroupNorm statistics are written per consumer WAVE (no cross-wave reduction -> no extra barrier).
// RU_SB2_DEBUG bit 64: consumer wave 0 of every workgroup brackets the sections of its item loop with s_memtime and adds the sums
// here (cycles): [0] index math before group 0, [1] group 0, [2] between the groups, [3] group 1, [4] barrier, [5] items, [6] drain, [7] workgroups
static __device__ unsigned long long sb2_prof[8];     // (one per translation unit; read by ru_dbg_sb2_prof of the voxel-major unit)
// MULTI (more than one 16-channel input chunk): the weight fragments of the NEXT item's chunk are fetched K-step by K-step into
// the registers group 1 has just finished with, instead of 28 loads at the start of every item with the matrix pipe waiting on
// the first (that exposed L2 latency was ~20 % of the kernel at 32..128 channels).
// BST: fused GroupNorm-backward statistics in the epilogue (Conv3Args::bst_*), C16 output only.
// ADD: a residual tensor is added in the epilogue (Conv3Args::add).  Compile-time, like BST: without a per-row operand the consumer's
// stream holds NO loads, so its s_waitcnt vmcnt never has to wait for older row stores to be acknowledged (vmcnt counts in order).
// NP: MFMA products per operand pair.  3 = split-bf16 (hi*hi + lo*hi + hi*lo, ~2^-17 relative).  1 = plain bf16 operands (hi*hi only,
// fp32 accumulate): the gradient precision RU_PREC_BF16 of the engine's backward -- the staging writes and the consumers read the hi
// planes only, the weights' lo fragments are never fetched.  Voxel-major input only.
template <int TZ, int TY, bool IN16, bool OUT16, bool MULTI, bool BST, bool ADD, int NP = 3, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv3_sb2_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx, int nchunk, int dbg_arg) {
    // dbg (RU_SB2_DEBUG, ablation only; results are wrong when set): 1 = producers skip transform/split/LDS store,
    // 2 = producers skip global loads, 4 = consumers skip the MFMAs, 8 = consumers skip the epilogue.
    // COMPILE-TIME only: a -DRU_SB2_DBG=<bits> build (python -m brats2019_amd.build --dbg <bits> -> lib/libresunet_hip_dbg<bits>.so,
    // for tools/) has them; the product library compiles every switch out.  (A runtime switch put a branch around every unrolled step
    // of the consumer and the ablation then timed different code.)
#ifdef RU_SB2_DBG
    constexpr int dbg = RU_SB2_DBG;
#else
    constexpr int dbg = 0;
#endif
    (void)dbg_arg;
    static_assert(NP == 3 || (NP == 1 && IN16), "one-product variant: voxel-major input only");
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
    const int cog = blockIdx.y;
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
    // z-walk order (when the shape allows): the G/8 workgroups of an XCD own a fixed panel of G/8 (ty, tx) tile positions of one
    // sample and walk DOWN z with it, step by step, then take their next panel.  The two z-halo planes a tile shares with the tile
    // above it were loaded by the same XCD one step earlier and are still in its L2 (a step stages ~2 MB per XCD); in the plain order
    // the z neighbour belongs to another XCD in the same step and both L2s fetch the planes.
    const int tiles_xy = nty * ntx, Pn = G / 8;
    const bool zwalk = !(dbg & 256) && (G % 8 == 0) && Pn > 0 && (tiles_xy % Pn == 0) && ((a.N * (tiles_xy / Pn)) % 8 == 0);
    const int zw_pps = zwalk ? tiles_xy / Pn : 1;        // panels per sample
    const int zw_xcd = blockIdx.x % 8, zw_j = blockIdx.x / 8;
    const int nsteps = zwalk ? (a.N * zw_pps / 8) * ntz : (swz < ntile ? (ntile - swz + G - 1) / G : 0);
    const int nitems = nsteps * nchunk;
    auto tile_of = [&](int step) {
        if (!zwalk) return t_begin + step * G;
        const int q = step / ntz, tz = step - q * ntz;
        const int panel = zw_xcd + 8 * q;
        const int n = panel / zw_pps, pb = panel - n * zw_pps;
        return (n * ntz + tz) * tiles_xy + pb * Pn + zw_j;
    };
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
        if constexpr ((dbg & 65536) != 0) __builtin_amdgcn_s_setprio(3);     // devtools: static priority for the staging waves
        // ---------------------------------------------------------------- producers
        // Work items of one (tile, 16-channel chunk):
        //   interior: halo row x 4 aligned float4 segments [x0+4q, +4)  -> NROW*4 items, every lane has 4 valid voxels
        //   edge    : halo row x {x0-1, x0+16}                           -> NROW*2 items, one scalar per channel
        // (the six-segment cover of the 18-wide row wasted 6 of 24 loaded floats and half of the lanes' VALU work)
        if constexpr (!IN16) {
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
            const int tile = tile_of(item / nchunk), chunk = item % nchunk;
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
        // ---- C16 input: position p of the halo image (row-major, HX voxels per row) x channel half; the 8 lanes of a
        // ds_write_b128 group hold 8 consecutive positions of one half (conflict free), and every voxel is two aligned
        // float4 loads: a halo row is ONE contiguous run of 18 x 64 bytes
        constexpr int NPOS = NROW * HX, NR = (NPOS + 127) / 128;
        const int hsel = (ptid >> 3) & 1;
        const int pslot = (ptid >> 4) * 8 + (ptid & 7);
        float4 v16[NR][2];
        // head form: the residual of the input (Conv3Args::in_res), same positions.  A RING of RD rounds, requested inside store() RD rounds ahead of their
        // use: a full item of them in flight beside v16 (72 + 72 registers) spilled the staging waves (head conv 493 -> 890 us at 8 x 128^3)
        constexpr int RD = 4;
        float4 r16[HEAD ? RD : 1][2];
        const bool res = HEAD && a.in_res != nullptr;
        __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0, 0x00020000);      // of the item whose loads are in the registers
        __amdgpu_buffer_rsrc_t out_rs = res_rs;
        const bool sum_out = HEAD && res && a.in_sum_out != nullptr;
        int res_base = 0;
        float4 sc4[2], sh4[2];
        unsigned vmask = 0;
        // Per thread and round the halo position p = r*128 + pslot is FIXED for the whole kernel: its coordinates (hz, hy, xc) and its byte
        // offset inside a 16-channel block relative to the tile's halo origin are computed once.  Per item a round then costs three adds,
        // three compares, one add and one select (the index arithmetic -- two divisions by constants and two 64-bit multiply-adds per
        // round -- was a third of the staging waves' VALU instructions, and next to a wave that issues MFMAs back to back a VALU
        // instruction of the other wave gets ONE issue slot per MFMA: tools/coissue_probe.hip).
        const bool s16 = a.in_s16 != 0;                  // split form in HBM (gn_bwd_apply16_launch): the staging is a plain copy of hi / lo packets
        const unsigned lofs = (s16 ? hsel * 4 : hsel * 8) * 4, second = s16 ? 32u : 16u;
        int pk[NR], dlt[NR];                             // (hz | hy << 8 | xc << 16), byte offset of (hz, hy, xc) + this lane's half
        // head form, Conv3Args::in_sum_out: which of this thread's positions it WRITES -- bit r of own_in: round r is a voxel of the tile itself; of own_nx:
        // a voxel of the NEXT tile down z whose image rows the z-walk copies from this item's planes 4 / 5 instead of converting them again
        unsigned own_in = 0, own_nx = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int p = r * 128 + pslot;
            const int row = p / HX, xc = p - row * HX;
            const int hz = row / HY, hy = row - hz * HY;
            pk[r] = hz | (hy << 8) | (xc << 16);
            dlt[r] = ((hz * H + hy) * W + xc) * 64 + (int)lofs;
            const bool inyx = hy >= 1 && hy <= TY && xc >= 1 && xc <= 16 && p < NPOS;
            own_in |= (inyx && hz >= 1 && hz <= TZ) ? (1u << r) : 0u;
            own_nx |= (inyx && hz == TZ + 1 && p - TZ * HY * HX < ((2 * HY * HX) / 128) * 128) ? (1u << r) : 0u;
        }
        const bool plast = (NR - 1) * 128 + pslot < NPOS;    // the last round covers positions beyond the image
        // z-walk order, one chunk: the tile of this step sits right below the previous one, so its halo planes 0 and 1 ARE planes 4 and 5
        // of the image staged one item earlier (already transformed and split, same y/x zero padding): the first CR rounds (positions
        // < CR*128 <= 2 planes) are copied LDS -> LDS from the other buffer instead of being loaded and converted again
        constexpr int CR = (2 * HY * HX) / 128;
        bool st_chain = false;                           // of the item whose loads are in the registers
        bool nx_chain = false;                           // ... and whether the item after it will copy its planes 0 / 1 from this one's 4 / 5
        auto issue = [&](int item) {
            if (dbg & 2) return;
            const int step = item / nchunk;
            const int tile = tile_of(step), chunk = item % nchunk;
            int n, z0, y0, x0, tis;
            tile_origin(tile, n, z0, y0, x0, tis);
            // one buffer descriptor per (sample, chunk): a position outside the volume gets an offset beyond num_records and the load
            // returns zeros -- the zero padding costs no VALU select per value, and the address is one 32-bit offset per lane
            const float* xb = a.x + ((size_t)(n * nchunk + chunk) * DHW) * 16;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(DHW * 64), 0x00020000);
            const int zm1 = z0 - 1, ym1 = y0 - 1, xm1 = x0 - 1;
            const int base = ((zm1 * H + ym1) * W + xm1) * 64;            // may be negative: only used where the position is inside
            vmask = 0;
            if constexpr (HEAD) {
                if (res) {
                    res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in_res + ((size_t)(n * nchunk + chunk) * DHW) * 16), 0, (int)(DHW * 64), 0x00020000);
                    if (sum_out) out_rs = __builtin_amdgcn_make_buffer_rsrc(a.in_sum_out + ((size_t)(n * nchunk + chunk) * DHW) * 16, 0, (int)(DHW * 64), 0x00020000);
                    res_base = base;
                }
            }
            st_chain = zwalk && !MULTI && !(dbg & 512) && CR > 0 && (step % ntz) != 0;
            nx_chain = zwalk && !MULTI && !(dbg & 512) && CR > 0 && ((step + 1) % ntz) != 0;
            auto ld_round = [&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value;
                const int gz = zm1 + (pk[r] & 0xff), gy = ym1 + ((pk[r] >> 8) & 0xff), gx = xm1 + ((pk[r] >> 16) & 0xff);
                bool ok = ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);     // (no short-circuit branches)
                if constexpr ((r + 1) * 128 > NPOS) ok = ok & plast;
                // (devtools bit 25: every staging load of the voxel-major path hits one cache line -- results wrong -- is memory latency / bandwidth on the critical path?)
                const unsigned ofs = ok ? ((dbg & (1 << 25)) ? 0u : (unsigned)(base + dlt[r])) : 0x80000000u;
                vmask |= ok ? (1u << r) : 0u;
                v16[r][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ofs, 0, 0));
                if (NP == 3 || !s16) v16[r][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ofs, second, 0));   // (split form, one product: the lo packet is not read)
            };
            if (st_chain) static_for<NR - CR>([&](auto R) { ld_round(std::integral_constant<int, decltype(R)::value + CR>{}); });    // one wave-uniform branch
            else static_for<NR>(ld_round);
            if (xform) {
                const int cofs = n * a.Cin + chunk * 16 + hsel * 8;
                sc4[0] = *reinterpret_cast<const float4*>(a.in_scale + cofs); sc4[1] = *reinterpret_cast<const float4*>(a.in_scale + cofs + 4);
                sh4[0] = *reinterpret_cast<const float4*>(a.in_shift + cofs); sh4[1] = *reinterpret_cast<const float4*>(a.in_shift + cofs + 4);
            }
        };
        auto store = [&](int item, u32x4* buf) {
            (void)item;
            if (dbg & 1) {
                if (!(dbg & 2)) {
                    float acc0 = 0.f;
#pragma unroll
                    for (int r = 0; r < NR; ++r) acc0 += v16[r][0].x + v16[r][1].x;
                    if (acc0 == 12345.678f) buf[0] = u32x4{1u, 2u, 3u, 4u};
                }
                return;
            }
            float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (xform) {
                sc[0] = sc4[0].x; sc[1] = sc4[0].y; sc[2] = sc4[0].z; sc[3] = sc4[0].w; sc[4] = sc4[1].x; sc[5] = sc4[1].y; sc[6] = sc4[1].z; sc[7] = sc4[1].w;
                sh[0] = sh4[0].x; sh[1] = sh4[0].y; sh[2] = sh4[0].z; sh[3] = sh4[0].w; sh[4] = sh4[1].x; sh[5] = sh4[1].y; sh[6] = sh4[1].z; sh[7] = sh4[1].w;
            }
            if (st_chain) {                              // planes 0, 1 <- planes 4, 5 of the other buffer (complete since the last barrier)
                const u32x4* prev = buf == lds ? lds + BUF : lds;
                u32x4 ch[CR > 0 ? CR : 1][2];
#pragma unroll
                for (int r = 0; r < CR; ++r) {
                    const int p = r * 128 + pslot;
                    ch[r][0] = prev[hsel * HVOLP + p + 4 * HY * HX];
                    if constexpr (NP == 3) ch[r][1] = prev[(2 + hsel) * HVOLP + p + 4 * HY * HX];
                }
#pragma unroll
                for (int r = 0; r < CR; ++r) {
                    const int p = r * 128 + pslot;
                    buf[hsel * HVOLP + p] = ch[r][0];
                    if constexpr (NP == 3) buf[(2 + hsel) * HVOLP + p] = ch[r][1];
                }
            }
            // ONE wave-uniform dispatch per item, then a branch-free unrolled loop: mode 0 plain, 1 fused transform, 2 split-form copy
            auto body = [&](auto MODE, auto R0) {
                constexpr int mode = decltype(MODE)::value;
                constexpr int r0 = decltype(R0)::value;
                auto res_load = [&](auto RR) __attribute__((always_inline)) {           // round rr of the item in the registers -> ring slot (rr - r0) % RD
                    constexpr int rr = decltype(RR)::value;
                    if constexpr (mode == 3 && rr < NR) {
                        const unsigned ofs = ((vmask >> rr) & 1u) ? (unsigned)(res_base + dlt[rr]) : 0x80000000u;
                        r16[(rr - r0) % RD][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, ofs, 0, 0));
                        r16[(rr - r0) % RD][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, ofs, 16, 0));
                    }
                };
                if constexpr (mode == 3) static_for<RD>([&](auto J) { res_load(std::integral_constant<int, r0 + decltype(J)::value>{}); });
                static_for<NR - r0>([&](auto RI) __attribute__((always_inline)) {
                    constexpr int r = r0 + decltype(RI)::value;
                    constexpr int rslot = (r - r0) % RD;
                    [&]() __attribute__((always_inline)) {
                    const int p = r * 128 + pslot;
                    if ((r + 1) * 128 > NPOS && p >= NPOS) return;
                    u32x4 hi, lo = u32x4{0u, 0u, 0u, 0u};
                    if constexpr (mode == 2) {                               // positions outside the volume were loaded as zeros
                        hi = __builtin_bit_cast(u32x4, v16[r][0]);
                        if constexpr (NP == 3) lo = __builtin_bit_cast(u32x4, v16[r][1]);
                    } else {
                        const float f[8] = {v16[r][0].x, v16[r][0].y, v16[r][0].z, v16[r][0].w, v16[r][1].x, v16[r][1].y, v16[r][1].z, v16[r][1].w};
                        float t[8];
                        if constexpr (mode == 1 || mode == 3) {
                            // the zero padding applies to the ACTIVATED tensor: lanes outside the volume skip the arithmetic under the
                            // exec mask and store zeros (plain VALU only here -- packed-f32 instructions starve beside the MFMA waves)
                            if (!((vmask >> r) & 1u)) {
                                const u32x4 z = u32x4{0u, 0u, 0u, 0u};
                                buf[hsel * HVOLP + p] = z;
                                if constexpr (NP == 3) buf[(2 + hsel) * HVOLP + p] = z;
                                return;
                            }
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const float u = fmaf(f[c], sc[c], sh[c]);
                                t[c] = fmaxf(u, u * slope);
                            }
                            if constexpr (mode == 3) {                       // + the residual of the input (the last Residual block's x: model.py:114)
                                const float g[8] = {r16[rslot][0].x, r16[rslot][0].y, r16[rslot][0].z, r16[rslot][0].w, r16[rslot][1].x, r16[rslot][1].y, r16[rslot][1].z, r16[rslot][1].w};
#pragma unroll
                                for (int c = 0; c < 8; ++c) t[c] = g[c] + t[c];      // (gn_apply16_kernel's order: x + activation)
                                if (sum_out) {                               // training: the block output the backward reads, every voxel from exactly one thread
                                    const bool mine = ((own_in >> r) & 1u) || (nx_chain && ((own_nx >> r) & 1u));
                                    const unsigned ofs = mine ? (unsigned)(res_base + dlt[r]) : 0x80000000u;
                                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(t[0], t[1], t[2], t[3])), out_rs, ofs, 0, 0);
                                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(t[4], t[5], t[6], t[7])), out_rs, ofs, 16, 0);
                                }
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < 8; ++c) t[c] = f[c];
                        }
                        if constexpr (NP == 3) {
                            split8(t, hi, lo);
                        } else {                                             // hi = bf16_rne(v) only: one v_cvt_pk_bf16_f32 per two values
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                ru_bf16x2 h2;
                                h2[0] = (__bf16)t[2 * c];
                                h2[1] = (__bf16)t[2 * c + 1];
                                hi[c] = __builtin_bit_cast(unsigned, h2);
                            }
                        }
                    }
                    buf[hsel * HVOLP + p] = hi;
                    if constexpr (NP == 3) buf[(2 + hsel) * HVOLP + p] = lo;
                    }();
                    res_load(std::integral_constant<int, r + RD>{});             // into the slot this round has just read
                });
            };
            auto dispatch = [&](auto R0) __attribute__((always_inline)) {
                if constexpr (HEAD) {
                    if (res) { body(std::integral_constant<int, 3>{}, R0); return; }       // (launch check: in_res comes with the fused transform)
                }
                if (s16) body(std::integral_constant<int, 2>{}, R0);
                else if (xform) body(std::integral_constant<int, 1>{}, R0);
                else body(std::integral_constant<int, 0>{}, R0);
            };
            if (st_chain) dispatch(std::integral_constant<int, CR>{});
            else dispatch(std::integral_constant<int, 0>{});
        };
        // ---- split-form input, multi-chunk shapes (the deep-level data-gradient convs): the staging is a COPY of 16-byte packets, so it goes
        // global -> LDS directly (`buffer_load_dwordx4 ... lds`: lane l's 16 bytes land at M0 + 16 l; an out-of-range offset writes the zero
        // padding) -- no VGPR round trip, no ds_write, 17 instructions per staging wave and item instead of 18 loads + 36 LDS stores.  Staging
        // wave q owns plane q of the image (hi / lo x channel half); chunk c of a plane = its positions 64 c .. 64 c + 63 (the last chunk's
        // lanes beyond the image fall into the padding of the plane, HVOLP - NPOS = 8 packets).  The image of item w+1 is requested when
        // item w starts (its buffer was last read during item w-1) and has landed -- vmcnt(0), then the barrier -- before anybody reads it.
        constexpr bool kDma = MULTI && !(dbg & 131072);
        if (kDma && s16) {
            constexpr int NCH = (NPOS + 63) / 64;
            static_assert(NCH * 64 <= HVOLP, "the last chunk must stay inside the plane's padding");
            const int plane = rw;                                        // 0: hi half 0, 1: hi half 1, 2: lo half 0, 3: lo half 1
            const bool live_plane = NP == 3 || plane < 2;                // one product: the lo planes are never read
            int dpk[NCH], ddl[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int p = c * 64 + lane;
                const int row = p / HX, xc = p - row * HX;
                const int hz = row / HY, hy = row - hz * HY;
                dpk[c] = (p < NPOS) ? (hz | (hy << 8) | (xc << 16)) : 0x00ffffff;       // (beyond the image: coordinates that fail every test)
                ddl[c] = ((hz * H + hy) * W + xc) * 64 + plane * 16;
            }
            auto dma = [&](int item, u32x4* buf) __attribute__((always_inline)) {
                if (!live_plane) return;
                const int step = item / nchunk;
                const int tile = tile_of(step), chunk = item % nchunk;
                int n, z0, y0, x0, tis;
                tile_origin(tile, n, z0, y0, x0, tis);
                const float* xb = a.x + ((size_t)(n * nchunk + chunk) * DHW) * 16;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(DHW * 64), 0x00020000);
                const int zm1 = z0 - 1, ym1 = y0 - 1, xm1 = x0 - 1;
                const int base = ((zm1 * H + ym1) * W + xm1) * 64;
                u32x4* dst = buf + plane * HVOLP;
                static_for<NCH>([&](auto C) __attribute__((always_inline)) {
                    constexpr int c = decltype(C)::value;
                    const int gz = zm1 + (dpk[c] & 0xff), gy = ym1 + ((dpk[c] >> 8) & 0xff), gx = xm1 + ((dpk[c] >> 16) & 0xff);
                    const bool ok = ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                    const unsigned ofs = ok ? (unsigned)(base + ddl[c]) : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + c * 64), 16, ofs, 0, 0, 0);
                });
            };
            if (nitems > 0) dma(0, lds);
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0), expcnt and lgkmcnt left alone: the image has landed in LDS
            __syncthreads();
            for (int w = 0; w < nitems; ++w) {
                if (w + 1 < nitems) dma(w + 1, lds + ((w + 1) & 1) * BUF);
                __builtin_amdgcn_s_waitcnt(0x0f70);
                __syncthreads();
            }
        } else {
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
        }
        }
        __syncthreads();                                // (the consumers' closing barrier: their last statistics flush)
    } else {
        // ---------------------------------------------------------------- consumers
        // Wave rw owns output plane z0 + rw: its 8 M-tiles are the 8 rows of that plane (tile i = row y0 + i), and the A fragment of
        // (tile i, tap (dz, dy, dx)) sits at halo row i + dy of halo plane rw + dz.  The K-steps pair taps so that the SAME fragment
        // serves three tiles (sb_tap): K-step 3f + dy holds tap dy of two dy-chains, so the fragment F(f, r) read at halo row r is the
        // operand of tile r (K-step 3f), tile r-1 (K-step 3f+1) and tile r-2 (K-step 3f+2).  The item is walked ROW-MAJOR: for each
        // halo row r = 0..9, four family fragments + one fragment of the ninth chain (K-steps 12 / 13) -- 50 fragment pairs (hi, lo) per
        // item instead of 8 tiles x 14 K-steps = 112: the LDS read traffic of the consumers was what held the matrix pipe at ~55 %
        // (917 KB of ds_read_b128 per item and CU next to 139 KB of producer writes, on a 256 B/clk array).  Same MFMAs (336), same
        // weights in registers.  Tile i is complete after row i + 2 and stored while the next rows compute (one store per row).
        static_assert(MT == TY && MT == 8 && TZ == 4, "one output plane per consumer wave");
        // devtools bit 32768: static priority for the matrix waves (65536: for the staging waves).  Measured round 3, same box: conv probe
        // -2..3 % at 32 / 128 channels and nothing at 16 with 32768, +10 % at the 16-channel level with 65536; the whole step does not move
        // (16.44-16.77 vs 16.43-16.75 ms) -- not enabled.
        if constexpr ((dbg & 32768) != 0) __builtin_amdgcn_s_setprio(3);
        const int mz = rw, my0 = 0;
        const int kg = lane >> 4;
        int fbase[6];                                   // packet offset of each fragment form at halo row 0 (hi plane; lo plane = + 2*HVOLP)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int t = sb_tap(3 * f, kg >> 1);       // dy = 0 tap of this lane's chain
            fbase[f] = (kg & 1) * HVOLP + ((mz + t / 9) * HY + my0) * HX + t % 3 + (lane & 15);
        }
        fbase[4] = (kg & 1) * HVOLP + ((mz + 2) * HY + my0 + (kg >> 1)) * HX + 2 + (lane & 15);   // ninth chain (dz 2, dx 2): slot 0 = row r, slot 1 = row r+1
        fbase[5] = (kg & 1) * HVOLP + ((mz + 2) * HY + my0) * HX + 2 + (lane & 15);               // rows 8, 9: both slots row r (slot 1 meets zero weights)
        static_assert(!HEAD || (!OUT16 && !MULTI && !BST && !ADD), "head form: one input chunk, NCDHW output, plain epilogue");
        constexpr int NKS = HEAD ? SB_HEAD_KSTEPS : SB_KSTEPS;
        u32x4 wreg[NKS][2];
        auto wptr = [&](int chunk) { return wfrag + ((size_t)(cog * nchunk + chunk) * (NKS * 2)) * 64 + lane; };
        auto load_w = [&](int chunk) {
            const u32x4* wp = wptr(chunk);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                wreg[ks][0] = wp[(ks * 2 + 0) * 64];
                if constexpr (NP == 3) wreg[ks][1] = wp[(ks * 2 + 1) * 64];
            }
        };
        load_w(0);                                      // one chunk: the weights stay in registers for the whole run of tiles
        f32x4 acc[HEAD ? 1 : MT];
        f32x4 hacc[HEAD ? NP : 1][HEAD ? MT + 2 : 1];   // head form: one accumulator per HALO row and product (consecutive MFMAs never share one)
        // OUT16: operands swapped -> D[m = cout][n = voxel] (lane owns 4 couts of one voxel, see sb_out_tile)
        auto mm = [](const bf16x8& av, const bf16x8& wv, const f32x4& c) -> f32x4 {
            if constexpr (OUT16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv, c, 0, 0, 0);
        };
        constexpr int NS = OUT16 ? 4 : 1;
        f32x4 s1, s2;                                   // whole vectors: the per-row update is 2 v_pk_add + 2 v_pk_fma, no packing moves
#pragma unroll
        for (int r = 0; r < NS; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        f32x4 kc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // BST: (k1, k2, thr) of this lane's 4 channels for sample kc_n
        int kc_n = -1;
        auto need_kc = [&](int n) __attribute__((always_inline)) {      // wave-uniform, reloads only when the sample changes
            if constexpr (BST) {
                if (n != kc_n) {
                    const float* kp = a.bst_k + (size_t)n * 3 * a.Cout + cog * 16 + 4 * (lane >> 4);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const float4 q = *reinterpret_cast<const float4*>(kp + (size_t)t * a.Cout);
                        kc[t] = f32x4{q.x, q.y, q.z, q.w};
                    }
                    kc_n = n;
                }
            }
        };
        // per-row operand of the epilogue (residual `add`, or BST: the forward tensor y): loaded three rows ahead of the store from a
        // clamped address; without such an operand every lane reads one dummy line and the value is dropped by a select (no branch
        // in the MFMA stream)
        constexpr bool has_r = BST || ADD;
        // GroupNorm statistics: ONE partial per (workgroup, consumer wave, sample) -- the tiles of a workgroup come in increasing
        // order, so a sample's tiles are consecutive; the partial is flushed when the sample changes and the samples this
        // workgroup never sees get zeros (the finalize kernel then reads gridDim.x*4 partials per channel instead of 4 per tile)
        const int stat_blk = blockIdx.x, stat_nblk = G;
        unsigned flushed = 0;                           // bit n: sample n has been written
        int n_acc = -1;                                 // sample whose statistics are being accumulated
        // flush: this wave's sums go to the LDS scratch (generation `par`); they are combined with the other consumer waves' and published
        // by wave 0 behind the next workgroup barrier (commit_stats at the start of the next item, or after the loop)
        float* stat_lds = smem + BUF * 8;                // behind the two image buffers (BUF packets of 16 bytes each)
        int pend_n = -1, pend_par = 0, par = 0;
        auto flush_stats = [&](int n) {
            if (a.stat_partials) sb_stats_to_lds<OUT16>(s1, s2, stat_lds + (par * 4 + rw) * 32, lane);
            pend_n = n; pend_par = par; par ^= 1;
            flushed |= 1u << (n & 31);
#pragma unroll
            for (int r = 0; r < NS; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        };
        auto commit_stats = [&]() {
            if (pend_n >= 0) {
                if (rw == 0 && a.stat_partials) sb_stats_commit(a, stat_lds + pend_par * 128, pend_n, cog, stat_blk, stat_nblk, lane);
                pend_n = -1;
            }
        };
        __syncthreads();                                // item 0 is staged
        const bool prof = (dbg & 64) != 0;
        unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
        // The tile of step k is t_begin + k*G: its (sample, tz, ty, tx) digits advance by the digits of G with carries -- scalar adds
        // instead of the five integer divisions per item (each ~40 instructions in this wave's stream between its MFMAs)
        int cn, ctz, cty, ctx, cstep = 0;
        auto digits_of_step = [&](int step) {
            int b = tile_of(step);
            cn = b / tiles_per_sample; b -= cn * tiles_per_sample;
            ctx = b % ntx; b /= ntx;
            cty = b % nty; ctz = b / nty;
        };
        digits_of_step(0);
        int gn, gz, gy, gx;
        {
            int b = G;
            gx = b % ntx; b /= ntx;
            gy = b % nty; b /= nty;
            gz = b % ntz; gn = b / ntz;
        }
        constexpr int NSTEP = 50;                       // (halo row r = 0..9) x (4 families + ninth chain)
        float dbg_sink = 0.f;
        int chunk = 0;
        for (int w = 0; w < nitems; ++w) {
            if (prof) t0 = __builtin_readcyclecounter();
            const bool last = chunk == nchunk - 1;      // this item completes its tile: the rows are stored as they finish
            const u32x4* wnext = wptr(chunk + 1 < nchunk ? chunk + 1 : 0);
            const u32x4* buf = lds + (w & 1) * BUF;
            const int n = cn;                            // sample of this item's tile
            SbOut so{};
            int ybase = 0;
            commit_stats();                              // (a flush of the previous item is complete in LDS since that item's barrier)
            if (last) {
                so = sb_out_prepare<OUT16>(a, cn, ctz * TZ + mz, ctx * 16, cog, lane);
                ybase = cty * TY + my0;
                if (n != n_acc) {                        // first tile of a new sample here: the previous sample's partial is complete
                    if (n_acc >= 0) flush_stats(n_acc);
                    n_acc = n;
                }
                need_kc(n);
            }
            if (MULTI && chunk == 0) {
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (prof) { t1 = __builtin_readcyclecounter(); pt[0] += t1 - t0; t0 = t1; }
            auto frag_ofs = [&](auto S) __attribute__((always_inline)) {
                constexpr int r = decltype(S)::value / 5, f = decltype(S)::value % 5;
                return fbase[f < 4 ? f : (r < 8 && !HEAD ? 4 : 5)] + r * HX;      // (head form: slot 1 of the ninth chain meets zero weights in every row)
            };
            constexpr int RING = (dbg & 16384) ? 4 : 3;  // (devtools bit 16384: one more step of look-ahead; +8 VGPRs)
            constexpr int AH = RING - 1;                 // steps of look-ahead
            bf16x8 fh[RING], fl[RING];                   // fragment ring: step s lives in slot s % RING, fetched AH steps ahead
            static_for<AH>([&](auto J) {
                constexpr int j = decltype(J)::value;
                fh[j] = __builtin_bit_cast(bf16x8, buf[frag_ofs(std::integral_constant<int, j>{})]);
                if constexpr (NP == 3) fl[j] = __builtin_bit_cast(bf16x8, buf[frag_ofs(std::integral_constant<int, j>{}) + 2 * HVOLP]);
            });
            float4 radd[3], rbst[3];                     // per-row operands (residual / BST forward tensor): tile i uses slot i % 3, loaded at row i + 1,
#pragma unroll                                           // consumed at row i + 3
            for (int j = 0; j < 3; ++j) { radd[j] = make_float4(0.f, 0.f, 0.f, 0.f); rbst[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
            auto store_tile = [&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                if constexpr (HEAD) {
                    // output row i = columns (co, dy 0) of halo row i + columns (co, 1) of halo row i+1 + columns (co, 2) of halo row i+2: four and eight
                    // lanes up in the same 16-lane row (same x positions) -- DPP row shifts, no LDS
                    f32x4 v[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        v[j] = hacc[0][i + j];
#pragma unroll
                        for (int pr = 1; pr < NP; ++pr) v[j] += hacc[pr][i + j];
                    }
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (v[0][e] + sb_row_shl<4>(v[1][e])) + sb_row_shl<8>(v[2][e]);
                    }
                    sb2_out_row<OUT16, has_r>(a, so, ybase + i, o, radd[i % 3], s1, s2);       // (lanes of columns >= Cout are masked by so.ok)
                } else if constexpr ((dbg & 8) != 0) {
                    dbg_sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];                  // ablation: the MFMAs stay, the row is dropped
                } else if constexpr (BST) {
                    sb_out_tile_bst<!MULTI>(a, so, ybase + i, acc[i], rbst[i % 3], kc, a.bst_slope, s1, s2, ADD ? &radd[i % 3] : nullptr);
                } else {
                    sb2_out_row<OUT16, has_r, !MULTI>(a, so, ybase + i, acc[i], radd[i % 3], s1, s2);
                }
            };
            // devtools bits 20-22 (value k = 1..3): the consumers drop the last k fragment families of every halo row -- their MFMAs AND their
            // LDS reads (k = 1: the ninth chain, 288 of 336 MFMAs left; 2: 216; 3: 144).  Results are wrong; the staging is unchanged.  It
            // answers "what does this skeleton do with 1.5x / 2.3x fewer matrix operations per staged image" before a reduced-product
            // kernel is written (profiles/r05_notes.txt).
            constexpr int NF_KEEP = 5 - ((dbg >> 20) & 7);
            static_for<NSTEP>([&](auto S) {
                constexpr int s = decltype(S)::value, r = s / 5, f = s % 5, cur = s % RING, nxt = (s + AH) % RING;
                const bf16x8 ah = fh[cur], al = fl[cur];
                bool fetched = (s + AH >= NSTEP) || ((s + AH) % 5 >= NF_KEEP);
                auto fetch = [&]() __attribute__((always_inline)) {            // the slot of step s-1 is free once its MFMAs are issued
                    if constexpr (s + AH < NSTEP && (s + AH) % 5 < NF_KEEP) {
                        const int o = frag_ofs(std::integral_constant<int, (s + AH < NSTEP ? s + AH : 0)>{});
                        fh[nxt] = __builtin_bit_cast(bf16x8, buf[o]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (NP == 3) {
                            fl[nxt] = __builtin_bit_cast(bf16x8, buf[o + 2 * HVOLP]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    fetched = true;
                };
                if constexpr (HEAD) {
                    // one MFMA per product: the columns (dy, co) of halo row r's accumulator take its contribution to output rows r, r-1, r-2
                    static_for<NP>([&](auto PR) {
                        constexpr int pr = decltype(PR)::value;
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[f][0]);
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[f][1]);
                        const f32x4 c = f == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : hacc[pr][r];
                        if constexpr (NP == 1) hacc[pr][r] = mm(ah, bh, c);
                        else if constexpr (pr == 0) hacc[pr][r] = mm(al, bh, c);
                        else if constexpr (pr == 1) hacc[pr][r] = mm(ah, bl, c);
                        else hacc[pr][r] = mm(ah, bh, c);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!fetched) fetch();
                    });
                } else if (!(dbg & 4) && f < NF_KEEP) {
                    if constexpr (f < 4) {
                        // tiles r-2 (dy 2), r-1 (dy 1), r (dy 0); products lo*hi, hi*lo, hi*hi -- product-major, so MFMAs on one accumulator
                        // are three apart
                        static_for<3>([&](auto PR) {
                            constexpr int pr = decltype(PR)::value;
                            static_for<3>([&](auto E) {
                                constexpr int dy = 2 - decltype(E)::value, i = r - dy;
                                if constexpr (i >= 0 && i < MT) {
                                    constexpr int ks = 3 * f + dy;
                                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[ks][0]);
                                    const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[ks][1]);
                                    constexpr bool first = !MULTI && f == 0 && dy == 0;      // first touch of tile i (one chunk): zero operand
                                    if constexpr (NP == 1) {
                                        if constexpr (pr == 2) acc[i] = first ? mm(ah, bh, f32x4{0.f, 0.f, 0.f, 0.f}) : mm(ah, bh, acc[i]);
                                    } else if constexpr (pr == 0) {
                                        acc[i] = first ? mm(al, bh, f32x4{0.f, 0.f, 0.f, 0.f}) : mm(al, bh, acc[i]);
                                    } else if constexpr (pr == 1) {
                                        acc[i] = mm(ah, bl, acc[i]);
                                    } else {
                                        acc[i] = mm(ah, bh, acc[i]);
                                    }
                                    if constexpr (NP == 3 || pr == 2) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        if (!fetched) fetch();
                                    }
                                }
                            });
                        });
                    } else {
                        // ninth chain: K-step 12 = (dy 0, dy 1) feeds tile r, K-step 13 = (dy 2, phantom) feeds tile r-2 from the same fragment
                        static_for<3>([&](auto PR) {
                            constexpr int pr = decltype(PR)::value;
                            static_for<2>([&](auto E) {
                                constexpr int ks = decltype(E)::value == 0 ? 13 : 12, i = decltype(E)::value == 0 ? r - 2 : r;
                                if constexpr (i >= 0 && i < MT) {
                                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[ks][0]);
                                    const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[ks][1]);
                                    if constexpr (NP == 1) { if constexpr (pr == 2) acc[i] = mm(ah, bh, acc[i]); }
                                    else if constexpr (pr == 0) acc[i] = mm(al, bh, acc[i]);
                                    else if constexpr (pr == 1) acc[i] = mm(ah, bl, acc[i]);
                                    else acc[i] = mm(ah, bh, acc[i]);
                                    if constexpr (NP == 3 || pr == 2) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        if (!fetched) fetch();
                                    }
                                }
                            });
                        });
                    }
                }
                if (!fetched) fetch();
                // ---- row bookkeeping between the MFMAs
                if constexpr (has_r && f == 0 && r >= 1 && r <= MT) {   // epilogue operands of tile r-1, two rows ahead of its store (unconditional, clamped address)
                    const int yy = ybase + r - 1;
                    const size_t ri = (last && so.ok && yy < H) ? sb_out_index<OUT16>(a, so, yy) : 0;
                    if constexpr (ADD) radd[(r - 1) % 3] = *reinterpret_cast<const float4*>(a.add + ri);
                    if constexpr (BST) rbst[(r - 1) % 3] = *reinterpret_cast<const float4*>(a.bst_y + ri);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (f == 1 && r >= 3) {           // tile r-3 was completed by row r-1: its MFMAs have drained by now
                    if (last) store_tile(std::integral_constant<int, r - 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MULTI && !(dbg & 128)) {      // (ablation bit 128: the weights are never refilled)
                    // a K-step's weights are dead for this item after their last tile: fetch the next chunk's into the same registers
                    // (dy 0 after row 7, dy 1 after row 8, dy 2 after row 9)
                    constexpr int ksd = f < 4 ? (r >= 7 ? 3 * f + (r - 7) : -1) : (r == 7 ? 12 : (r == 9 ? 13 : -1));
                    if constexpr (ksd >= 0) {
                        wreg[ksd][0] = wnext[(ksd * 2 + 0) * 64];
                        if constexpr (NP == 3) wreg[ksd][1] = wnext[(ksd * 2 + 1) * 64];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (s == 24) {
                    if (prof) { t1 = __builtin_readcyclecounter(); pt[1] += t1 - t0; t0 = t1; }
                }
            });
            if (last) store_tile(std::integral_constant<int, MT - 1>{});      // completed by the last row (tile 6 went out in row 9)
            if (++chunk == nchunk) {                    // next tile
                chunk = 0;
                ++cstep;
                if (zwalk) {
                    if (++ctz == ntz) digits_of_step(cstep);    // next panel (once per ntz steps)
                } else {
                    ctx += gx; if (ctx >= ntx) { ctx -= ntx; ++cty; }
                    cty += gy; if (cty >= nty) { cty -= nty; ++ctz; }
                    ctz += gz; if (ctz >= ntz) { ctz -= ntz; ++cn; }
                    cn += gn;
                }
            }
            if (prof) { t1 = __builtin_readcyclecounter(); pt[3] += t1 - t0; t0 = t1; }
            __syncthreads();
            if (prof) { t1 = __builtin_readcyclecounter(); pt[4] += t1 - t0; pt[5] += 1; }
        }
        if (prof) t0 = __builtin_readcyclecounter();
        if ((dbg & 8) && dbg_sink == 12345.678f) a.y[0] = dbg_sink;
        commit_stats();
        if (n_acc >= 0) flush_stats(n_acc);
        if (prof && rw == 0 && lane == 0) {
            pt[6] = __builtin_readcyclecounter() - t0;
#pragma unroll
            for (int i = 0; i < 7; ++i) atomicAdd(&sb2_prof[i], pt[i]);
            atomicAdd(&sb2_prof[7], 1ull);
        }
        __syncthreads();                                // (matched by the producers' closing barrier) the last flush is in LDS
        commit_stats();
        if (a.stat_partials && rw == 0 && !(dbg & 8)) {  // zeros for the samples this workgroup did not touch
            for (int n = 0; n < a.N; ++n)
                if (n >= 32 || !((flushed >> n) & 1u)) sb_stats_commit(a, nullptr, n, cog, stat_blk, stat_nblk, lane);
        }
    }
    fin_tail(a.fin, a.stat_partials, smem);              // RU_FUSE_TAIL_FINALIZE: the last workgroup of the launch finalizes the partials
}



template <int TZ, int TY, bool IN16, bool OUT16, bool MULTI, bool BST = false, bool ADD = false, int NP = 3, bool HEAD = false>
static int sb2_cfg_m(const Conv3Args& a, hipStream_t s) {
    using P = SB<TZ, TY>;
    static PerDevice attr_done;
    constexpr int LDS2 = 2 * P::LDS_BYTES + SB_STAT_LDS_FLOATS * 4;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_sb2_kernel<TZ, TY, IN16, OUT16, MULTI, BST, ADD, NP, HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_sb2)");
        attr_done.set();
    }
    static_assert(TZ == 4 && TY == 8, "sb2_grid_x assumes the (4,8,16) tile");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_sb2: at most 32 samples per call when statistics are requested");
    const int ntz = cdiv(a.D, TZ), nty = cdiv(a.H, TY), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)sb2_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)cdiv(a.Cout, 16));
    constexpr int dbg = 0;                           // (kernel argument kept for ABI stability of the launch; switches are compile-time)
    RU_REQUIRE(ADD == (a.add != nullptr), "conv3_sb2: residual operand and kernel variant disagree");
    RU_REQUIRE(!a.in_res || (HEAD && a.in_scale && !a.in_s16), "conv3_sb2: a residual of the INPUT exists in the head form only, together with the fused input transform");
    RU_REQUIRE(!a.in_sum_out || a.in_res, "conv3_sb2: the staged sum is written only where it is formed (in_res)");
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)LDS2),
               "conv3_sb2: tail descriptor does not match the launch");
    RU_REQUIRE(!IN16 || (size_t)a.D * a.H * a.W * 64 < ((size_t)1 << 31), "conv3_sb2: a 16-channel block of the voxel-major input must be smaller than 2 GiB (buffer addressing)");
    // head form: its fragments sit behind the direct ones of the same weight (conv3_sb_frag_bytes_direct; channel counts below 32 have no Winograd-z form)
    const u32x4* wfr = (const u32x4*)a.wfrag + (HEAD ? (size_t)cdiv(a.Cout, 16) * cdiv(a.Cin, 16) * SB_KSTEPS * 2 * 64 : 0);
    hipLaunchKernelGGL((conv3_sb2_kernel<TZ, TY, IN16, OUT16, MULTI, BST, ADD, NP, HEAD>), grid, dim3(512), LDS2, s, a, wfr, ntz, nty, ntx, cdiv(a.Cin, 16), dbg);
    RU_CHECK_LAUNCH("conv3_sb2_kernel");
    return RU_OK;
}
template <int TZ, int TY, bool IN16, bool OUT16, int NP = 3>
static int sb2_cfg(const Conv3Args& a, hipStream_t s) {
    if constexpr (IN16 && OUT16) {
        if (a.bst_y && a.add) return a.Cin > 16 ? sb2_cfg_m<TZ, TY, IN16, OUT16, true, true, true, NP>(a, s) : sb2_cfg_m<TZ, TY, IN16, OUT16, false, true, true, NP>(a, s);
        if (a.bst_y) return a.Cin > 16 ? sb2_cfg_m<TZ, TY, IN16, OUT16, true, true, false, NP>(a, s) : sb2_cfg_m<TZ, TY, IN16, OUT16, false, true, false, NP>(a, s);
    }
    if (a.add) return a.Cin > 16 ? sb2_cfg_m<TZ, TY, IN16, OUT16, true, false, true, NP>(a, s) : sb2_cfg_m<TZ, TY, IN16, OUT16, false, false, true, NP>(a, s);
    if constexpr (IN16 && !OUT16) {
        if (sb_head_shape(a.Cin, a.Cout) && conv3_sb_head_form_enabled()) return sb2_cfg_m<TZ, TY, IN16, OUT16, false, false, false, NP, true>(a, s);
    }
    return a.Cin > 16 ? sb2_cfg_m<TZ, TY, IN16, OUT16, true, false, false, NP>(a, s) : sb2_cfg_m<TZ, TY, IN16, OUT16, false, false, false, NP>(a, s);
}

// defined in conv3_sb2_c16.hip / conv3_sb2_c16_p1.hip (one product) / conv3_sb2_mixed.hip
int conv3_sb2_launch_c16(const Conv3Args& a, hipStream_t s);
int conv3_sb2_launch_c16_p1(const Conv3Args& a, hipStream_t s);
int conv3_sb2_launch_mixed(const Conv3Args& a, hipStream_t s);

}  // namespace ru
