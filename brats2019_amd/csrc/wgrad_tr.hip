// wgrad_tr.hip -- weight gradient of the 3x3x3 convolution for voxel-major (C16) tensors: split-bf16 operands on
// v_mfma_f32_16x16x32_bf16 (hi*hi + lo*hi + hi*lo, fp32 accumulate; numerics as in conv3_sb.hip / wgrad_sb.hip), fed by the
// LDS TRANSPOSE read of gfx950.
//
//   dw[o][c][tap] = sum_{n,voxel} dy[n][voxel][o] * act(x)[n][voxel + tap][c]
//
// GEMM per tap: M = 16 output channels o, N = 16 input channels c, K = voxels.  Both operands need, per lane, 8 K-values
// (voxels) of ONE channel, while memory and the LDS images are voxel-major ([position][16 channels] bf16, 32-byte rows).
// ds_read_b64_tr_b16 does that transposition: per 16-lane group, lane i supplies the address of 4 bf16 (row i>>2, columns
// 4*(i&3)..+3) and receives column i of the 4 x 16 block -- 4 consecutive voxels of channel i (semantics checked with
// tools/tr_probe.hip).  Two such reads fill one MFMA operand.  K-slot mapping of a K-block (2 tile rows x 16 x), the same
// for both operands: k-group g (= lane>>4), element e < 4 -> (row 0, x = 4g + e), e >= 4 -> (row 1, x = 4g + e - 4); so each
// read covers 4 consecutive positions (128 contiguous bytes) and the groups processed together are 128 bytes apart:
// conflict free, and a tap (dz,dy,dx) is nothing but a compile-time address offset into the halo image.
//
// Persistent 512-thread workgroups, as conv3_sb2: waves 0-3 CONSUME (each owns 7 of the 27 taps, compile-time, and keeps
// their accumulators in registers over every step it sees), waves 4-7 PRODUCE (two aligned float4 loads per voxel half ->
// fused GroupNorm-affine + LeakyReLU on x -> hi/lo split -> one ds_write_b128 per plane); the loads of item w+2 are in
// flight while item w+1 is converted and item w is on the matrix cores.  Partials per workgroup are combined in a fixed
// order by wgrad_reduce_kernel (wgrad_f32.hip).
#include "ru_common.h"

#include <utility>

namespace ru {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


template <int... Is, class F>
__device__ __forceinline__ void wt_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wt_static_for(F&& f) {
    wt_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ void wt_split8(const float (&t)[8], u32x4& hi, u32x4& lo) {
    split_n<4>(t, hi, lo);
}

__device__ __forceinline__ void wt_hi8(const float (&t)[8], u32x4& hi) {       // bf16_rne of 8 values, packed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ru_bf16x2 h;
        h[0] = (__bf16)t[2 * i];
        h[1] = (__bf16)t[2 * i + 1];
        hi[i] = __builtin_bit_cast(unsigned, h);
    }
}

__device__ __forceinline__ bf16x8 wt_read_tr(const char* p0, const char* p1) {
    // two transposed reads: rows 0-3 and 4-7 of this lane group's K-slots
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// ------------------------------------------------------------------ the kernel: z-marching columns
// A workgroup walks COLUMNS of the volume: for a fixed (sample, 8 rows, 16 x) it steps through z two planes at a time.  The
// x halo image is a ring of 8 plane slots in LDS (hi and lo): a step needs planes z0-1 .. z0+2, of which only the last two are
// new, so a step stages 2 x 180 halo positions + 256 dy positions for 256 voxels -- 2.4 positions per voxel instead of the
// 3.5 of an isolated (4,4,16) tile (the producers were the slower side: 288 us against 258 us of the consumers at L0).
// dy is double buffered as before.  Plane p of a column lives in slot (column_base + p) & 7, column_base advancing by D + 2 per
// column, so the 4 planes the consumers read and the up to 4 planes the producers write for the next item never collide.
template <int OT>
struct WTZ {
    static constexpr int TY = 8, HY = 10, HX = 18, NSLOT = 8;
    static constexpr int PPOS = HY * HX;                 // halo positions per plane
    static constexpr int PLANE = PPOS * 32;              // bytes of one plane image (hi or lo)
    static constexpr int XRING = NSLOT * PLANE;
    static constexpr int X_OFF = 0, XLO_OFF = XRING;
    static constexpr int DPOS = 2 * TY * 16, DPLANE = DPOS * 32;
    static constexpr int DBUF = OT * 2 * DPLANE;         // one dy buffer: OT blocks x (hi, lo)
    static constexpr int D_OFF = 2 * XRING;
    static constexpr int LDS = 2 * XRING + 2 * DBUF;
    static constexpr int NKB = 8;                        // K-blocks per step: 2 planes x 4 row pairs
};

// Consumer wave: ALL 27 taps for its share of the step's voxels (OT = 1: one plane, 4 of the 8 rows; OT = 2: one of the two
// o-tiles, one plane, all 8 rows) -- 27 accumulators, summed over the waves by wgrad_reduce_kernel (each wave writes its own
// partial).  K-blocks pair rows (y, y+2): for a 4-row half the two K-blocks (0,2) and (1,3) and the three dy taps touch only the
// four row pairs (p, p+2), p = 0..3 of the halo image, so per (dz, dx) 8 operands (hi and lo) feed 18 MFMAs; with the dy image
// read once per half that is 0.94 transposed reads per MFMA instead of 1.5, and no wave re-reads another wave's dy fragments.
// The reads of the next (dz, dx) group are issued between the MFMAs of the current one.
// NP = 1: plain bf16 operands (hi*hi only; gradient precision RU_PREC_BF16) -- the lo operands are neither read nor multiplied.
// PACK (x is a 4-channel copy, XS == 2): the staged x packet of halo position p holds the 4 channels of positions p, p+1, p+2, p+3, so the
// dx = 0 read of a unit delivers the operands of the taps dx = 0, 1, 2 at once in its 16 columns (j, c): 3 units (dz) per half and nine
// accumulators (dz, dy) instead of 9 units and 27 -- a third of the matrix work for an operand that has 4 real channels of 16.
template <int OT, int NP, bool PACK>
__device__ __forceinline__ void wtz_consume(const char* __restrict__ xl, const int (&pw)[3], const char* __restrict__ dl, f32x4 (&acc)[27]) {
    using P = WTZ<OT>;
    constexpr int HX = P::HX, NHALF = OT, NG = PACK ? 3 : 9, NU = NHALF * NG;       // units: (row half, (dz, dx))
    bf16x8 bh[2][4], bl[2][4];                                  // x operands of a unit: row pairs p = 0..3, double buffered
    bf16x8 ah[2][2], al[2][2];                                  // dy operands of a half: K-blocks (0,2), (1,3), double buffered by half
    // read r of unit u: r < 8 -> x operand (pair r>>1, hi/lo r&1); then, for the first unit of a half, the 4 dy operands
    // (one product: the odd reads -- the lo operands -- do not exist, read index rr counts the hi reads only)
    auto read_one = [&](auto U, auto R) {
        constexpr int u = decltype(U)::value, r = NP == 3 ? decltype(R)::value : 2 * decltype(R)::value;
        constexpr int h = u / NG, g = u % NG, dz = PACK ? g : g / 3, dx = PACK ? 0 : g % 3, set = u & 1;
        if constexpr (r < 8) {
            constexpr int pr = r >> 1;
            constexpr int off0 = ((4 * h + pr) * HX + dx) * 32, off1 = off0 + 2 * HX * 32;
            const char* p = xl + pw[dz];
            if constexpr ((r & 1) == 0) bh[set][pr] = wt_read_tr(p + P::X_OFF + off0, p + P::X_OFF + off1);
            else bl[set][pr] = wt_read_tr(p + P::XLO_OFF + off0, p + P::XLO_OFF + off1);
        } else {
            constexpr int ra = r - 8, kbi = ra >> 1;
            constexpr int off0 = (4 * h + kbi) * 16 * 32, off1 = off0 + 2 * 16 * 32;
            if constexpr ((ra & 1) == 0) ah[h & 1][kbi] = wt_read_tr(dl + off0, dl + off1);
            else al[h & 1][kbi] = wt_read_tr(dl + P::DPLANE + off0, dl + P::DPLANE + off1);
        }
    };
    auto nreads = [](int u) constexpr { return ((u % NG == 0) ? 12 : 8) / (NP == 3 ? 1 : 2); };
    wt_static_for<nreads(0)>([&](auto R) { read_one(std::integral_constant<int, 0>{}, R); });
    wt_static_for<NU>([&](auto U) {
        constexpr int u = decltype(U)::value, h = u / NG, g = u % NG, dz = PACK ? g : g / 3, dx = PACK ? 0 : g % 3, set = u & 1;
        constexpr int nr = u + 1 < NU ? nreads(u + 1) : 0;
        constexpr int NM = 6 * NP;
        wt_static_for<NM>([&](auto M) {
            constexpr int m = decltype(M)::value, prod = NP == 3 ? m / 6 : 2, kbi = (m / 3) % 2, dy = m % 3;
            constexpr int tap = PACK ? dz * 3 + dy : dz * 9 + dy * 3 + dx, pr = kbi + dy;    // PACK: accumulator (dz, dy), columns (dx, c)
            if constexpr (prod == 0) acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[h & 1][kbi], bh[set][pr], acc[tap], 0, 0, 0);
            if constexpr (prod == 1) acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[h & 1][kbi], bl[set][pr], acc[tap], 0, 0, 0);
            if constexpr (prod == 2) acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[h & 1][kbi], bh[set][pr], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int ra = m * nr / NM, rb = (m + 1) * nr / NM;
            wt_static_for<rb - ra>([&](auto K) {
                read_one(std::integral_constant<int, u + 1>{}, std::integral_constant<int, ra + decltype(K)::value>{});
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// XS: source of x -- 0 C16, 1 C4 copy (a 16-channel block that is zero beyond channel 3; kept for reference, not instantiated), 2 C4 copy
// with the three dx taps packed into the block's 16 columns (wtz_consume PACK; what the launcher uses for a 4-channel x).  DS: source of dy -- 0 C16, 1 split C16 (hi/lo
// packets in HBM, copied), 2 C4 copy.  Compile-time: a runtime branch inside the unrolled load loops breaks the load batches apart
// (it cost ~100 us per launch when these were kernel arguments).
template <int OT, int XS, int DS, int NP = 3>
__global__ __launch_bounds__(512, 2) void wgrad3_tz_kernel(const Wgrad3Args a, float* __restrict__ partials, int ntz, int nty, int ntx, int ncg, int CoP, int CiP) {
    using P = WTZ<OT>;
    constexpr int HX = P::HX, PPOS = P::PPOS, DPOS = P::DPOS, TY = P::TY;
    constexpr bool FA = DS == 3 || DS == 4, G16 = DS == 4;      // fused GroupNorm-backward apply; ... publishing the gradient-operand form of the MX scheme (compile time: a run-time
                                                                // branch in the unrolled store loop cost what the form's reader gains)
    extern __shared__ __attribute__((aligned(256))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3, ptid = tid & 255;
    const int og = blockIdx.y / ncg, cgp = blockIdx.y % ncg;
    const int D = a.D, H = a.H, W = a.W;
    const size_t DHW = (size_t)D * H * W;
    const int CBi = a.Cin >> 4, CBo = a.Cout >> 4;
    const int cols_per_sample = nty * ntx, ncol = a.N * cols_per_sample;
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;    // XCD-compact column order
    const int mycols = swz < ncol ? (ncol - swz + G - 1) / G : 0;
    const int nitems = mycols * ntz;
    const int colstride = 2 * ntz + 2;                  // planes a column occupies in the ring numbering
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;

    auto item_origin = [&](int item, int& n, int& k, int& y0, int& x0, int& ring0) {
        const int j = item / ntz;                        // j-th column of this workgroup
        k = item - j * ntz;
        int b = swz + j * G;
        n = b / cols_per_sample;
        b -= n * cols_per_sample;
        const int tx = b % ntx, ty = b / ntx;
        y0 = ty * TY; x0 = tx * 16;
        ring0 = j * colstride;                           // ring number of halo plane 0 (z = -1) of this column
    };

    if (producer) {
        if constexpr (G16) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);      // MODE.FP16_OVFL: the e4m3 conversions of the published gradient saturate
        // x units: (plane of the new set, halo position, channel half); dy units: (o-block, position, half)
        constexpr int NRX = (4 * PPOS + 127) / 128, NR2 = (2 * PPOS + 127) / 128, NRD = OT * ((DPOS + 127) / 128);     // first step of a column: 4 planes
        const int hsel = ptid & 1, pslot = ptid >> 1;
        float4 vx[NRX][2], vd[NRD][2];
        float4 vg[FA ? NRD : 1][2], gc4[FA ? 10 : 1];       // DS == 3: the gradient stream and (scale, shift, coefficients) of this thread's 8 channels
        int st_n = 0, st_y0 = 0, st_x0 = 0;
        float4 sc4[2], sh4[2];
        unsigned mx = 0, md = 0, mx2 = 0;                // (mx2: XS == 2, validity of the second 4-channel group of this half)
        // The x part and the dy part of an item are staged one after the other, and each part's loads for the NEXT item to be staged are issued as soon as
        // its registers have been converted -- x loads fly while the dy part converts, dy loads while the next x part does: a head start of a whole item
        // minus the part's own conversion, with ONE register set.  (All loads issued behind the whole conversion, as before, were needed right behind the
        // next barrier: the staging waves -- the pole of the 16-channel kernels, ~160 MFMAs per item beside them -- sat out a memory round trip per item;
        // profiles/r05_notes.txt, sections 6 and 10.)  State of the item whose x / dy loads are in the registers:
        int stx_ring0 = 0, stx_k = 0, std_k = 0;
        int nx_n = 0, nx_k = 0, nx_y0 = 0, nx_x0 = 0;    // the item issue_x was last called for: issue_d stages the same one
        // issue() is called for items 0, 1, 2, ... in order: (column, step) advance as counters and the column origin is recomputed
        // once per column, not with four integer divisions per item in this wave's VALU stream
        int is_k = 0, is_n = 0, is_y0 = 0, is_x0 = 0, is_ring0 = 0;
        auto load_gc = [&](int n, int q) __attribute__((always_inline)) {          // (scale, shift, 3 coefficients) of this thread's 8 channels of output block q
            if constexpr (FA) {
                const size_t go = (size_t)n * a.Cout + (og * OT + q) * 16 + hsel * 8;
                gc4[0] = *reinterpret_cast<const float4*>(a.gb_scale + go); gc4[1] = *reinterpret_cast<const float4*>(a.gb_scale + go + 4);
                gc4[2] = *reinterpret_cast<const float4*>(a.gb_shift + go); gc4[3] = *reinterpret_cast<const float4*>(a.gb_shift + go + 4);
#pragma unroll
                for (int t = 0; t < 6; ++t) gc4[4 + t] = *reinterpret_cast<const float4*>(a.gb_coef + go * 3 + 4 * t);
            }
        };
        // (item >= nitems: the loads of the last item are issued once more and never staged -- no branch around a batch of loads, so the compiler's
        // s_waitcnt vmcnt counts stay exact: with `if (w + 2 < nitems) issue(...)` the wait for the dy registers had to assume NO younger loads, i.e. vmcnt(0),
        // and sat out the x loads issued a moment earlier)
        auto issue_x = [&](int item) {
            if (item < nitems) {
                if (is_k == 0) item_origin(item, is_n, is_k, is_y0, is_x0, is_ring0);
                nx_n = is_n; nx_k = is_k; nx_y0 = is_y0; nx_x0 = is_x0;
                stx_ring0 = is_ring0; stx_k = is_k;
                if (++is_k == ntz) is_k = 0;
            }
            const int n = nx_n, k = nx_k, y0 = nx_y0, x0 = nx_x0;
            const int hp0 = k == 0 ? 0 : 2 * k + 2, npl = k == 0 ? 4 : 2;       // new halo planes hp0 .. hp0 + npl - 1 (halo plane hp <-> z = hp - 1)
            const float* xb = a.x + ((size_t)(n * CBi + cgp) * DHW) * 16 + hsel * 8;
            mx = 0; mx2 = 0;
            // rounds 0 .. NR2-1 cover the two planes every step loads; rounds NR2 .. NRX-1 only exist at the start of a column (four
            // planes): ONE wave-uniform branch around them, none inside the unrolled loops (a branch per round splits the load batch)
            auto xround = [&](auto R) {
                constexpr int r = decltype(R)::value;
                const int u = r * 128 + pslot;
                const int pl = u / PPOS, p = u - pl * PPOS;
                const int hy = p / HX, xc = p - hy * HX;
                const int gz = hp0 + pl - 1, gy = y0 + hy - 1, gx = x0 + xc - 1;
                const bool live = pl < npl;
                const bool ok = live && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#if defined(RU_SB2_DBG) && (RU_SB2_DBG & (1 << 24))
                const size_t ofs = 0;                    // devtools bit 24: every staging load hits one cache line (results wrong): is the staging LATENCY on the critical path?
#else
                const size_t ofs = ok ? (size_t)((gz * H + gy) * W + gx) * 16 : 0;
#endif
                if constexpr (XS == 1) {                 // 4-channel copy: channels 0-3 real (half 0), the rest of the block is zero
                    mx |= (ok && hsel == 0) ? (1u << r) : 0u;
                    vx[r][0] = *reinterpret_cast<const float4*>(a.x + (size_t)n * DHW * 4 + (ofs >> 2));
                    vx[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if constexpr (XS == 2) {          // 4-channel copy, taps packed: channels (j, c) = x4[gx + j][c]; this half holds j = 2*hsel, 2*hsel + 1
                    const bool rowok = live && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H;
                    const int g0 = gx + 2 * hsel, g1 = g0 + 1;
                    const bool ok0 = rowok && (unsigned)g0 < (unsigned)W, ok1 = rowok && (unsigned)g1 < (unsigned)W;
                    mx |= ok0 ? (1u << r) : 0u;
                    mx2 |= ok1 ? (1u << r) : 0u;
                    const float* x4 = a.x + (size_t)n * DHW * 4;
                    const size_t row = rowok ? (size_t)(gz * H + gy) * W : 0;
                    vx[r][0] = *reinterpret_cast<const float4*>(x4 + (row + (ok0 ? g0 : 0)) * 4);      // unconditional, clamped
                    vx[r][1] = *reinterpret_cast<const float4*>(x4 + (row + (ok1 ? g1 : 0)) * 4);
                } else {
                    mx |= ok ? (1u << r) : 0u;
                    vx[r][0] = *reinterpret_cast<const float4*>(xb + ofs);          // unconditional, clamped
                    vx[r][1] = *reinterpret_cast<const float4*>(xb + ofs + 4);
                }
            };
            wt_static_for<NR2>(xround);
            if (k == 0) wt_static_for<NRX - NR2>([&](auto R) { xround(std::integral_constant<int, NR2 + decltype(R)::value>{}); });
            if (xform) {
                const int cofs = n * a.Cin + cgp * 16 + hsel * 8;
                sc4[0] = *reinterpret_cast<const float4*>(a.in_scale + cofs); sc4[1] = *reinterpret_cast<const float4*>(a.in_scale + cofs + 4);
                sh4[0] = *reinterpret_cast<const float4*>(a.in_shift + cofs); sh4[1] = *reinterpret_cast<const float4*>(a.in_shift + cofs + 4);
            }
        };
        auto issue_d = [&]() {
            const int n = nx_n, k = nx_k, y0 = nx_y0, x0 = nx_x0;
            std_k = k;
            md = 0;
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                constexpr int RPB = (DPOS + 127) / 128;
                const int q = r / RPB, p = (r - q * RPB) * 128 + pslot;
                const int row = p >> 4, xc = p & 15;
                const int z = row / TY, y = row - z * TY;
                const int gz = 2 * k + z, gy = y0 + y, gx = x0 + xc;
                const bool ok = p < DPOS && gz < D && gy < H && gx < W;
                const float* db = a.dy + ((size_t)(n * CBo + og * OT + q) * DHW) * 16 + hsel * 8;
#if defined(RU_SB2_DBG) && (RU_SB2_DBG & (1 << 24))
                const size_t ofs = 0;
#else
                const size_t ofs = ok ? (size_t)((gz * H + gy) * W + gx) * 16 : 0;
#endif
                if constexpr (FA) {                 // GroupNorm-backward apply on the fly: forward tensor y and gradient d
                    md |= ok ? (1u << r) : 0u;
                    const size_t cb = ((size_t)(n * CBo + og * OT + q) * DHW) * 16 + ofs + hsel * 8;
                    vd[r][0] = *reinterpret_cast<const float4*>(a.gb_y + cb);
                    vd[r][1] = *reinterpret_cast<const float4*>(a.gb_y + cb + 4);
                    vg[r][0] = *reinterpret_cast<const float4*>(a.gb_d + cb);
                    vg[r][1] = *reinterpret_cast<const float4*>(a.gb_d + cb + 4);
                } else if constexpr (DS == 1) {                 // split form in HBM: hi and lo packets of this half, copied as they are
                    md |= ok ? (1u << r) : 0u;
                    const float* ds = a.dy + ((size_t)(n * CBo + og * OT + q) * DHW) * 16 + ofs + hsel * 4;
                    vd[r][0] = *reinterpret_cast<const float4*>(ds);
                    if constexpr (NP == 3) vd[r][1] = *reinterpret_cast<const float4*>(ds + 8);      // (one product: the lo packet is not read)
                } else if constexpr (DS == 2) {
                    md |= (ok && hsel == 0) ? (1u << r) : 0u;
                    vd[r][0] = *reinterpret_cast<const float4*>(a.dy + (size_t)n * DHW * 4 + (ofs >> 2));
                    vd[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    md |= ok ? (1u << r) : 0u;
                    vd[r][0] = *reinterpret_cast<const float4*>(db + ofs);
                    vd[r][1] = *reinterpret_cast<const float4*>(db + ofs + 4);
                }
            }
            if constexpr (FA) {
                if constexpr (OT == 1) load_gc(n, 0);    // one output block: its constants travel with the loads (two blocks: fetched per block
                st_n = n; st_y0 = y0; st_x0 = x0;        // in store(), 40 registers live instead of 80 held across the whole item)
            }
        };
        auto store_x = [&]() {
            float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (xform) {
                sc[0] = sc4[0].x; sc[1] = sc4[0].y; sc[2] = sc4[0].z; sc[3] = sc4[0].w; sc[4] = sc4[1].x; sc[5] = sc4[1].y; sc[6] = sc4[1].z; sc[7] = sc4[1].w;
                sh[0] = sh4[0].x; sh[1] = sh4[0].y; sh[2] = sh4[0].z; sh[3] = sh4[0].w; sh[4] = sh4[1].x; sh[5] = sh4[1].y; sh[6] = sh4[1].z; sh[7] = sh4[1].w;
            }
            const int hp0 = stx_k == 0 ? 0 : 2 * stx_k + 2, npl = stx_k == 0 ? 4 : 2;
            auto xbody = [&](auto XFORM) {                // one wave-uniform dispatch, then a branch-free unrolled loop
                constexpr bool XF = decltype(XFORM)::value;
            auto sround = [&](auto R) {
                constexpr int r = decltype(R)::value;
                const int u = r * 128 + pslot;
                const int pl = u / PPOS, p = u - pl * PPOS;
                if (pl >= npl) return;
                const bool ok = (mx >> r) & 1u;
                const bool ok_hi = XS == 2 ? ((mx2 >> r) & 1u) != 0 : ok;          // packed taps: the second 4-channel group has its own x position
                const float f[8] = {vx[r][0].x, vx[r][0].y, vx[r][0].z, vx[r][0].w, vx[r][1].x, vx[r][1].y, vx[r][1].z, vx[r][1].w};
                float t[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool okc = c < 4 ? ok : ok_hi;
                    if constexpr (XF) {
                        const float u2 = fmaf(f[c], sc[c], sh[c]);
                        t[c] = okc ? fmaxf(u2, u2 * slope) : 0.f;   // zero padding applies to the ACTIVATED tensor
                    } else {
                        t[c] = okc ? f[c] : 0.f;
                    }
                }
                u32x4 hi, lo;
                if constexpr (NP == 3) wt_split8(t, hi, lo);
                else wt_hi8(t, hi);                      // one product: only the hi image exists
                const int slot = (stx_ring0 + hp0 + pl) & (P::NSLOT - 1);
                char* dst = lds + slot * P::PLANE + p * 32 + hsel * 16;
                *reinterpret_cast<u32x4*>(dst + P::X_OFF) = hi;
                if constexpr (NP == 3) *reinterpret_cast<u32x4*>(dst + P::XLO_OFF) = lo;
            };
            wt_static_for<NR2>(sround);
            if (stx_k == 0) wt_static_for<NRX - NR2>([&](auto R) { sround(std::integral_constant<int, NR2 + decltype(R)::value>{}); });
            };
            if (xform) xbody(std::true_type{});
            else xbody(std::false_type{});
        };
        auto store_d = [&](char* dbuf) {
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                constexpr int RPB = (DPOS + 127) / 128;
                const int q = r / RPB, p = (r - q * RPB) * 128 + pslot;
                if constexpr (FA && OT > 1) {
                    if (r % RPB == 0) load_gc(st_n, q);  // (r is a compile-time constant after unrolling)
                }
                const bool ok = (md >> r) & 1u;
                const float f[8] = {vd[r][0].x, vd[r][0].y, vd[r][0].z, vd[r][0].w, vd[r][1].x, vd[r][1].y, vd[r][1].z, vd[r][1].w};
                float t[8];
                if constexpr (FA) {                 // the expression of gn_bwd_apply16_split_kernel, term for term
                    const float g[8] = {vg[r][0].x, vg[r][0].y, vg[r][0].z, vg[r][0].w, vg[r][1].x, vg[r][1].y, vg[r][1].z, vg[r][1].w};
                    const float ga[8] = {gc4[0].x, gc4[0].y, gc4[0].z, gc4[0].w, gc4[1].x, gc4[1].y, gc4[1].z, gc4[1].w};
                    const float gb[8] = {gc4[2].x, gc4[2].y, gc4[2].z, gc4[2].w, gc4[3].x, gc4[3].y, gc4[3].z, gc4[3].w};
                    const float cf[24] = {gc4[4].x, gc4[4].y, gc4[4].z, gc4[4].w, gc4[5].x, gc4[5].y, gc4[5].z, gc4[5].w, gc4[6].x, gc4[6].y, gc4[6].z, gc4[6].w,
                                          gc4[7].x, gc4[7].y, gc4[7].z, gc4[7].w, gc4[8].x, gc4[8].y, gc4[8].z, gc4[8].w, gc4[9].x, gc4[9].y, gc4[9].z, gc4[9].w};
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float dyv = cf[3 * c] * ((f[c] * ga[c] + gb[c]) > 0.f ? g[c] : g[c] * a.gb_slope) + (cf[3 * c + 1] * f[c] + cf[3 * c + 2]);
                        t[c] = ok ? dyv : 0.f;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) t[c] = ok ? f[c] : 0.f;
                }
                u32x4 hi, lo = u32x4{0u, 0u, 0u, 0u};
                float lof[FA ? 8 : 1], amax = 0.f;  // DS == 3: the exact residuals and the largest |hi| of this half (the gradient-operand form of the published tensor)
                if constexpr (DS == 1) {
                    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
                    hi = ok ? __builtin_bit_cast(u32x4, vd[r][0]) : z;
                    if constexpr (NP == 3) lo = ok ? __builtin_bit_cast(u32x4, vd[r][1]) : z;
                } else if constexpr (FA) {          // wt_split8's hi / lo (fused apply: the published gradient keeps its lo half whatever this kernel multiplies)
                    amax = mxg_hi8(t, hi, lof);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ru_bf16x2 l2;
                        l2[0] = (__bf16)lof[2 * c]; l2[1] = (__bf16)lof[2 * c + 1];
                        lo[c] = __builtin_bit_cast(unsigned, l2);
                    }
                } else if constexpr (NP == 3) {
                    wt_split8(t, hi, lo);
                } else {
                    wt_hi8(t, hi);
                }
                *reinterpret_cast<u32x4*>(dbuf + q * 2 * P::DPLANE + p * 32 + hsel * 16) = hi;
                if constexpr (NP == 3) *reinterpret_cast<u32x4*>(dbuf + q * 2 * P::DPLANE + P::DPLANE + p * 32 + hsel * 16) = lo;
                if constexpr (FA) {                 // publish dy in split form (every position is staged once by input-channel group 0)
                    if (ok && cgp == 0 && a.gb_out) {    // (no output tensor: nobody but this weight gradient consumes the gradient, e.g. the stem)
                        const int row = p >> 4, z = row / TY;
                        const size_t vox = (size_t)((2 * std_k + z) * H + st_y0 + (row - z * TY)) * W + st_x0 + (p & 15);
                        u32x4* op = reinterpret_cast<u32x4*>(a.gb_out) + ((size_t)(st_n * CBo + og * OT + q) * DHW + vox) * 4;
                        op[hsel] = hi;
                        if constexpr (G16) {             // [hi ch 0-7 | hi ch 8-15 | e4m3 (lo, g) ch 0-7 | e4m3 (lo, g) ch 8-15], e from the voxel's largest |hi| (partner lane: the other half)
                            const float m = fmaxf(amax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, amax), 0xb1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
                            unsigned l8[2], x8[2];
                            mxg_cvt8(t, lof, m, l8, x8);
                            op[2 + hsel] = u32x4{l8[0], l8[1], x8[0], x8[1]};      // this half's e4m3(lo) and e4m3(value): ONE 16-byte store, as in the split form
                        } else {
                            op[2 + hsel] = lo;
                        }
                    }
                }
            }
        };
        if (nitems > 0) {
            issue_x(0);
            issue_d();
            store_x();
            issue_x(1);
            store_d(lds + P::D_OFF);
            issue_d();
        }
        __syncthreads();
        for (int w = 0; w < nitems; ++w) {
            if (w + 1 < nitems) {
                store_x();
                issue_x(w + 2);
                store_d(lds + P::D_OFF + ((w + 1) & 1) * P::DBUF);
                issue_d();
            }
            __syncthreads();
        }
        __syncthreads();                                // the consumers' accumulators are in LDS
    } else {
        f32x4 acc[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int i16 = lane & 15, g = lane >> 4;
        const int lane_off = (4 * g + (i16 >> 2)) * 32 + (i16 & 3) * 8;
        // this wave's share of a step: plane zw; OT = 1: rows 4*(rw&1) .. +3 of it, OT = 2: all 8 rows and o-tile rw & 1
        const int zw = rw >> 1;
        const int yb = OT == 1 ? 4 * (rw & 1) : 0, qw = OT == 1 ? 0 : (rw & 1);
        const int xrow_off = yb * HX * 32 + lane_off;
        const int drow_off = (zw * TY + yb) * 16 * 32 + qw * 2 * P::DPLANE + lane_off;
        __syncthreads();                                // item 0 is staged
        int cj = 0, ck = 0;
        for (int w = 0; w < nitems; ++w) {
            const int j = cj, k = ck;
            if (++ck == ntz) { ck = 0; ++cj; }
            const int s0 = j * colstride + 2 * k + zw;   // ring number of halo plane (z0 - 1) + zw: tap dz reads plane s0 + dz
            int pw[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) pw[p] = ((s0 + p) & (P::NSLOT - 1)) * P::PLANE + xrow_off;
            const char* dl = lds + P::D_OFF + (w & 1) * P::DBUF + drow_off;
            wtz_consume<OT, NP, XS == 2>(lds, pw, dl, acc);
            __syncthreads();
        }
        // ---- the waves that worked on the same output tile are summed through LDS (the staging memory is free now): ONE partial per
        // workgroup instead of four / two -- a quarter / half of the partial traffic here and in the reduce kernel
        f32x4* red = reinterpret_cast<f32x4*>(lds);      // [wave][tap][lane]
        constexpr int NT = XS == 2 ? 9 : 27;             // live accumulators (packed taps: (dz, dy))
#pragma unroll
        for (int t = 0; t < NT; ++t) red[(rw * 27 + t) * 64 + lane] = acc[t];
        __syncthreads();
    }
    // partials[workgroup][tap][o][c]; D lane = (rows o = 4*(l>>4) + r, column c = l&15); fixed summation order
    {
        const f32x4* red = reinterpret_cast<const f32x4*>(lds);
        const int c0 = cgp * 16;
        constexpr int NT = XS == 2 ? 9 : 27;
        for (int e = tid; e < OT * NT * 64; e += 512) {
            const int q = e & 63, t = (e >> 6) % NT, ot = (e >> 6) / NT;
            f32x4 v;
            if constexpr (OT == 1) v = (red[(0 * 27 + t) * 64 + q] + red[(1 * 27 + t) * 64 + q]) + (red[(2 * 27 + t) * 64 + q] + red[(3 * 27 + t) * 64 + q]);
            else v = red[(ot * 27 + t) * 64 + q] + red[((ot + 2) * 27 + t) * 64 + q];
            int c = c0 + (q & 15), tap = t;
            bool live = true;
            if constexpr (XS == 2) {                     // column (j, c) of accumulator (dz, dy) is tap (dz, dy, dx = j), input channel c
                const int j = (q & 15) >> 2;
                c = q & 3;
                tap = (t / 3) * 9 + (t % 3) * 3 + j;
                live = j < 3;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = og * OT * 16 + ot * 16 + (q >> 4) * 4 + r;
                if (live && o < CoP && c < CiP) partials[(((size_t)blockIdx.x * 27 + tap) * CoP + o) * CiP + c] = v[r];
            }
        }
    }
}

struct WTRChoice { int ot, nbx, ngroups, ncg; };
static WTRChoice wtr_choose(int N, int Cin, int Cout, int D, int H, int W) {
    WTRChoice c;
    c.ot = Cout >= 32 ? 2 : 1;
    c.ncg = Cin / 16;
    c.ngroups = (Cout / 16 / c.ot) * c.ncg;
    const long ntile = (long)N * cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 16);
    long nbx = 256 / c.ngroups;                          // one resident workgroup per CU in total
    if (nbx < 1) nbx = 1;
    if (nbx > ntile) nbx = ntile;
    c.nbx = (int)nbx;
    return c;
}

size_t wgrad3_tr_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    if (Cin % 16 || Cout % 16) return 0;
    const WTRChoice c = wtr_choose(N, Cin, Cout, D, H, W);
    return (size_t)c.nbx * 27 * Cout * Cin * sizeof(float);       // one partial per workgroup
}

template <int OT, int XS, int DS, int NP = 3>
static int wtz_cfg(const Wgrad3Args& a, const WTRChoice& c, hipStream_t s) {
    using P = WTZ<OT>;
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_tz_kernel<OT, XS, DS, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad3_tz)");
        attr_done.set();
    }
    const int ntz = cdiv(a.D, 2), nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    const long ncol = (long)a.N * nty * ntx;
    int nbx = c.nbx;
    if (nbx > ncol) nbx = (int)ncol;
    hipLaunchKernelGGL((wgrad3_tz_kernel<OT, XS, DS, NP>), dim3(nbx, c.ngroups), dim3(512), P::LDS, s, a, (float*)a.ws, ntz, nty, ntx, c.ncg, a.Cout, a.Cin);
    RU_CHECK_LAUNCH("wgrad3_tz_kernel");
    const int co = a.dw_cout > 0 ? a.dw_cout : a.Cout, ci = a.dw_cin > 0 ? a.dw_cin : a.Cin;
    if (a.swapped) return wgrad_reduce_launch((const float*)a.ws, nbx, 27, a.Cout, a.Cin, co, ci, a.dw, 27, co * 27, 0, s, 1, a.defer);   // dw[cout = c'][cin = o'][26 - t]
    return wgrad_reduce_launch((const float*)a.ws, nbx, 27, a.Cout, a.Cin, co, ci, a.dw, ci * 27, 27, 0, s, 0, a.defer);
}

int wgrad3_tr_launch(const Wgrad3Args& a, hipStream_t s) {
    RU_REQUIRE(a.x_c16 && a.dy_c16 && a.Cin % 16 == 0 && a.Cout % 16 == 0, "wgrad3_tr: needs voxel-major x and dy with channel counts %% 16 == 0");
    const WTRChoice c = wtr_choose(a.N, a.Cin, a.Cout, a.D, a.H, a.W);
    if (!a.ws || a.ws_bytes < wgrad3_tr_workspace_bytes(a.N, a.Cin, a.Cout, a.D, a.H, a.W)) {
        set_error("wgrad3_tr: workspace too small");
        return RU_ENOMEM;
    }
    const bool p1 = a.products == 1;                     // one-product forms exist for the variants the engine's backward runs
    if (a.gb_y) {
        RU_REQUIRE(!a.dy_c4 && a.gb_d && a.gb_scale && a.gb_shift && a.gb_coef && (a.gb_out || a.x_c4),
                   "wgrad3_tr: the fused GroupNorm-backward apply needs all of its operands");
        if (a.x_c4) {                                    // stem: x = network input
            RU_REQUIRE(c.ot == 1, "wgrad3_tr: a 4-channel copy stands for ONE 16-channel block");
            return p1 ? wtz_cfg<1, 2, 3, 1>(a, c, s) : wtz_cfg<1, 2, 3>(a, c, s);      // (taps packed into the 16 columns: XS == 2)
        }
        if (p1 && c.ot == 1) return wtz_cfg<1, 0, 3, 1>(a, c, s);
        if (c.ot == 2) return wtz_cfg<2, 0, 3>(a, c, s);   // two output blocks per workgroup: the constants of a block are fetched when it is converted
        return a.gb_g16 ? wtz_cfg<1, 0, 4>(a, c, s) : wtz_cfg<1, 0, 3>(a, c, s);
    }
    const int xs = a.x_c4 ? 1 : 0, ds = a.dy_c4 ? 2 : (a.dy_s16 ? 1 : 0);
    if (c.ot == 2) {
        RU_REQUIRE(xs == 0 && ds != 2, "wgrad3_tr: 4-channel copies stand for ONE 16-channel block");
        if (p1 && ds == 1) return wtz_cfg<2, 0, 1, 1>(a, c, s);
        return ds == 1 ? wtz_cfg<2, 0, 1>(a, c, s) : wtz_cfg<2, 0, 0>(a, c, s);
    }
    if (xs == 1) {
        RU_REQUIRE(ds != 2, "wgrad3_tr: only one operand can be a 4-channel copy");
        if (ds == 1) return wtz_cfg<1, 2, 1>(a, c, s);
        return p1 ? wtz_cfg<1, 2, 0, 1>(a, c, s) : wtz_cfg<1, 2, 0>(a, c, s);
    }
    if (ds == 2) return p1 ? wtz_cfg<1, 0, 2, 1>(a, c, s) : wtz_cfg<1, 0, 2>(a, c, s);
    return ds == 1 ? wtz_cfg<1, 0, 1>(a, c, s) : wtz_cfg<1, 0, 0>(a, c, s);
}

}  // namespace ru
