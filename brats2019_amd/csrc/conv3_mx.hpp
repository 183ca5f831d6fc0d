// conv3_mx.hpp -- the 3x3x3 FORWARD convolution of the 16-channel level (model.py:72-73 as used by model.py:89-91 / 340-345 / 379-395: Cin = 16, voxel-major in
// and out) with a cheaper product scheme on gfx950's new matrix formats (round 6).
//
//   split-bf16 (conv3_sb2_kernel):  x*w ~= hi*hi + lo*hi + hi*lo, three v_mfma_f32_16x16x32_bf16 per K-step: 3 x 14 = 42 MFMA units per 27-tap x 16-channel chain
//   here:                           x*w ~= f16(x)*f16(w)                                   14 x v_mfma_f32_16x16x32_f16          (14 units)
//                                         + 2^-19 * [ e4m3(x_lo * 2^11) * e4m3(w * 2^8)    } both cross terms in ONE K = 128 chain of
//                                                   + e4m3(x) * e4m3(w_lo * 2^19) ]        } v_mfma_scale_f32_16x16x128_f8f6f4  (7 x 2 = 14 units)
//   with x_lo = x - f16(x), w_lo = w - f16(w) (exact in fp32).  An fp16 significand has 11 bits, so the residuals are 2^-12 of the value and a 4-bit e4m3
//   significand on the cross terms leaves 2^-16 -- the error class of the three-product scheme (tools/mx_gate.py: max |dp| 1.5e-4 against 6.2e-5 on the whole
//   1 x 128^3 network with EVERY 3x3x3 convolution replaced; profiles/r06_mx_gate.txt).  The matrix pipe at the socket power limit takes 1.37-1.40x less time for
//   a chain (tools/mx_probe.hip, profiles/r06_mx_probe.txt); per item a consumer wave issues 112 + 56 = 168 matrix instructions (224 units) instead of 336.
//
//   Scales: SX + SWH == SWL, so ONE pair of E8M0 constants (activations 2^-11, weights 2^-8) serves both cross terms -- the hardware takes the scale of a
//   32-element K block from one lane group's register, and a block mixes both cross terms here (below), so per-term scales are not available; nor are
//   data-dependent ones (a staged LDS element meets different taps in different output voxels).  Range: |x| <= 448 and |w| <= 1.75 convert without saturation;
//   beyond, MODE.FP16_OVFL makes the conversions SATURATE (tools/mx_ovfl_probe.hip: without it e4m3 overflow is NaN) and the cross terms lose accuracy gradually.
//
//   K layout of the scaled MFMA (tools/mx_layout_probe.hip, mx_scale_probe*.hip): lane (row = l & 15, k-group g = l >> 4) holds 32 bytes; both operands are filled by
//   the same (lane, byte) rule, so only the pairing matters: byte b of lane group g of A meets byte b of lane group g of B.  Here bytes 0-15 = tap slot 0, bytes
//   16-31 = tap slot 1 (16 channels of one position each); g & 1 selects the cross term (0: x_lo plane x w_hi fragments, 1: x plane x w_lo fragments), g >> 1 the
//   tap pair.  The two lane groups that share an LDS cycle of a ds_read_b128 (g = 2j, 2j + 1) then read the SAME positions of two planes a multiple of 256 bytes
//   apart: conflict free for every tap.
//
//   Cross fragments: the 27 taps are 9 chains (dz, dx) of three dy taps (sb_tap).  X(h, r), h = 0 / 1: chains 4h .. 4h+3 at halo row r -- operand of output rows
//   r, r-1, r-2 with the weights of dy = 0, 1, 2 (the row-major walk of conv3_sb2_kernel); N(i): the ninth chain's three taps for output row i (rows i, i+1, i+2
//   in slots (j, slot) = (0,0), (0,1), (1,0); slot (1,1) meets zero weights).  Per item: 20 X fragments x (up to) 3 MFMAs + 8 N = 56 scaled MFMAs.
//
//   Same persistent producer / consumer skeleton, LDS budget and tile walk as conv3_sb2_kernel<4, 8, true, true, false, false, false> (the kernel bench.py's
//   `roofline` names), whose epilogue helpers it shares.  LDS image per buffer: planes 0 / 1 = fp16 halves (channels 0-7 / 8-15, 16-byte packets as before),
//   plane 2 = e4m3(x_lo * 2^11), plane 3 = e4m3(x): 16 channels of a position per 16-byte packet.
#pragma once
#include "conv3_mx_pack.hpp"

namespace ru {

// cross step xi of an item (compile-time): rows 0 and 1 have X(0, r), X(1, r); rows 2..9 have X(0, r), X(1, r), N(r - 2)
struct MxCross { int r, q; };                       // q: 0 / 1 = X(q, r), 2 = N(r - 2)
__host__ __device__ constexpr MxCross mx_cross_of(int xi) { return xi < 4 ? MxCross{xi / 2, xi % 2} : MxCross{2 + (xi - 4) / 3, (xi - 4) % 3}; }
constexpr int MX_NCROSS = 4 + 8 * 3;                // 28 per item
constexpr int MX_NMAIN = 50;                        // (halo row) x (4 families + ninth chain), as in conv3_sb2_kernel

#ifdef RU_SB2_DBG
static __device__ unsigned long long mx_prof[8];   // devtools bit 64: section cycle counters of consumer wave 0 (layout of sb2_prof)
#endif

// GRAD (round 6, late): the same kernel for a GRADIENT input -- the data-gradient convolutions of the 16-channel level (model.py:89-91 backward).  The input arrives in
// the gradient operand form (Conv3Args::in_g16; conv3_mx_pack.hpp MXG_*): [bf16 hi ch 0-7 | hi ch 8-15 | e4m3(lo / 2^(e-8)), e4m3(g / 2^e) ch 0-7 | the same ch 8-15] per voxel with ONE exponent e per voxel
// (LDS planes 2 / 3 then hold a channel half's two correction planes each, and the gradient variant of the weight fragments is packed in that order),
// so the staging is a copy (conv3_sb2_kernel's split-form mode) plus the voxel's exponent byte, recomputed from the hi packets (the largest 15-bit pattern of the
// voxel's 16 bf16 values: ~19 VALU per position, no byte stored or loaded), the main term runs on v_mfma_f32_16x16x32_bf16 (a gradient needs bf16's
// range), and each scaled MFMA takes its data-side scale from the exponent plane: lane group b supplies block b's scale and a block is one voxel's 16 channels x
// {lo, value} (tools/mx_scale_probe3.hip), so lane (n, g) reads the byte of the voxel at tap (j, slot) = (g & 1, g >> 1) of its fragment.  bf16 residual 2^-8, e4m3 on it
// 2^-4: a 2^-12 class product (one bf16 product: 2^-8; three: 2^-16).  BST / ADD: conv3_sb2_kernel's data-gradient epilogues (GroupNorm-backward sums, residual).
template <bool GRAD = false, bool BST = false, bool ADD = false>
__global__ __launch_bounds__(512, 2) void conv3_mx_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx) {
#ifdef RU_SB2_DBG
    constexpr int dbg = RU_SB2_DBG;                 // 1 = staging waves skip convert + LDS stores, 2 = skip their loads, 4 = consumers skip the MFMAs, 8 = skip the row stores, 64 = section counters
#else
    constexpr int dbg = 0;
#endif
    constexpr int TZ = 4, TY = 8;
    using P = SB<TZ, TY>;
    constexpr int MT = P::MT, HY = P::HY, HX = P::HX, HVOLP = P::HVOLP, NROW = P::NROW;
    constexpr int BUF = 4 * HVOLP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    constexpr int EPL = (HVOLP + 15) & ~15;             // GRAD: exponent plane of a buffer (one byte per halo position), behind the images and the statistics rows
    unsigned char* elds = reinterpret_cast<unsigned char*>(smem) + 2 * BUF * 16 + SB_STAT_LDS_FLOATS * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3;
    const int ptid = tid & 255;
    const int cog = blockIdx.y;
    const int D = a.D, H = a.H, W = a.W;
    const size_t DHW = (size_t)D * H * W;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    // tile order: conv3_sb2_kernel's (XCD-compact steps; z-walk when the shape allows, with the LDS -> LDS copy of the two shared halo planes)
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int t_begin = swz;
    const int tiles_xy = nty * ntx, Pn = G / 8;
    const bool zwalk = (G % 8 == 0) && Pn > 0 && (tiles_xy % Pn == 0) && ((a.N * (tiles_xy / Pn)) % 8 == 0);
    const int zw_pps = zwalk ? tiles_xy / Pn : 1;
    const int zw_xcd = blockIdx.x % 8, zw_j = blockIdx.x / 8;
    const int nitems = zwalk ? (a.N * zw_pps / 8) * ntz : (swz < ntile ? (ntile - swz + G - 1) / G : 0);
    auto tile_of = [&](int step) {
        if (!zwalk) return t_begin + step * G;
        const int q = step / ntz, tz = step - q * ntz;
        const int panel = zw_xcd + 8 * q;
        const int n = panel / zw_pps, pb = panel - n * zw_pps;
        return (n * ntz + tz) * tiles_xy + pb * Pn + zw_j;
    };
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;
    auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
        n = tile / tiles_per_sample;
        int b = tile - n * tiles_per_sample;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty;
        const int tz = b / nty;
        z0 = tz * TZ; y0 = ty * TY; x0 = tx * 16;
    };

    if (producer) {
        // ---------------------------------------------------------------- staging waves: conv3_sb2_kernel's voxel-major path with another conversion
        mx_set_saturating_conversions();
        constexpr int NPOS = NROW * HX, NR = (NPOS + 127) / 128;
        const int hsel = (ptid >> 3) & 1;
        const int pslot = (ptid >> 4) * 8 + (ptid & 7);
        float4 v16[NR][2];
        float4 sc4[2], sh4[2];
        unsigned vmask = 0;
        const unsigned lofs = GRAD ? hsel * 16u : hsel * 32u;    // GRAD: packets hsel (bf16 hi of this half) and 2 + hsel (hsel 0: the e4m3 lo packet, 1: the e4m3 value packet)
        int pk[NR], dlt[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int p = r * 128 + pslot;
            const int row = p / HX, xc = p - row * HX;
            const int hz = row / HY, hy = row - hz * HY;
            pk[r] = hz | (hy << 8) | (xc << 16);
            dlt[r] = ((hz * H + hy) * W + xc) * 64 + (int)lofs;
        }
        // devtools bit 2048 (results wrong): the two loads of a round take whole 64-byte voxels -- lane = (position, 16-byte quarter), first load positions
        // [32 w, 32 w + 16) of the round's 128, second [32 w + 16, 32 w + 32) -- instead of half a voxel per lane: what does the access shape of the loads cost?
        int pkq[(dbg & 2048) ? NR : 1][2], dltq[(dbg & 2048) ? NR : 1][2];
        if constexpr ((dbg & 2048) != 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int p = r * 128 + (ptid >> 6) * 32 + j * 16 + ((ptid & 63) >> 2);
                    if (p >= NPOS) p = NPOS - 1;
                    const int row = p / HX, xc = p - row * HX;
                    const int hz = row / HY, hy = row - hz * HY;
                    pkq[r][j] = hz | (hy << 8) | (xc << 16);
                    dltq[r][j] = ((hz * H + hy) * W + xc) * 64 + (ptid & 3) * 16;
                }
        }
        const bool plast = (NR - 1) * 128 + pslot < NPOS;
        constexpr int CR = (2 * HY * HX) / 128;
        bool st_chain = false;
        auto issue = [&](int item) {
            if (dbg & 2) return;
            const int tile = tile_of(item);
            int n, z0, y0, x0;
            tile_origin(tile, n, z0, y0, x0);
            const float* xb = a.x + ((size_t)n * DHW) * 16;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(DHW * 64), 0x00020000);
            const int zm1 = z0 - 1, ym1 = y0 - 1, xm1 = x0 - 1;
            const int base = ((zm1 * H + ym1) * W + xm1) * 64;
            vmask = 0;
            st_chain = zwalk && CR > 0 && (item % ntz) != 0;
            auto ld_round = [&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value;
                const int gz = zm1 + (pk[r] & 0xff), gy = ym1 + ((pk[r] >> 8) & 0xff), gx = xm1 + ((pk[r] >> 16) & 0xff);
                bool ok = ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                if constexpr ((r + 1) * 128 > NPOS) ok = ok & plast;
                const unsigned ofs = ok ? (unsigned)(base + dlt[r]) : 0x80000000u;       // out of range: the load returns the zero padding
                vmask |= ok ? (1u << r) : 0u;
                if constexpr ((dbg & 2048) != 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int qz = zm1 + (pkq[r][j] & 0xff), qy = ym1 + ((pkq[r][j] >> 8) & 0xff), qx = xm1 + ((pkq[r][j] >> 16) & 0xff);
                        const bool qok = ((unsigned)qz < (unsigned)D) & ((unsigned)qy < (unsigned)H) & ((unsigned)qx < (unsigned)W);
                        v16[r][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, qok ? (unsigned)(base + dltq[r][j]) : 0x80000000u, 0, 0));
                    }
                    return;
                }
                v16[r][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ofs, 0, 0));
                v16[r][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ofs, GRAD ? 32 : 16, 0));
            };
            if (st_chain) static_for<NR - CR>([&](auto R) { ld_round(std::integral_constant<int, decltype(R)::value + CR>{}); });
            else static_for<NR>(ld_round);
            if (xform) {
                const int cofs = n * a.Cin + hsel * 8;
                sc4[0] = *reinterpret_cast<const float4*>(a.in_scale + cofs); sc4[1] = *reinterpret_cast<const float4*>(a.in_scale + cofs + 4);
                sh4[0] = *reinterpret_cast<const float4*>(a.in_shift + cofs); sh4[1] = *reinterpret_cast<const float4*>(a.in_shift + cofs + 4);
            }
        };
        auto store = [&](u32x4* buf) {
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
            if (st_chain) {                              // z-walk: halo planes 0, 1 <- planes 4, 5 of the image staged one item earlier (all four LDS planes: thread
                const u32x4* prev = buf == lds ? lds + BUF : lds;        // (hsel, p) copies packet p of planes hsel and 2 + hsel)
                u32x4 ch[CR > 0 ? CR : 1][2];
#pragma unroll
                for (int r = 0; r < CR; ++r) {
                    const int p = r * 128 + pslot;
                    ch[r][0] = prev[hsel * HVOLP + p + 4 * HY * HX];
                    ch[r][1] = prev[(2 + hsel) * HVOLP + p + 4 * HY * HX];
                }
#pragma unroll
                for (int r = 0; r < CR; ++r) {
                    const int p = r * 128 + pslot;
                    buf[hsel * HVOLP + p] = ch[r][0];
                    buf[(2 + hsel) * HVOLP + p] = ch[r][1];
                }
                if constexpr (GRAD) {
                    if (hsel == 0) {
                        const unsigned char* eprev = elds + (buf == lds ? EPL : 0);
                        unsigned char* ecur = elds + (buf == lds ? 0 : EPL);
                        unsigned char eb[CR > 0 ? CR : 1];
#pragma unroll
                        for (int r = 0; r < CR; ++r) eb[r] = eprev[r * 128 + pslot + 4 * HY * HX];
#pragma unroll
                        for (int r = 0; r < CR; ++r) ecur[r * 128 + pslot] = eb[r];
                    }
                }
            }
            auto body = [&](auto MODE, auto R0) {
                constexpr int mode = decltype(MODE)::value;
                constexpr int r0 = decltype(R0)::value;
                static_for<NR - r0>([&](auto RI) __attribute__((always_inline)) {
                    constexpr int r = r0 + decltype(RI)::value;
                    const int p = r * 128 + pslot;
                    if ((r + 1) * 128 > NPOS && p >= NPOS) return;
                    if constexpr (GRAD) {                 // the operand form is the LDS image: a copy (positions outside the volume were loaded as zeros)
                        buf[hsel * HVOLP + p] = __builtin_bit_cast(u32x4, v16[r][0]);
                        buf[(2 + hsel) * HVOLP + p] = __builtin_bit_cast(u32x4, v16[r][1]);
                        // the voxel's exponent: largest |bf16 hi| of its 16 channels (this half's packet, then the partner lane's), as the writer took it (mxg_hi8)
                        const u32x4 hq = __builtin_bit_cast(u32x4, v16[r][0]);
                        unsigned mo = hq[0] & 0x7fff0000u, me = hq[0] & 0x7fffu;
#pragma unroll
                        for (int c = 1; c < 4; ++c) {
                            const unsigned o = hq[c] & 0x7fff0000u, e = hq[c] & 0x7fffu;
                            mo = o > mo ? o : mo;
                            me = e > me ? e : me;
                        }
                        unsigned m = (mo >> 16) > me ? (mo >> 16) : me;
                        const unsigned mp = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, true);     // row_ror:8 = lane ^ 8: the partner half (hsel = bit 3 of ptid)
                        m = mp > m ? mp : m;
                        if (hsel == 0) (elds + (buf == lds ? 0 : EPL))[p] = (unsigned char)mxg_exponent_byte((m >> 7) & 0xffu);
                        return;
                    }
                    uint2* l8p = reinterpret_cast<uint2*>(buf + 2 * HVOLP + p) + hsel;
                    uint2* x8p = reinterpret_cast<uint2*>(buf + 3 * HVOLP + p) + hsel;
                    const float f[8] = {v16[r][0].x, v16[r][0].y, v16[r][0].z, v16[r][0].w, v16[r][1].x, v16[r][1].y, v16[r][1].z, v16[r][1].w};
                    float t[8];
                    if constexpr (mode == 1) {
                        if (!((vmask >> r) & 1u)) {       // the zero padding applies to the ACTIVATED tensor
                            buf[hsel * HVOLP + p] = u32x4{0u, 0u, 0u, 0u};
                            *l8p = make_uint2(0u, 0u);
                            *x8p = make_uint2(0u, 0u);
                            return;
                        }
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float u = fmaf(f[c], sc[c], sh[c]);
                            t[c] = fmaxf(u, u * slope);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) t[c] = f[c];
                    }
                    u32x4 h16;
                    unsigned l8[2], x8[2];
                    mx_split8(t, h16, l8, x8);
                    buf[hsel * HVOLP + p] = h16;
                    *l8p = make_uint2(l8[0], l8[1]);
                    *x8p = make_uint2(x8[0], x8[1]);
                });
            };
            auto dispatch = [&](auto R0) __attribute__((always_inline)) {
                if (xform) body(std::integral_constant<int, 1>{}, R0);
                else body(std::integral_constant<int, 0>{}, R0);
            };
            if (st_chain) dispatch(std::integral_constant<int, CR>{});
            else dispatch(std::integral_constant<int, 0>{});
        };
        if (nitems > 0) {
            issue(0);
            store(lds);
            if (nitems > 1) issue(1);
        }
        __syncthreads();
        for (int w = 0; w < nitems; ++w) {
            if (w + 1 < nitems) {
                store(lds + ((w + 1) & 1) * BUF);
                if (w + 2 < nitems) issue(w + 2);
            }
            __syncthreads();
        }
        __syncthreads();                                // (the consumers' closing barrier: their last statistics flush)
    } else {
        // ---------------------------------------------------------------- matrix waves: wave rw owns output plane z0 + rw (8 rows x 16 x)
        static_assert(MT == TY && MT == 8 && TZ == 4, "one output plane per consumer wave");
        const int mz = rw;
        const int kg = lane >> 4;
        int fbase[6];                                   // fp16 fragments: conv3_sb2_kernel's offsets (plane kg & 1)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int t = sb_tap(3 * f, kg >> 1);
            fbase[f] = (kg & 1) * HVOLP + ((mz + t / 9) * HY) * HX + t % 3 + (lane & 15);
        }
        fbase[4] = (kg & 1) * HVOLP + ((mz + 2) * HY + (kg >> 1)) * HX + 2 + (lane & 15);
        fbase[5] = (kg & 1) * HVOLP + ((mz + 2) * HY) * HX + 2 + (lane & 15);
        int xbase[2][2], nbase[2];                      // cross fragments: plane 2 + (kg & 1), position of (h, slot) at halo row 0 / of the ninth chain at output row 0
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int t = mx_cross_tap(h, kg >> 1, sl, 0);
                xbase[h][sl] = (2 + (kg & 1)) * HVOLP + ((mz + t / 9) * HY) * HX + t % 3 + (lane & 15);
            }
        nbase[0] = (2 + (kg & 1)) * HVOLP + ((mz + 2) * HY + 2 * (kg >> 1)) * HX + 2 + (lane & 15);
        nbase[1] = (2 + (kg & 1)) * HVOLP + ((mz + 2) * HY + ((kg >> 1) ? 2 : 1)) * HX + 2 + (lane & 15);       // (1, 1): phantom, any valid row
        // GRAD: byte offsets in the exponent plane.  Block b of a scaled MFMA = bytes 0-15 (b < 2) / 16-31 of lane groups 2 (b & 1), 2 (b & 1) + 1 = tap pair j = b & 1,
        // slot b >> 1, and its scale comes from lane group b: this lane supplies the exponent of the voxel its column meets at (j, slot) = (kg & 1, kg >> 1)
        int ebase[2] = {0, 0}, enb = 0;
        if constexpr (GRAD) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = mx_cross_tap(h, kg & 1, kg >> 1, 0);
                ebase[h] = ((mz + t / 9) * HY) * HX + t % 3 + (lane & 15);
            }
            const int dyn = 2 * (kg & 1) + (kg >> 1);     // ninth chain: (j, slot) is row tap dy = 2 j + slot; (1, 1) is the phantom (zero weights: any valid byte)
            enb = ((mz + 2) * HY + (dyn < 3 ? dyn : 2)) * HX + 2 + (lane & 15);
        }
        u32x4 wm[SB_KSTEPS];
        mx_i32x8 wx[2][3], wn;
        {
            const u32x4* wp = wfrag + (size_t)cog * MX_UNITS * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < SB_KSTEPS; ++ks) wm[ks] = wp[ks * 64];
            auto ld8 = [&](int u) -> mx_i32x8 {
                const u32x4 p0 = wp[u * 64], p1 = wp[(u + 1) * 64];
                return mx_i32x8{(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3], (int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
            };
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) wx[h][dy] = ld8(SB_KSTEPS + (3 * h + dy) * 2);
            wn = ld8(26);
        }
        f32x4 acc[MT];
        // operands swapped (voxel-major output): D[m = cout][n = voxel], a lane owns 4 consecutive couts of one voxel
        auto mm16 = [](const u32x4& av, const u32x4& wv, const f32x4& c) -> f32x4 {
            if constexpr (GRAD) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, av), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_f16x8, wv), __builtin_bit_cast(mx_f16x8, av), c, 0, 0, 0);
        };
        auto mm8 = [](const mx_i32x8& av, const mx_i32x8& wv, const f32x4& c, int esc) -> f32x4 {       // esc (GRAD): byte 0 = the E8M0 exponent of this lane's block
            if constexpr (GRAD) return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, c, 0, 0, 0, MXG_SCALE_W, 0, esc);
            else return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, c, 0, 0, 0, MX_SCALE_W, 0, MX_SCALE_ACT);
        };
        f32x4 kc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // BST: (k1, k2, thr) of this lane's 4 channels for sample kc_n
        int kc_n = -1;
        auto need_kc = [&](int n) __attribute__((always_inline)) {
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
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        const int stat_blk = blockIdx.x, stat_nblk = G;
        unsigned flushed = 0;
        int n_acc = -1;
        float* stat_lds = smem + BUF * 8;
        int pend_n = -1, pend_par = 0, par = 0;
        auto flush_stats = [&](int n) {
            if (a.stat_partials) sb_stats_to_lds<true>(s1, s2, stat_lds + (par * 4 + rw) * 32, lane);
            pend_n = n; pend_par = par; par ^= 1;
            flushed |= 1u << (n & 31);
            s1 = f32x4{0.f, 0.f, 0.f, 0.f}; s2 = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto commit_stats = [&]() {
            if (pend_n >= 0) {
                if (rw == 0 && a.stat_partials) sb_stats_commit(a, stat_lds + pend_par * 128, pend_n, cog, stat_blk, stat_nblk, lane);
                pend_n = -1;
            }
        };
        __syncthreads();                                // item 0 is staged
#ifdef RU_SB2_DBG
        const bool prof = (dbg & 64) != 0;
        unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
#endif
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
        float dbg_sink = 0.f;
        for (int w = 0; w < nitems; ++w) {
#ifdef RU_SB2_DBG
            if (prof) t0 = __builtin_readcyclecounter();
#endif
            const u32x4* buf = lds + (w & 1) * BUF;
            const int n = cn;
            commit_stats();
            const SbOut so = sb_out_prepare<true>(a, cn, ctz * TZ + mz, ctx * 16, cog, lane);
            const int ybase = cty * TY;
            if (n != n_acc) {
                if (n_acc >= 0) flush_stats(n_acc);
                n_acc = n;
            }
            need_kc(n);
            const unsigned char* ebuf = elds + (w & 1) * EPL;
            float4 radd[3], rbst[3];                     // per-row operands (residual / BST forward tensor): tile i uses slot i % 3, loaded at row i + 1, consumed at row i + 3
#pragma unroll
            for (int j = 0; j < 3; ++j) { radd[j] = make_float4(0.f, 0.f, 0.f, 0.f); rbst[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
#ifdef RU_SB2_DBG
            if (prof) { t1 = __builtin_readcyclecounter(); pt[0] += t1 - t0; t0 = t1; }
#endif
            // fragment rings: fp16 steps two ahead in a ring of three (12 registers), cross steps one ahead in a ring of two (32 registers)
            u32x4 fm[3];
            mx_i32x8 fx[2];
            int fe[2] = {0, 0};
            auto main_ofs = [&](auto MI) __attribute__((always_inline)) {
                constexpr int r = decltype(MI)::value / 5, f = decltype(MI)::value % 5;
                return fbase[f < 4 ? f : (r < 8 ? 4 : 5)] + r * HX;
            };
            auto fetch_main = [&](auto MI) __attribute__((always_inline)) {
                constexpr int mi = decltype(MI)::value;
                if constexpr (mi < MX_NMAIN) {
                    fm[mi % 3] = buf[main_ofs(MI)];
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto fetch_cross = [&](auto XI) __attribute__((always_inline)) {
                constexpr int xi = decltype(XI)::value;
                if constexpr (xi < MX_NCROSS) {
                    constexpr MxCross c = mx_cross_of(xi);
                    const int o0 = c.q < 2 ? xbase[c.q & 1][0] + c.r * HX : nbase[0] + (c.r - 2) * HX;
                    const int o1 = c.q < 2 ? xbase[c.q & 1][1] + c.r * HX : nbase[1] + (c.r - 2) * HX;
                    const u32x4 p0 = buf[o0];
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 p1 = buf[o1];
                    __builtin_amdgcn_sched_barrier(0);
                    fx[xi % 2] = mx_i32x8{(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3], (int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
                    if constexpr (GRAD) {
                        fe[xi % 2] = ebuf[c.q < 2 ? ebase[c.q & 1] + c.r * HX : enb + (c.r - 2) * HX];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            fetch_main(std::integral_constant<int, 0>{});
            fetch_main(std::integral_constant<int, 1>{});
            fetch_cross(std::integral_constant<int, 0>{});
            auto store_tile = [&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                if constexpr ((dbg & 8) != 0) dbg_sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
                else if constexpr (BST) sb_out_tile_bst<true>(a, so, ybase + i, acc[i], rbst[i % 3], kc, a.bst_slope, s1, s2, ADD ? &radd[i % 3] : nullptr);
                else sb2_out_row<true, ADD, true>(a, so, ybase + i, acc[i], radd[i % 3], s1, s2);
            };
            // one fp16 step: fragment F(f, r) feeds tiles r-2 (dy 2), r-1 (dy 1), r (dy 0); f == 4: the ninth chain's K-steps 13 (tile r-2) and 12 (tile r)
            auto main_step = [&](auto MI) __attribute__((always_inline)) {
                constexpr int mi = decltype(MI)::value, r = mi / 5, f = mi % 5;
                const u32x4 av = fm[mi % 3];
                bool fetched = false;
                auto after = [&]() __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!fetched) { fetch_main(std::integral_constant<int, mi + 2>{}); fetched = true; }
                };
                if constexpr ((dbg & 4) == 0) {
                    if constexpr (f < 4) {
                        static_for<3>([&](auto E) {
                            constexpr int dy = 2 - decltype(E)::value, i = r - dy;
                            if constexpr (i >= 0 && i < MT) {
                                constexpr bool first = f == 0 && dy == 0;           // first touch of tile i: zero C operand
                                acc[i] = first ? mm16(av, wm[3 * f + dy], f32x4{0.f, 0.f, 0.f, 0.f}) : mm16(av, wm[3 * f + dy], acc[i]);
                                after();
                            }
                        });
                    } else {
                        static_for<2>([&](auto E) {
                            constexpr int ks = decltype(E)::value == 0 ? 13 : 12, i = decltype(E)::value == 0 ? r - 2 : r;
                            if constexpr (i >= 0 && i < MT) {
                                acc[i] = mm16(av, wm[ks], acc[i]);
                                after();
                            }
                        });
                    }
                }
                if (!fetched) fetch_main(std::integral_constant<int, mi + 2>{});
                if constexpr ((BST || ADD) && f == 0 && r >= 1 && r <= MT) {     // epilogue operands of tile r-1, two rows ahead of its store (unconditional, clamped address)
                    const int yy = ybase + r - 1;
                    const size_t ri = (so.ok && yy < H) ? sb_out_index<true>(a, so, yy) : 0;
                    if constexpr (ADD) radd[(r - 1) % 3] = *reinterpret_cast<const float4*>(a.add + ri);
                    if constexpr (BST) rbst[(r - 1) % 3] = *reinterpret_cast<const float4*>(a.bst_y + ri);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto cross_step = [&](auto XI) __attribute__((always_inline)) {
                constexpr int xi = decltype(XI)::value;
                constexpr MxCross c = mx_cross_of(xi);
                const mx_i32x8 av = fx[xi % 2];
                const int esc = fe[xi % 2];
                bool fetched = false;
                auto after = [&]() __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!fetched) { fetch_cross(std::integral_constant<int, xi + 1>{}); fetched = true; }
                };
                if constexpr ((dbg & 4) == 0) {
                    if constexpr (c.q < 2) {
                        static_for<3>([&](auto E) {
                            constexpr int dy = 2 - decltype(E)::value, i = c.r - dy;
                            if constexpr (i >= 0 && i < MT) {
                                acc[i] = mm8(av, wx[c.q & 1][dy], acc[i], esc);
                                after();
                            }
                        });
                    } else {
                        acc[c.r - 2] = mm8(av, wn, acc[c.r - 2], esc);
                        after();
                    }
                }
                if (!fetched) fetch_cross(std::integral_constant<int, xi + 1>{});
            };
            // row r: M0 X0 M1 X1 M2 [N] M3 M9 -- the scaled MFMAs (32 cycles each) between the fp16 ones, so that the next fragment of either kind has a long head start
            static_for<10>([&](auto R) {
                constexpr int r = decltype(R)::value;
                constexpr int x0 = r < 2 ? 2 * r : 4 + 3 * (r - 2);
                main_step(std::integral_constant<int, 5 * r + 0>{});
                cross_step(std::integral_constant<int, x0>{});
                main_step(std::integral_constant<int, 5 * r + 1>{});
                if constexpr (r >= 3) {                     // tile r-3 was completed by row r-1: its MFMAs have drained by now
                    store_tile(std::integral_constant<int, r - 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                cross_step(std::integral_constant<int, x0 + 1>{});
                main_step(std::integral_constant<int, 5 * r + 2>{});
                if constexpr (r >= 2) cross_step(std::integral_constant<int, x0 + 2>{});
                main_step(std::integral_constant<int, 5 * r + 3>{});
                main_step(std::integral_constant<int, 5 * r + 4>{});
#ifdef RU_SB2_DBG
                if constexpr (r == 4) { if (prof) { t1 = __builtin_readcyclecounter(); pt[1] += t1 - t0; t0 = t1; } }
#endif
            });
            store_tile(std::integral_constant<int, MT - 1>{});
            ++cstep;
            if (zwalk) {
                if (++ctz == ntz) digits_of_step(cstep);
            } else {
                ctx += gx; if (ctx >= ntx) { ctx -= ntx; ++cty; }
                cty += gy; if (cty >= nty) { cty -= nty; ++ctz; }
                ctz += gz; if (ctz >= ntz) { ctz -= ntz; ++cn; }
                cn += gn;
            }
#ifdef RU_SB2_DBG
            if (prof) { t1 = __builtin_readcyclecounter(); pt[3] += t1 - t0; t0 = t1; }
#endif
            __syncthreads();
#ifdef RU_SB2_DBG
            if (prof) { t1 = __builtin_readcyclecounter(); pt[4] += t1 - t0; pt[5] += 1; }
#endif
        }
        if ((dbg & 8) && dbg_sink == 12345.678f) a.y[0] = dbg_sink;
        commit_stats();
        if (n_acc >= 0) flush_stats(n_acc);
#ifdef RU_SB2_DBG
        if (prof && rw == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) atomicAdd(&mx_prof[i], pt[i]);
            atomicAdd(&mx_prof[7], 1ull);
        }
#endif
        __syncthreads();                                // (matched by the staging waves' closing barrier) the last flush is in LDS
        commit_stats();
        if (a.stat_partials && rw == 0 && !(dbg & 8)) {  // zeros for the samples this workgroup did not touch
            for (int n = 0; n < a.N; ++n)
                if (n >= 32 || !((flushed >> n) & 1u)) sb_stats_commit(a, nullptr, n, cog, stat_blk, stat_nblk, lane);
        }
    }
    fin_tail(a.fin, a.stat_partials, smem);              // RU_FUSE_TAIL_FINALIZE: the last workgroup of the launch finalizes the partials
}

}  // namespace ru
