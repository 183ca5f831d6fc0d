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
// their accumulators in registers over every tile it sees), waves 4-7 PRODUCE (two aligned float4 loads per voxel half ->
// fused GroupNorm-affine + LeakyReLU on x -> hi/lo split -> one ds_write_b128 per plane); double-buffered LDS, the loads
// of item w+2 are in flight while item w+1 is converted and item w is on the matrix cores.  Partials per workgroup are
// combined in a fixed order by wgrad_reduce_kernel (wgrad_f32.hip).
#include "ru_common.h"

#include <stdlib.h>
#include <utility>

namespace ru {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void wgrad_reduce_kernel(const float* __restrict__ partials, int nparts, int taps, int CoP, int CiP, int Cout, int Cin,
                                    float* __restrict__ dw, int so, int sc, int split);

template <int... Is, class F>
__device__ __forceinline__ void wt_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wt_static_for(F&& f) {
    wt_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ void wt_split8(const float (&t)[8], u32x4& hi, u32x4& lo) {
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

template <int TZ, int TY, int OT>
struct WTR {
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = 18;
    static constexpr int XPOS = HZ * HY * HX, DPOS = TZ * TY * 16;
    static constexpr int XPLANE = XPOS * 32, DPLANE = DPOS * 32;            // bytes of one hi or lo plane (multiples of 256)
    static constexpr int X_OFF = 0, XLO_OFF = XPLANE, D_OFF = 2 * XPLANE;    // dy block p: D_OFF + p*2*DPLANE (+ DPLANE for lo)
    static constexpr int BUF = 2 * XPLANE + OT * 2 * DPLANE;                 // bytes per LDS buffer
    static constexpr int NKB = TZ * TY / 2;
    static_assert(XPLANE % 256 == 0 && DPLANE % 256 == 0 && TY % 2 == 0, "plane alignment / row pairs");
};

__device__ __forceinline__ bf16x8 wt_read_tr(const char* p0, const char* p1) {
    // two transposed reads: rows 0-3 and 4-7 of this lane group's K-slots
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// consumer wave WAVE: taps WAVE, WAVE+4, ... (7 slots; slot 6 of wave 3 is tap 27 = a dummy that repeats tap 0 and is never written).
// Per K-block the 7 taps run as two half-steps (4 + 3 taps): inside a half-step the MFMAs go product-major over the taps, so two
// MFMAs on the same accumulator are 3-4 instructions apart (back-to-back dependent MFMAs ran at ~24 cycles each), and the
// transposed reads of the NEXT half-step's fragments are issued one or two at a time between the MFMAs.
template <int TZ, int TY, int OT, int WAVE>
__device__ __forceinline__ void wtr_consume(const char* __restrict__ buf, f32x4 (&acc)[7][OT], int lane_off) {
    using P = WTR<TZ, TY, OT>;
    constexpr int HY = P::HY, HX = P::HX, NKB = P::NKB;
    constexpr int NH = NKB * 2;                                 // half-steps: (kb, taps 0-3), (kb, taps 4-6)
    bf16x8 ah[2][OT], al[2][OT];                                // dy fragments of K-block kb, double buffered by kb parity
    bf16x8 bh[2][4], bl[2][4];                                  // x fragments of the taps of a half-step, double buffered
    const char* p = buf + lane_off;
    // one transposed read pair = one MFMA operand; READ index r of half-step h: r < 2*nt -> x fragment (tap slot r>>1, hi/lo r&1),
    // then (only before a first half of a K-block) the dy fragments of that K-block
    auto read_one = [&](auto H, auto R) {
        constexpr int h = decltype(H)::value, r = decltype(R)::value;
        constexpr int kb = h / 2, half = h % 2, nt = half == 0 ? 4 : 3, set = h % 2;
        if constexpr (r < 2 * nt) {
            constexpr int j = half * 4 + (r >> 1);
            constexpr int tap = (WAVE + 4 * j < 27) ? WAVE + 4 * j : 0;
            constexpr int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            constexpr int r0 = 2 * kb, z = r0 / TY, y = r0 % TY;
            constexpr int off0 = (((z + dz) * HY + y + dy) * HX + dx) * 32, off1 = off0 + HX * 32;
            if constexpr ((r & 1) == 0) bh[set][r >> 1] = wt_read_tr(p + P::X_OFF + off0, p + P::X_OFF + off1);
            else bl[set][r >> 1] = wt_read_tr(p + P::XLO_OFF + off0, p + P::XLO_OFF + off1);
        } else {
            constexpr int ra = r - 2 * nt, q = ra >> 1;
            constexpr int off0 = (2 * kb) * 16 * 32, off1 = off0 + 16 * 32;
            constexpr int base = P::D_OFF + q * 2 * P::DPLANE;
            if constexpr ((ra & 1) == 0) ah[kb & 1][q] = wt_read_tr(p + base + off0, p + base + off1);
            else al[kb & 1][q] = wt_read_tr(p + base + P::DPLANE + off0, p + base + P::DPLANE + off1);
        }
    };
    auto nreads = [](int h) constexpr { return (h % 2 == 0) ? 8 + 2 * OT : 6; };
    wt_static_for<nreads(0)>([&](auto R) { read_one(std::integral_constant<int, 0>{}, R); });
    wt_static_for<NH>([&](auto H) {
        constexpr int h = decltype(H)::value, kb = h / 2, half = h % 2, nt = half == 0 ? 4 : 3, set = h % 2;
        constexpr int nm = 3 * nt * OT;                          // MFMAs of this half-step
        constexpr int nr = h + 1 < NH ? nreads(h + 1) : 0;       // reads of the next half-step, spread over the MFMAs
        wt_static_for<nm>([&](auto M) {
            constexpr int m = decltype(M)::value, prod = m / (nt * OT), t = (m / OT) % nt, q = m % OT, j = half * 4 + t;
            if constexpr (prod == 0) acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[kb & 1][q], bh[set][t], acc[j][q], 0, 0, 0);
            if constexpr (prod == 1) acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb & 1][q], bl[set][t], acc[j][q], 0, 0, 0);
            if constexpr (prod == 2) acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb & 1][q], bh[set][t], acc[j][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // reads [m*nr/nm, (m+1)*nr/nm) of the next half-step go here
            constexpr int ra = m * nr / nm, rb = (m + 1) * nr / nm;
            wt_static_for<rb - ra>([&](auto K) {
                read_one(std::integral_constant<int, h + 1>{}, std::integral_constant<int, ra + decltype(K)::value>{});
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

template <int TZ, int TY, int OT>
__global__ __launch_bounds__(512, 2) void wgrad3_tr_kernel(const Wgrad3Args a, float* __restrict__ partials, int ntz, int nty, int ntx, int ncg, int CoP, int CiP, int dbg) {
    // dbg (RU_WTR_DEBUG, ablation only; results are wrong when set): 1 = producers skip conversion + LDS store, 2 = producers skip the
    // global loads, 4 = consumers skip the MFMAs
    using P = WTR<TZ, TY, OT>;
    constexpr int HY = P::HY, HX = P::HX, XPOS = P::XPOS, DPOS = P::DPOS;
    extern __shared__ __attribute__((aligned(256))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3, ptid = tid & 255;
    const int og = blockIdx.y / ncg, cgp = blockIdx.y % ncg;      // output-channel group (OT blocks of 16), input-channel block
    const int D = a.D, H = a.H, W = a.W;
    const size_t DHW = (size_t)D * H * W;
    const int CBi = a.Cin >> 4, CBo = a.Cout >> 4;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;    // XCD-compact tile order per step
    const int nitems = swz < ntile ? (ntile - swz + G - 1) / G : 0;
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;

    auto tile_origin = [&](int item, int& n, int& z0, int& y0, int& x0) {
        int b = swz + item * G;
        n = b / tiles_per_sample;
        b -= n * tiles_per_sample;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty;
        const int tz = b / nty;
        z0 = tz * TZ; y0 = ty * TY; x0 = tx * 16;
    };

    if (producer) {
        // unit = (position, channel half); lanes 2k, 2k+1 hold the two halves of one position: a ds_write_b128 group of 8 lanes
        // covers 4 positions x 32 bytes = 128 contiguous bytes (conflict free)
        constexpr int NRX = (XPOS + 127) / 128, NRD = OT * ((DPOS + 127) / 128);
        const int hsel = ptid & 1, pslot = ptid >> 1;
        float4 vx[NRX][2], vd[NRD][2];
        float4 sc4[2], sh4[2];
        unsigned mx = 0, md = 0;
        auto issue = [&](int item) {
            if (dbg & 2) return;
            int n, z0, y0, x0;
            tile_origin(item, n, z0, y0, x0);
            const float* xb = a.x + ((size_t)(n * CBi + cgp) * DHW) * 16 + hsel * 8;
            mx = 0; md = 0;
#pragma unroll
            for (int r = 0; r < NRX; ++r) {
                const int p = r * 128 + pslot;
                const int row = p / HX, xc = p - row * HX;
                const int hz = row / HY, hy = row - hz * HY;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + xc - 1;
                const bool ok = p < XPOS && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                const size_t ofs = ok ? (size_t)((gz * H + gy) * W + gx) * 16 : 0;
                mx |= ok ? (1u << r) : 0u;
                vx[r][0] = *reinterpret_cast<const float4*>(xb + ofs);          // unconditional, clamped
                vx[r][1] = *reinterpret_cast<const float4*>(xb + ofs + 4);
            }
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                constexpr int RPB = (DPOS + 127) / 128;
                const int q = r / RPB, p = (r - q * RPB) * 128 + pslot;
                const int row = p >> 4, xc = p & 15;
                const int z = row / TY, y = row - z * TY;
                const int gz = z0 + z, gy = y0 + y, gx = x0 + xc;
                const bool ok = p < DPOS && gz < D && gy < H && gx < W;
                const float* db = a.dy + ((size_t)(n * CBo + og * OT + q) * DHW) * 16 + hsel * 8;
                const size_t ofs = ok ? (size_t)((gz * H + gy) * W + gx) * 16 : 0;
                md |= ok ? (1u << r) : 0u;
                vd[r][0] = *reinterpret_cast<const float4*>(db + ofs);
                vd[r][1] = *reinterpret_cast<const float4*>(db + ofs + 4);
            }
            if (xform) {
                const int cofs = n * a.Cin + cgp * 16 + hsel * 8;
                sc4[0] = *reinterpret_cast<const float4*>(a.in_scale + cofs); sc4[1] = *reinterpret_cast<const float4*>(a.in_scale + cofs + 4);
                sh4[0] = *reinterpret_cast<const float4*>(a.in_shift + cofs); sh4[1] = *reinterpret_cast<const float4*>(a.in_shift + cofs + 4);
            }
        };
        auto store = [&](char* buf) {
            if (dbg & 1) {
                if (!(dbg & 2)) {
                    float acc0 = 0.f;
#pragma unroll
                    for (int r = 0; r < NRX; ++r) acc0 += vx[r][0].x + vx[r][1].x;
#pragma unroll
                    for (int r = 0; r < NRD; ++r) acc0 += vd[r][0].x + vd[r][1].x;
                    if (acc0 == 12345.678f) buf[0] = 1;
                }
                return;
            }
            float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (xform) {
                sc[0] = sc4[0].x; sc[1] = sc4[0].y; sc[2] = sc4[0].z; sc[3] = sc4[0].w; sc[4] = sc4[1].x; sc[5] = sc4[1].y; sc[6] = sc4[1].z; sc[7] = sc4[1].w;
                sh[0] = sh4[0].x; sh[1] = sh4[0].y; sh[2] = sh4[0].z; sh[3] = sh4[0].w; sh[4] = sh4[1].x; sh[5] = sh4[1].y; sh[6] = sh4[1].z; sh[7] = sh4[1].w;
            }
#pragma unroll
            for (int r = 0; r < NRX; ++r) {
                const int p = r * 128 + pslot;
                if ((r + 1) * 128 > XPOS && p >= XPOS) continue;
                const bool ok = (mx >> r) & 1u;
                const float f[8] = {vx[r][0].x, vx[r][0].y, vx[r][0].z, vx[r][0].w, vx[r][1].x, vx[r][1].y, vx[r][1].z, vx[r][1].w};
                float t[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float u = fmaf(f[c], sc[c], sh[c]);
                    t[c] = ok ? fmaxf(u, u * slope) : 0.f;          // zero padding applies to the ACTIVATED tensor
                }
                u32x4 hi, lo;
                wt_split8(t, hi, lo);
                *reinterpret_cast<u32x4*>(buf + P::X_OFF + p * 32 + hsel * 16) = hi;
                *reinterpret_cast<u32x4*>(buf + P::XLO_OFF + p * 32 + hsel * 16) = lo;
            }
#pragma unroll
            for (int r = 0; r < NRD; ++r) {
                constexpr int RPB = (DPOS + 127) / 128;
                const int q = r / RPB, p = (r - q * RPB) * 128 + pslot;
                if (((r % RPB) + 1) * 128 > DPOS && p >= DPOS) continue;
                const bool ok = (md >> r) & 1u;
                const float f[8] = {vd[r][0].x, vd[r][0].y, vd[r][0].z, vd[r][0].w, vd[r][1].x, vd[r][1].y, vd[r][1].z, vd[r][1].w};
                float t[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) t[c] = ok ? f[c] : 0.f;
                u32x4 hi, lo;
                wt_split8(t, hi, lo);
                *reinterpret_cast<u32x4*>(buf + P::D_OFF + q * 2 * P::DPLANE + p * 32 + hsel * 16) = hi;
                *reinterpret_cast<u32x4*>(buf + P::D_OFF + q * 2 * P::DPLANE + P::DPLANE + p * 32 + hsel * 16) = lo;
            }
        };
        if (nitems > 0) {
            issue(0);
            store(lds);
            if (nitems > 1) issue(1);
        }
        __syncthreads();
        for (int w = 0; w < nitems; ++w) {
            if (w + 1 < nitems) {
                store(lds + ((w + 1) & 1) * P::BUF);
                if (w + 2 < nitems) issue(w + 2);
            }
            __syncthreads();
        }
    } else {
        f32x4 acc[7][OT];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int q = 0; q < OT; ++q) acc[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // lane part of every transposed read: position 4g + (i>>2) of the row, 8-byte column chunk i&3
        const int i16 = lane & 15, g = lane >> 4;
        const int lane_off = (4 * g + (i16 >> 2)) * 32 + (i16 & 3) * 8;
        __syncthreads();                                // item 0 is staged
        for (int w = 0; w < nitems; ++w) {
            const char* buf = lds + (w & 1) * P::BUF;
            if (dbg & 4) { __syncthreads(); continue; }
            if (rw == 0) wtr_consume<TZ, TY, OT, 0>(buf, acc, lane_off);
            else if (rw == 1) wtr_consume<TZ, TY, OT, 1>(buf, acc, lane_off);
            else if (rw == 2) wtr_consume<TZ, TY, OT, 2>(buf, acc, lane_off);
            else wtr_consume<TZ, TY, OT, 3>(buf, acc, lane_off);
            __syncthreads();
        }
        // ---- partials[blockIdx.x][tap][o][c]: D lane = (rows o = 4*(l>>4) + r, column c = l&15)
        const int o0 = og * OT * 16, c0 = cgp * 16;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int tap = rw + 4 * j;
            if (tap >= 27) continue;
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                const int c = c0 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = o0 + q * 16 + (lane >> 4) * 4 + r;
                    if (o < CoP && c < CiP) partials[(((size_t)blockIdx.x * 27 + tap) * CoP + o) * CiP + c] = acc[j][q][r];
                }
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
    return (size_t)c.nbx * 27 * Cout * Cin * sizeof(float);
}

template <int OT>
static int wtr_cfg(const Wgrad3Args& a, const WTRChoice& c, hipStream_t s) {
    using P = WTR<4, 4, OT>;
    static bool attr_done = false;
    constexpr int LDS = 2 * P::BUF;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_tr_kernel<4, 4, OT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad3_tr)");
        attr_done = true;
    }
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("RU_WTR_DEBUG"); dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL((wgrad3_tr_kernel<4, 4, OT>), dim3(c.nbx, c.ngroups), dim3(512), LDS, s, a, (float*)a.ws,
                       cdiv(a.D, 4), cdiv(a.H, 4), cdiv(a.W, 16), c.ncg, a.Cout, a.Cin, dbg);
    RU_CHECK_LAUNCH("wgrad3_tr_kernel");
    const int co = a.dw_cout > 0 ? a.dw_cout : a.Cout, ci = a.dw_cin > 0 ? a.dw_cin : a.Cin;      // real extents of dw (zero-padded operands)
    const int total = 27 * co * ci;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, s, (const float*)a.ws, c.nbx, 27, a.Cout, a.Cin,
                       co, ci, a.dw, ci * 27, 27, 0);
    RU_CHECK_LAUNCH("wgrad_reduce_kernel");
    return RU_OK;
}

int wgrad3_tr_launch(const Wgrad3Args& a, hipStream_t s) {
    RU_REQUIRE(a.x_c16 && a.dy_c16 && a.Cin % 16 == 0 && a.Cout % 16 == 0, "wgrad3_tr: needs voxel-major x and dy with channel counts %% 16 == 0");
    const WTRChoice c = wtr_choose(a.N, a.Cin, a.Cout, a.D, a.H, a.W);
    if (!a.ws || a.ws_bytes < wgrad3_tr_workspace_bytes(a.N, a.Cin, a.Cout, a.D, a.H, a.W)) {
        set_error("wgrad3_tr: workspace too small");
        return RU_ENOMEM;
    }
    if (c.ot == 2) return wtr_cfg<2>(a, c, s);
    return wtr_cfg<1>(a, c, s);
}

}  // namespace ru
