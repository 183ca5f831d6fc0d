// wgrad_sb.hip -- weight gradient of the 3x3x3 convolution on the bf16 matrix cores with SPLIT operands
// (hi + lo bf16, products hi*hi + lo*hi + hi*lo, fp32 accumulate; see conv3_sb.hip for the numerics).
//
//   dw[o][c][tap] = sum_{n,voxel} dy[n][o][voxel] * act(x)[n][c][voxel + tap]
//
// v_mfma_f32_16x16x32_bf16: M = 16 output channels o, N = 16 input channels c, K = 32 voxels = 2 rows (y, y+1) x 16 x.
// Lane l (r = l&15, k-group g = l>>4: row g>>1, x half g&1) supplies 8 CONSECUTIVE x voxels -> one aligned 16-byte
// packet per operand:
//   dy image  dyL[hl][x-half][o][row]            row = z*TY + y of the tile, packet = 8 bf16 of x 0-7 or 8-15
//   x  image  xL [hl][column block 0..2][c][row]  row = hz*HY + hy of the halo tile; a stored row covers the 24 floats
//             [x0-4, x0+20) (the six aligned float4 segments the staging loads), column block = 8 of them
// All packets of one lane group of ds_read_b128 differ in the channel index only; the per-channel pitch is an ODD
// number of packets and the planes are multiples of 256 B apart -> every read is bank-conflict free.
// A tap (dz,dy,dx) needs the x voxels shifted by s = 3 + dx elements inside the 16-element window (packets b, b+1):
//   dx = 1: dwords 2..5 of the window (pure register selection);  dx = 0 / 2: four v_alignbit_b32 (16-bit funnel shift).
// The 4 waves of a workgroup split the 27 taps (7,7,7,6) and keep their accumulators over all tiles the (persistent)
// workgroup walks; partials are reduced by wgrad_reduce_kernel in a fixed order (wgrad_f32.hip).
#include "ru_common.h"

namespace ru {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));


template <int TZ, int TY, int OT>
struct WSB {
    static constexpr int HZ = TZ + 2, HY = TY + 2;
    static constexpr int XROWS = HZ * HY, XRP = XROWS | 1;             // odd pitch (packets) per channel
    static constexpr int DROWS = TZ * TY, DRP = DROWS | 1;
    static constexpr int XPLANE = 16 * XRP;                             // packets per (hl, column block) plane: multiple of 16
    static constexpr int DPLANE = OT * 16 * DRP;                        // packets per (hl, x-half) plane
    static constexpr int DPLANE_P = (DPLANE + 15) / 16 * 16;
    static constexpr int X_PACKETS = 2 * 3 * XPLANE, D_PACKETS = 2 * 2 * DPLANE_P;
    static constexpr int LDS_BYTES = (X_PACKETS + D_PACKETS) * 16;
    static constexpr int NKB = DROWS / 2;                               // K-blocks (row pairs) per tile
    static_assert(TY % 2 == 0, "row pairs must not straddle z");
};

// 4 floats -> 4 bf16 hi (8 bytes) + 4 bf16 lo (8 bytes)
__device__ __forceinline__ void split4(const float (&t)[4], u32x2& hi, u32x2& lo) {
    split_n<2>(t, hi, lo);
}

// the 8 elements starting `s` elements (3, 4 or 5) into the 16-element window (w0 = packet b, w1 = packet b+1)
template <int S>
__device__ __forceinline__ bf16x8 window_shift(const u32x4& w0, const u32x4& w1) {
    u32x4 r;
    if (S == 4) {
        r = u32x4{w0[2], w0[3], w1[0], w1[1]};
    } else if (S == 3) {
        r = u32x4{__builtin_amdgcn_alignbit(w0[2], w0[1], 16), __builtin_amdgcn_alignbit(w0[3], w0[2], 16),
                  __builtin_amdgcn_alignbit(w1[0], w0[3], 16), __builtin_amdgcn_alignbit(w1[1], w1[0], 16)};
    } else {
        r = u32x4{__builtin_amdgcn_alignbit(w0[3], w0[2], 16), __builtin_amdgcn_alignbit(w1[0], w0[3], 16),
                  __builtin_amdgcn_alignbit(w1[1], w1[0], 16), __builtin_amdgcn_alignbit(w1[2], w1[1], 16)};
    }
    return __builtin_bit_cast(bf16x8, r);
}

template <int TZ, int TY, int OT, int WAVE>
__device__ __forceinline__ void wsb_compute(const u32x4* __restrict__ xL, const u32x4* __restrict__ dL, f32x4 (&acc)[7][OT],
                                            int r16, int rowsel, int xh) {
    using P = WSB<TZ, TY, OT>;
    constexpr int HY = P::HY, XRP = P::XRP, DRP = P::DRP, XPLANE = P::XPLANE, DPLANE = P::DPLANE_P, NKB = P::NKB;
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        const int drow = 2 * kb + rowsel;                                 // dy row of this lane's k-group
        const int z = drow / TY, y = drow - z * TY;
        bf16x8 ah[OT], al[OT];
#pragma unroll
        for (int p = 0; p < OT; ++p) {
            const int pk = xh * DPLANE + (p * 16 + r16) * DRP + drow;
            ah[p] = __builtin_bit_cast(bf16x8, dL[pk]);
            al[p] = __builtin_bit_cast(bf16x8, dL[2 * DPLANE + pk]);
        }
        const int xbase = (xh * 16 + r16) * XRP + z * HY + y;             // column block xh, channel r16, halo row (z, y)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            // wave 3's 7th slot (tap 27) recomputes tap 0 into an accumulator that is never written out
            constexpr int dummy = 0;
            const int tap = (WAVE + 4 * j < 27) ? WAVE + 4 * j : dummy;
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int pk = xbase + dz * HY + dy;
            const u32x4 h0 = xL[pk], h1 = xL[XPLANE + pk];
            const u32x4 l0 = xL[3 * XPLANE + pk], l1 = xL[4 * XPLANE + pk];
            bf16x8 bh, bl;
            if (dx == 0) { bh = window_shift<3>(h0, h1); bl = window_shift<3>(l0, l1); }
            else if (dx == 1) { bh = window_shift<4>(h0, h1); bl = window_shift<4>(l0, l1); }
            else { bh = window_shift<5>(h0, h1); bl = window_shift<5>(l0, l1); }
#pragma unroll
            for (int p = 0; p < OT; ++p) {
                acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[p], bh, acc[j][p], 0, 0, 0);
                acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[p], bl, acc[j][p], 0, 0, 0);
                acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[p], bh, acc[j][p], 0, 0, 0);
            }
        }
    }
}

// 8 floats -> 8 bf16 hi + 8 bf16 lo (one 16-byte packet each)
__device__ __forceinline__ void wsplit8(const float (&t)[8], u32x4& hi, u32x4& lo) {
    split_n<4>(t, hi, lo);
}

// X16 / DY16: the x / dy tensor is voxel-major (C16, [N][C/16][D][H][W][16]).  The packets need 8 consecutive x voxels of
// ONE channel, so a thread loads the aligned float4 (4 channels) of 8 (x) or 4 (dy) consecutive voxels and transposes in
// registers: 4 channels x 8 voxels -> 4 hi + 4 lo packets.  Every load instruction of a wave reads whole 64-byte voxels.
template <int TZ, int TY, int OT, bool X16, bool DY16>
__global__ __launch_bounds__(256, 2) void wgrad3_sb_kernel(const Wgrad3Args a, float* __restrict__ partials,
                                                          int ntz, int nty, int ntx, int ncg, int CoP, int CiP) {
    using P = WSB<TZ, TY, OT>;
    constexpr int HY = P::HY, XRP = P::XRP, DRP = P::DRP, XPLANE = P::XPLANE, DPLANE = P::DPLANE_P;
    constexpr int XROWS = P::XROWS, DROWS = P::DROWS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* xL = reinterpret_cast<u32x4*>(smem);                  // [hl][cb][c][XRP]
    u32x4* dL = xL + P::X_PACKETS;                               // [hl][xhalf][o][DRP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = blockIdx.y / ncg, cgp = blockIdx.y % ncg;
    const int o0 = og * OT * 16, c0 = cgp * 16;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;

    f32x4 acc[7][OT];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int p = 0; p < OT; ++p) acc[j][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kg = lane >> 4, rowsel = kg >> 1, xh = kg & 1;
    const int ntile = a.N * ntz * nty * ntx;

    // staging items (tile independent part)
    constexpr int XITEMS = 16 * XROWS * 6, XBATCH = OT == 1 ? 5 : 3, NXI = ((XITEMS + 255) / 256 + XBATCH - 1) / XBATCH * XBATCH;   // float4 segments of the x halo tile
    constexpr int DITEMS = OT * 16 * DROWS * 4, NDI = (DITEMS + 255) / 256;  // float4 segments of the dy tile

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int b = tile;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty; b /= nty;
        const int tz = b % ntz;
        const int n = b / ntz;
        const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16;
        __syncthreads();                                   // previous tile's reads are done
        if constexpr (X16) {
            // ---- x halo tile, C16: item = (halo row, 8-column block, channel quad): 8 float4 loads -> 4 channels x 8 voxels
            constexpr int XI16 = XROWS * 12, NX16 = (XI16 + 255) / 256;
            const float* xb = a.x + ((size_t)(n * (a.Cin >> 4) + cgp) * DHW) * 16;
#pragma unroll 1
            for (int rd = 0; rd < NX16; ++rd) {
                const int it = tid + rd * 256;
                const int cq = it & 3, rest = it >> 2;
                const int row = rest / 3, cb = rest - row * 3;
                const int hz = row / HY, hy = row - hz * HY;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx0 = x0 - 4 + 8 * cb;
                const bool rok = it < XI16 && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H;
                const size_t rbase = rok ? ((size_t)gz * H + gy) * W : 0;
                float4 v[8];
                unsigned vm = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int gx = gx0 + j;
                    const bool ok = rok && (unsigned)gx < (unsigned)W;
                    vm |= ok ? (1u << j) : 0u;
                    v[j] = *reinterpret_cast<const float4*>(xb + (rbase + (ok ? gx : 0)) * 16 + 4 * cq);     // unconditional, clamped
                }
                if (it >= XI16) continue;
                float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xform) {
                    sc4 = *reinterpret_cast<const float4*>(a.in_scale + n * a.Cin + c0 + 4 * cq);
                    sh4 = *reinterpret_cast<const float4*>(a.in_shift + n * a.Cin + c0 + 4 * cq);
                }
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = k == 0 ? v[j].x : (k == 1 ? v[j].y : (k == 2 ? v[j].z : v[j].w));
                        const float u = fmaf(f, scv[k], shv[k]);
                        t[j] = ((vm >> j) & 1u) ? fmaxf(u, u * slope) : 0.f;       // zero padding applies to the ACTIVATED tensor
                    }
                    u32x4 hi, lo;
                    wsplit8(t, hi, lo);
                    const int pk = (cb * 16 + 4 * cq + k) * XRP + row;             // [cb][c][row]
                    xL[pk] = hi;
                    xL[3 * XPLANE + pk] = lo;
                }
            }
        } else {
            // ---- x halo tile: 16 channels x XROWS rows x 6 aligned float4 -> fused transform -> hi/lo packets (8-byte halves)
            constexpr int XB = XBATCH;                                   // float4 loads in flight per thread and batch (register budget)
    #pragma unroll 1
            for (int jb = 0; jb < NXI; jb += XB) {
                float4 v[XB];
                int live[XB];
    #pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int it = tid + (jb + u) * 256;
                    const int ch = it / (XROWS * 6), rem = it - ch * (XROWS * 6);
                    const int row = rem / 6, q = rem - row * 6;
                    const int hz = row / HY, hy = row - hz * HY;
                    const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 - 4 + 4 * q, c = c0 + ch;
                    const bool ok = it < XITEMS && c < a.Cin && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && gx >= 0 && gx < W;
                    const size_t off = ok ? ((size_t)n * a.Cin + c) * DHW + (size_t)gz * HW + (size_t)gy * W + gx : 0;
                    v[u] = *reinterpret_cast<const float4*>(a.x + off);          // unconditional, clamped
                    live[u] = ok ? 1 : 0;
                }
    #pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int it = tid + (jb + u) * 256;
                    if (it >= XITEMS) continue;
                    const int ch = it / (XROWS * 6), rem = it - ch * (XROWS * 6);
                    const int row = rem / 6, q = rem - row * 6;
                    const int c = c0 + ch < a.Cin ? c0 + ch : a.Cin - 1;
                    const float m = live[u] ? 1.f : 0.f;                         // dead lanes: (0, 0) -> exact zeros without a select
                    float sc = m, sh = 0.f;
                    if (xform) { sc = a.in_scale[n * a.Cin + c] * m; sh = a.in_shift[n * a.Cin + c] * m; }   // wave-uniform branch
                    float t[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t[e] = fmaf(t[e], sc, sh);
                        t[e] = fmaxf(t[e], t[e] * slope);                        // LeakyReLU for 0 < slope <= 1 (zero stays zero)
                    }
                    u32x2 hi, lo;
                    split4(t, hi, lo);
                    const int pk = ((q >> 1) * 16 + ch) * XRP + row;             // [cb][c][row]
                    *(reinterpret_cast<u32x2*>(xL + pk) + (q & 1)) = hi;
                    *(reinterpret_cast<u32x2*>(xL + 3 * XPLANE + pk) + (q & 1)) = lo;
                }
            }
        }
        if constexpr (DY16) {
            // ---- dy tile, C16: item = (row, 4-voxel quarter, channel quad): 4 float4 loads -> 4 channels x 4 voxels (half packets)
            constexpr int DI16 = DROWS * 4 * OT * 4, ND16 = (DI16 + 255) / 256;
#pragma unroll
            for (int rd = 0; rd < ND16; ++rd) {
                const int it = tid + rd * 256;
                const int oq = it & (OT * 4 - 1), rest = it / (OT * 4);
                const int q = rest & 3, row = rest >> 2;
                const int z = row / TY, y = row - z * TY;
                const int gz = z0 + z, gy = y0 + y;
                const bool rok = it < DI16 && gz < D && gy < H;
                const float* db = a.dy + ((size_t)(n * (a.Cout >> 4) + og * OT + (oq >> 2)) * DHW) * 16 + 4 * (oq & 3);
                const size_t rbase = rok ? ((size_t)gz * H + gy) * W : 0;
                float4 v[4];
                unsigned vm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = x0 + 4 * q + j;
                    const bool ok = rok && gx < W;
                    vm |= ok ? (1u << j) : 0u;
                    v[j] = *reinterpret_cast<const float4*>(db + (rbase + (ok ? gx : 0)) * 16);
                }
                if (it >= DI16) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float f = k == 0 ? v[j].x : (k == 1 ? v[j].y : (k == 2 ? v[j].z : v[j].w));
                        t[j] = ((vm >> j) & 1u) ? f : 0.f;
                    }
                    u32x2 hi, lo;
                    split4(t, hi, lo);
                    const int pk = (q >> 1) * DPLANE + (4 * oq + k) * DRP + row;   // [xhalf][o][row]
                    *(reinterpret_cast<u32x2*>(dL + pk) + (q & 1)) = hi;
                    *(reinterpret_cast<u32x2*>(dL + 2 * DPLANE + pk) + (q & 1)) = lo;
                }
            }
        } else {
            // ---- dy tile: OT*16 channels x DROWS rows x 4 float4
            {
                float4 v[NDI];
                int live[NDI];
    #pragma unroll
                for (int j = 0; j < NDI; ++j) {
                    const int it = tid + j * 256;
                    const int ch = it / (DROWS * 4), rem = it - ch * (DROWS * 4);
                    const int row = rem / 4, q = rem - row * 4;
                    const int z = row / TY, y = row - z * TY;
                    const int gz = z0 + z, gy = y0 + y, gx = x0 + 4 * q, o = o0 + ch;
                    const bool ok = it < DITEMS && o < a.Cout && gz < D && gy < H && gx < W;
                    const size_t off = ok ? ((size_t)n * a.Cout + o) * DHW + (size_t)gz * HW + (size_t)gy * W + gx : 0;
                    v[j] = *reinterpret_cast<const float4*>(a.dy + off);
                    live[j] = ok ? 1 : 0;
                }
    #pragma unroll
                for (int j = 0; j < NDI; ++j) {
                    const int it = tid + j * 256;
                    if (it >= DITEMS) continue;
                    const int ch = it / (DROWS * 4), rem = it - ch * (DROWS * 4);
                    const int row = rem / 4, q = rem - row * 4;
                    const float m = live[j] ? 1.f : 0.f;
                    float t[4] = {v[j].x * m, v[j].y * m, v[j].z * m, v[j].w * m};
                    u32x2 hi, lo;
                    split4(t, hi, lo);
                    const int pk = (q >> 1) * DPLANE + ch * DRP + row;           // [xhalf][o][row]
                    *(reinterpret_cast<u32x2*>(dL + pk) + (q & 1)) = hi;
                    *(reinterpret_cast<u32x2*>(dL + 2 * DPLANE + pk) + (q & 1)) = lo;
                }
            }
        }
        __syncthreads();
        // ---- K loop: NKB row pairs x this wave's 7 taps x 3 products (taps are compile-time per wave: no branch between MFMAs)
        if (wave == 0) wsb_compute<TZ, TY, OT, 0>(xL, dL, acc, r16, rowsel, xh);
        else if (wave == 1) wsb_compute<TZ, TY, OT, 1>(xL, dL, acc, r16, rowsel, xh);
        else if (wave == 2) wsb_compute<TZ, TY, OT, 2>(xL, dL, acc, r16, rowsel, xh);
        else wsb_compute<TZ, TY, OT, 3>(xL, dL, acc, r16, rowsel, xh);
    }
    // ---- partials[blockIdx.x][tap][o][c]
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int tap = wave + 4 * j;
        if (tap >= 27) continue;
#pragma unroll
        for (int p = 0; p < OT; ++p) {
            const int c = c0 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + p * 16 + (lane >> 4) * 4 + r;
                if (o < CoP && c < CiP) partials[(((size_t)blockIdx.x * 27 + tap) * CoP + o) * CiP + c] = acc[j][p][r];
            }
        }
    }
}

struct WSBChoice { int ot, nbx, ngroups, ncg; };
static WSBChoice wsb_choose(int N, int Cin, int Cout, int D, int H, int W) {
    WSBChoice c;
    const int CoP = round_up(Cout, 16), CiP = round_up(Cin, 16);
    c.ot = CoP >= 32 ? 2 : 1;
    c.ncg = CiP / 16;
    c.ngroups = cdiv(CoP, 16 * c.ot) * c.ncg;
    const int tz = c.ot == 1 ? 4 : 2;                 // one (o,c) pair: 4x4x16 tile (74 KB LDS, 2 workgroups / CU); two o-tiles: 2x4x16
    const long ntile = (long)N * cdiv(D, tz) * cdiv(H, 4) * cdiv(W, 16);
    long nbx = (c.ot == 1 ? 512 : 768) / c.ngroups;
    if (nbx < 1) nbx = 1;
    if (nbx > ntile) nbx = ntile;
    c.nbx = (int)nbx;
    return c;
}

size_t wgrad3_sb_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    const WSBChoice c = wsb_choose(N, Cin, Cout, D, H, W);
    return (size_t)c.nbx * 27 * round_up(Cout, 16) * round_up(Cin, 16) * sizeof(float);
}

template <int OT, bool X16, bool DY16>
static int wsb_cfg(const Wgrad3Args& a, const WSBChoice& c, hipStream_t s) {
    constexpr int TZ = OT == 1 ? 4 : 2;
    using P = WSB<TZ, 4, OT>;
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_sb_kernel<TZ, 4, OT, X16, DY16>), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad3_sb)");
        attr_done.set();
    }
    const int CoP = round_up(a.Cout, 16), CiP = round_up(a.Cin, 16);
    hipLaunchKernelGGL((wgrad3_sb_kernel<TZ, 4, OT, X16, DY16>), dim3(c.nbx, c.ngroups), dim3(256), P::LDS_BYTES, s, a, (float*)a.ws,
                       cdiv(a.D, TZ), cdiv(a.H, 4), cdiv(a.W, 16), c.ncg, CoP, CiP);
    RU_CHECK_LAUNCH("wgrad3_sb_kernel");
    return wgrad_reduce_launch((const float*)a.ws, c.nbx, 27, CoP, CiP, a.Cout, a.Cin, a.dw, a.Cin * 27, 27, 0, s, 0, a.defer);
}

int wgrad3_sb_launch(const Wgrad3Args& a, hipStream_t s) {
    RU_REQUIRE((a.W & 3) == 0 || (a.x_c16 && a.dy_c16), "wgrad3_sb: W must be a multiple of 4 for NCDHW tensors");
    RU_REQUIRE(!a.x_c16 || a.Cin % 16 == 0, "wgrad3_sb: C16 x needs Cin %% 16 == 0");
    RU_REQUIRE(!a.dy_c16 || a.Cout % 16 == 0, "wgrad3_sb: C16 dy needs Cout %% 16 == 0");
    const WSBChoice c = wsb_choose(a.N, a.Cin, a.Cout, a.D, a.H, a.W);
    if (!a.ws || a.ws_bytes < wgrad3_sb_workspace_bytes(a.N, a.Cin, a.Cout, a.D, a.H, a.W)) {
        set_error("wgrad3_sb: workspace too small");
        return RU_ENOMEM;
    }
    if (c.ot == 2) {
        if (a.x_c16) return a.dy_c16 ? wsb_cfg<2, true, true>(a, c, s) : wsb_cfg<2, true, false>(a, c, s);
        return a.dy_c16 ? wsb_cfg<2, false, true>(a, c, s) : wsb_cfg<2, false, false>(a, c, s);
    }
    if (a.x_c16) return a.dy_c16 ? wsb_cfg<1, true, true>(a, c, s) : wsb_cfg<1, true, false>(a, c, s);
    return a.dy_c16 ? wsb_cfg<1, false, true>(a, c, s) : wsb_cfg<1, false, false>(a, c, s);
}

}  // namespace ru
