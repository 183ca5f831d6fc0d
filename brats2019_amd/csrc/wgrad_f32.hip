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

// One tile's staging state of a thread (vector path): global offsets of its float4 slots (clamped loads are always
// valid; `live` says whether the value is used) and where they land in LDS.
template <int NSX, int NSD>
struct W3Slots {
    int gx[NSX];     // x halo: float offset inside one channel volume (>= 0), -1 = zero fill, -2 = no slot
    int lx[NSX];     // LDS offset of element 0 of the segment inside one channel (may start 3 before the row)
    long gd[NSD];    // dy: absolute float offset, -1 = zero fill / no slot
    int n;           // sample index of the tile
};

template <int TZ, int TY, int OT, int CT>
__global__ __launch_bounds__(512, 4) void wgrad3_f32_kernel(const Wgrad3Args a, float* __restrict__ partials,
                                                           int ntz, int nty, int ntx, int ncg, int CoP, int CiP) {
    using P = W3<TZ, TY, OT, CT>;
    constexpr int DS = P::DS, CS = P::CS, HY = P::HY, HX = P::HX, HVOL = P::HVOL, TVOL = P::TVOL;
    constexpr int NROW = P::HZ * HY;
    constexpr int NT = 512;                               // 8 waves: tap group (wave & 3) x voxel half (wave >> 2)
    constexpr int NSX = (NROW * 6 + NT - 1) / NT;         // x halo: six aligned 16-byte segments per row (as conv3_f32)
    constexpr int Q4 = TVOL / 4;
    constexpr int NSD = (OT * 16 * Q4 + NT - 1) / NT;     // dy tile: float4 per thread
    constexpr int NCH = CT * 16;
    // second wave group (waves 4-7): for one (o,c) tile pair it takes the other voxel half; with OT == 2 it takes the
    // other output-channel tile instead (halves the accumulator registers: 2x2 pairs x 7 taps would not fit 128 VGPRs)
    constexpr bool SPLIT_O = OT == 2;
    constexpr int OTW = SPLIT_O ? 1 : OT;
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
    const bool vec = (W & 3) == 0;
    const float slope_eff = xform ? a.in_slope : 1.f;

    // this wave's taps and voxel half / output-channel tile
    const int tg = wave & 3, vh = wave >> 2;
    int toff[7];
    bool tval[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int tap = tg + 4 * j;
        tval[j] = tap < 27;
        const int t = tval[j] ? tap : 0;
        toff[j] = ((t / 9) * HY + (t / 3) % 3) * HX + t % 3;
    }
    f32x4 acc[7][OTW][CT];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int p = 0; p < OTW; ++p)
#pragma unroll
            for (int q = 0; q < CT; ++q) acc[j][p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int abase = (lane & 15) * DS + (lane >> 4) + (SPLIT_O ? vh * 16 * DS : 0);
    const int bbase = (lane & 15) * CS + (lane >> 4);
    const int ntile = a.N * ntz * nty * ntx;

    auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
        int b = tile;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty; b /= nty;
        const int tz = b % ntz;
        n = b / ntz;
        z0 = tz * TZ; y0 = ty * TY; x0 = tx * 16;
    };
    auto make_slots = [&](int tile, W3Slots<NSX, NSD>& sl) {
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        sl.n = n;
#pragma unroll
        for (int j = 0; j < NSX; ++j) {
            const int item = tid + j * NT;
            const int row = item / 6, q = item - row * 6;
            const int hz = row / HY, hy = row - hz * HY;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 - 4 + 4 * q;
            const bool slot = item < NROW * 6;
            const bool ok = slot && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && gx >= 0 && gx < W;
            sl.gx[j] = !slot ? -2 : (ok ? (gz * H + gy) * W + gx : -1);
            sl.lx[j] = row * HX + 4 * q - 3;
        }
#pragma unroll
        for (int j = 0; j < NSD; ++j) {
            const int e4 = tid + j * NT;
            const int ch = e4 / Q4, r = (e4 - ch * Q4) * 4;
            const int z = r / (TY * 16), y = (r / 16) % TY, x = r & 15;
            const int gz = z0 + z, gy = y0 + y, gx = x0 + x, o = o0 + ch;
            const bool ok = e4 < OT * 16 * Q4 && o < a.Cout && gz < D && gy < H && gx < W;
            sl.gd[j] = ok ? (long)(((size_t)n * a.Cout + o) * DHW + (size_t)gz * HW + (size_t)gy * W + gx) : -1;
        }
    };
    // unconditional loads from clamped addresses (a conditional load serialises: hipcc waits vmcnt(0) after each)
    auto load_x = [&](const W3Slots<NSX, NSD>& sl, int cb, float4 (&v)[NSX]) {
        const int cg = c0 + cb;
        const float* xp = a.x + ((size_t)sl.n * a.Cin + (cg < a.Cin ? cg : 0)) * DHW;
#pragma unroll
        for (int j = 0; j < NSX; ++j) v[j] = *reinterpret_cast<const float4*>(xp + (sl.gx[j] > 0 ? sl.gx[j] : 0));
    };
    auto store_x = [&](const W3Slots<NSX, NSD>& sl, int cb, const float4 (&v)[NSX]) {
        const int cg = c0 + cb;
        float sc = 1.f, sh = 0.f;
        if (xform && cg < a.Cin) { sc = a.in_scale[sl.n * a.Cin + cg]; sh = a.in_shift[sl.n * a.Cin + cg]; }
#pragma unroll
        for (int j = 0; j < NSX; ++j) {
            if (sl.gx[j] == -2) continue;
            float t[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            const bool live = sl.gx[j] >= 0 && cg < a.Cin;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[e] = fmaf(t[e], sc, sh);                       // branch-free: (1, 0, slope 1) when no transform
                t[e] = fmaxf(t[e], t[e] * slope_eff);            // LeakyReLU for 0 < slope <= 1
                t[e] = live ? t[e] : 0.f;                        // zero padding applies to the ACTIVATED tensor
            }
            const int q = (tid + j * NT) % 6;
            float* dst = xs + cb * CS + sl.lx[j];
            if (q == 0) dst[3] = t[3];
            else if (q == 5) dst[0] = t[0];
            else { dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3]; }
        }
    };
    auto load_dy = [&](const W3Slots<NSX, NSD>& sl, float4 (&dv)[NSD]) {
#pragma unroll
        for (int j = 0; j < NSD; ++j) dv[j] = *reinterpret_cast<const float4*>(a.dy + (sl.gd[j] > 0 ? sl.gd[j] : 0));
    };
    auto store_dy = [&](const W3Slots<NSX, NSD>& sl, const float4 (&dv)[NSD]) {
#pragma unroll
        for (int j = 0; j < NSD; ++j) {
            const int e4 = tid + j * NT;
            if (e4 >= OT * 16 * Q4) continue;
            const int ch = e4 / Q4, r = (e4 - ch * Q4) * 4;
            const bool live = sl.gd[j] >= 0;
            float2* dst = reinterpret_cast<float2*>(dys + ch * DS + r);     // DS is even and r % 4 == 0: 8-byte aligned
            dst[0] = live ? make_float2(dv[j].x, dv[j].y) : make_float2(0.f, 0.f);
            dst[1] = live ? make_float2(dv[j].z, dv[j].w) : make_float2(0.f, 0.f);
        }
    };
    auto compute = [&]() {
        const int zy0 = SPLIT_O ? 0 : vh * (TZ * TY / 2), zy1 = SPLIT_O ? TZ * TY : (vh + 1) * (TZ * TY / 2);
#pragma unroll 1
        for (int zy = zy0; zy < zy1; ++zy) {
            const int z = zy / TY, y = zy - z * TY;
            const int az = abase + zy * 16;
            const int bz = bbase + (z * HY + y) * HX;
#pragma unroll
            for (int xq = 0; xq < 4; ++xq) {
                float af[OTW];
#pragma unroll
                for (int p = 0; p < OTW; ++p) af[p] = dys[az + p * 16 * DS + xq * 4];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    // no branch here: the 4th wave's 7th tap (tap 27) runs on tap 0's data into an accumulator that is
                    // never written out -- a branch per tap puts every MFMA in its own basic block (LDS latency exposed)
                    float bf[CT];
#pragma unroll
                    for (int q = 0; q < CT; ++q) bf[q] = xs[bz + toff[j] + q * 16 * CS + xq * 4];
#pragma unroll
                    for (int p = 0; p < OTW; ++p)
#pragma unroll
                        for (int q = 0; q < CT; ++q)
                            acc[j][p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[p], bf[q], acc[j][p][q], 0, 0, 0);
                }
            }
        }
    };

    if (vec) {
        // ---- software-pipelined persistent loop: the global loads of tile t+1 (dy and the first PFC channels of x) are in
        // flight while tile t is on the matrix cores; they are consumed (transform + LDS store) after the next barrier.
        // two workgroups of 8 waves per CU (4 waves per SIMD): while one stages a tile the other is on the matrix cores, and
        // inside the MFMA loop the co-resident waves cover each other's LDS latency
        W3Slots<NSX, NSD> sl;
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            make_slots(tile, sl);
            float4 pdy[NSD];
            load_dy(sl, pdy);
            __syncthreads();                       // previous tile's MFMA reads are done
            constexpr int CBATCH = SPLIT_O ? 4 : 8;   // channels x NSX float4 in flight per thread (register budget: 128)
#pragma unroll 1
            for (int cb = 0; cb < NCH; cb += CBATCH) {
                float4 v[CBATCH][NSX];
#pragma unroll
                for (int c = 0; c < CBATCH; ++c) load_x(sl, cb + c, v[c]);
#pragma unroll
                for (int c = 0; c < CBATCH; ++c) store_x(sl, cb + c, v[c]);
            }
            store_dy(sl, pdy);
            __syncthreads();
            compute();
        }
    } else {
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            int n, z0, y0, x0;
            tile_origin(tile, n, z0, y0, x0);
            __syncthreads();
            // ---- scalar staging (ragged W): dy tile (zero outside the volume / beyond Cout), then the x halo tile
            for (int e = tid; e < OT * 16 * TVOL; e += NT) {
                const int ch = e / TVOL;
                const int r = e - ch * TVOL;
                const int z = r / (TY * 16), y = (r / 16) % TY, x = r & 15;
                const int gz = z0 + z, gy = y0 + y, gx = x0 + x, o = o0 + ch;
                float v = 0.f;
                if (o < a.Cout && gz < D && gy < H && gx < W) v = a.dy[((size_t)n * a.Cout + o) * DHW + (size_t)gz * HW + (size_t)gy * W + gx];
                dys[ch * DS + r] = v;
            }
            for (int e = tid; e < NCH * HVOL; e += NT) {
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
            compute();
        }
    }
    // ---- write this workgroup's two partials (one per voxel half): partials[2*blockIdx.x + vh][tap][o][c]
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (!tval[j]) continue;
        const int tap = tg + 4 * j;
        const size_t part = SPLIT_O ? (size_t)blockIdx.x : (size_t)(2 * blockIdx.x + vh);
#pragma unroll
        for (int p = 0; p < OTW; ++p)
#pragma unroll
            for (int q = 0; q < CT; ++q) {
                const int c = c0 + q * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = o0 + (p + (SPLIT_O ? vh : 0)) * 16 + (lane >> 4) * 4 + r;
                    if (o < CoP && c < CiP)
                        partials[((part * 27 + tap) * CoP + o) * CiP + c] = acc[j][p][q][r];
                }
            }
    }
}

// dw[o*so + c*sc + tap] = sum_parts partials[part][tap][o][c].  256 threads = 64 outputs x 4 partial slices (the slices are
// combined in a fixed order through LDS: deterministic), so a 512-partial reduction is 128 loads deep instead of 512.
// LO outputs per workgroup, 256 / LO slices of the partial range per output: few outputs with many partials (a 16x16 1x1x1
// gradient has 1024) want narrow blocks -- 64-wide blocks left a 256-load dependent chain per thread on 4 workgroups.
// One entry's share of the work: workgroup `blk` of the entry's workgroups (wgrad_reduce_blocks).  VEC = 4: an item is four consecutive
// input channels (one 16-byte load per partial: the rows [o][c] are contiguous and 64-byte aligned) -- the reduction is a stream of
// 7-14 MB per weight gradient and was latency-bound on 4-byte loads; the per-element summation order is the scalar kernel's.
template <int LO, int VEC>
__device__ __forceinline__ void wgrad_reduce_body(const WgradRedEntry& e, int blk, float* red /* [256 * VEC] */) {
    constexpr int NS = 256 / LO;
    const int lane_o = threadIdx.x % LO, slice = threadIdx.x / LO;
    const int i = blk * LO + lane_o;
    const int CQ = e.Cin / VEC;
    const int total = e.taps * e.Cout * CQ;
    const bool ok = i < total;
    const int ii = ok ? i : 0;
    const int c = (ii % CQ) * VEC;
    const int o = (ii / CQ) % e.Cout;
    const int tap = ii / (CQ * e.Cout);
    const size_t stride = (size_t)e.taps * e.CoP * e.CiP;
    const float* p = e.partials + ((size_t)tap * e.CoP + o) * e.CiP + c;
    const int per = (e.nparts + NS - 1) / NS;
    const int k0 = slice * per, k1 = (k0 + per < e.nparts) ? k0 + per : e.nparts;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    auto ld = [&](int k) -> vec_t {
        if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p + (size_t)k * stride); return vec_t{t.x, t.y, t.z, t.w}; }
        else return vec_t{p[(size_t)k * stride]};
    };
    vec_t s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0;
    for (; k + 7 < k1; k += 8) {
        const vec_t t0 = ld(k), t1 = ld(k + 1), t2 = ld(k + 2), t3 = ld(k + 3), t4 = ld(k + 4), t5 = ld(k + 5), t6 = ld(k + 6), t7 = ld(k + 7);
        s0 += t0; s1 += t1; s2 += t2; s3 += t3;
        s0 += t4; s1 += t5; s2 += t6; s3 += t7;
    }
    for (; k < k1; ++k) s0 += ld(k);
    const vec_t sum = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[(slice * LO + lane_o) * VEC + j] = sum[j];
    __syncthreads();
    if (slice == 0 && ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NS; ++q) t += red[(q * LO + lane_o) * VEC + j];
            const int cj = c + j;
            // split > 0: c = t*split + ci enumerates (tap t, channel ci) of a 2x2x2 conv whose gradient layout is [o][ci][8]
            // flip: the partials were computed with the operands exchanged (tap t there is tap taps-1-t of the convolution)
            const size_t dst = e.split > 0 ? (size_t)o * e.so + (size_t)(cj % e.split) * 8 + cj / e.split
                                           : (size_t)o * e.so + (size_t)cj * e.sc + (e.flip ? e.taps - 1 - tap : tap);
            e.dw[dst] = t;
        }
    }
}
static inline int wgrad_reduce_vec(const WgradRedEntry& e) { return (e.Cin % 4 == 0 && e.CiP % 4 == 0 && ((size_t)e.partials & 15) == 0) ? 4 : 1; }
static inline int wgrad_reduce_blocks(const WgradRedEntry& e) { return cdiv(e.taps * e.Cout * (e.Cin / wgrad_reduce_vec(e)), e.lo); }
__device__ __forceinline__ void wgrad_reduce_dispatch(const WgradRedEntry& e, int blk, float* red) {
    const bool v4 = (e.Cin % 4 == 0 && e.CiP % 4 == 0 && ((size_t)e.partials & 15) == 0);       // (wgrad_reduce_vec)
    if (e.lo == 16) { if (v4) wgrad_reduce_body<16, 4>(e, blk, red); else wgrad_reduce_body<16, 1>(e, blk, red); }
    else { if (v4) wgrad_reduce_body<64, 4>(e, blk, red); else wgrad_reduce_body<64, 1>(e, blk, red); }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradRedEntry e) {
    __shared__ float red[256 * 4];
    wgrad_reduce_dispatch(e, blockIdx.x, red);
}
// the queued reductions of a backward pass in one launch: entry i owns workgroups [blk0_i, blk0_{i+1})
constexpr int RU_RED_BATCH = 48;                     // 48 x 72 bytes of kernel arguments
struct WgradRedBatch { WgradRedEntry e[RU_RED_BATCH]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradRedBatch b) {
    __shared__ float red[256 * 4];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.e[i + 1].blk0) ++i;          // (scalar loop over kernel arguments)
    wgrad_reduce_dispatch(b.e[i], blockIdx.x - b.e[i].blk0, red);
}
// fixed summation order for a given (nparts, shape): deterministic.  LO outputs per workgroup: 16 for long partial lists, 64 otherwise.
int wgrad_reduce_launch(const float* partials, int nparts, int taps, int CoP, int CiP, int Cout, int Cin, float* dw, int so, int sc, int split, hipStream_t s, int flip_taps,
                        WgradRedList* defer) {
    const WgradRedEntry e{partials, dw, nparts, taps, CoP, CiP, Cout, Cin, so, sc, split, flip_taps, nparts >= 128 ? 16 : 64, 0};
    if (defer) { defer->e.push_back(e); return RU_OK; }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(e)), dim3(256), 0, s, e);
    RU_CHECK_LAUNCH("wgrad_reduce_kernel");
    return RU_OK;
}
int wgrad_reduce_flush(WgradRedList& l, hipStream_t s) {
    for (size_t i0 = 0; i0 < l.e.size(); i0 += RU_RED_BATCH) {
        WgradRedBatch b;
        b.n = (int)(l.e.size() - i0 < (size_t)RU_RED_BATCH ? l.e.size() - i0 : (size_t)RU_RED_BATCH);
        int blk = 0;
        for (int i = 0; i < b.n; ++i) {
            b.e[i] = l.e[i0 + i];
            b.e[i].blk0 = blk;
            blk += wgrad_reduce_blocks(b.e[i]);
        }
        for (int i = b.n; i < RU_RED_BATCH; ++i) b.e[i] = b.e[0];
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blk), dim3(256), 0, s, b);
        RU_CHECK_LAUNCH("wgrad_reduce_batch_kernel");
    }
    l.e.clear();
    return RU_OK;
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

static size_t wgrad3_f32_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    const W3Choice c = wgrad3_choose(N, Cin, Cout, D, H, W);
    return (size_t)2 * c.nbx * 27 * round_up(Cout, 16) * round_up(Cin, 16) * sizeof(float);
}
size_t wgrad3_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    const size_t a = wgrad3_f32_workspace_bytes(N, Cin, Cout, D, H, W), b = wgrad3_sb_workspace_bytes(N, Cin, Cout, D, H, W);
    const size_t c = wgrad3_tr_workspace_bytes(N, Cin, Cout, D, H, W);
    const size_t m = a > b ? a : b;
    return m > c ? m : c;
}

template <int TZ, int TY, int OT, int CT>
static int wgrad3_cfg(const Wgrad3Args& a, const W3Choice& c, hipStream_t s) {
    using P = W3<TZ, TY, OT, CT>;
    static PerDevice attr_done;
    const size_t lds = (size_t)P::LDS_FLOATS * sizeof(float);
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_f32_kernel<TZ, TY, OT, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad3)");
        attr_done.set();
    }
    const int CoP = round_up(a.Cout, 16), CiP = round_up(a.Cin, 16);
    dim3 grid(c.nbx, c.ngroups);
    hipLaunchKernelGGL((wgrad3_f32_kernel<TZ, TY, OT, CT>), grid, dim3(512), lds, s, a, (float*)a.ws,
                       cdiv(a.D, TZ), cdiv(a.H, TY), cdiv(a.W, 16), c.ncg, CoP, CiP);
    RU_CHECK_LAUNCH("wgrad3_f32_kernel");
    return wgrad_reduce_launch((const float*)a.ws, (OT == 2 ? 1 : 2) * c.nbx, 27, CoP, CiP, a.Cout, a.Cin, a.dw, a.Cin * 27, 27, 0, s, 0, a.defer);
}

int wgrad3_launch(const Wgrad3Args& a, hipStream_t s) {
    RU_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cout > 0 && a.D > 0 && a.H > 0 && a.W > 0, "wgrad3: bad shape");
    if (a.x_c16 || a.dy_c16) {
        RU_REQUIRE(a.mode == RU_PREC_BF16X3, "wgrad3: voxel-major tensors need the split-bf16 kernel");
        if (a.x_c16 && a.dy_c16) return wgrad3_tr_launch(a, s);
        return wgrad3_sb_launch(a, s);
    }
    if (a.mode == RU_PREC_BF16X3 && (a.W & 3) == 0) return wgrad3_sb_launch(a, s);
    const W3Choice c = wgrad3_choose(a.N, a.Cin, a.Cout, a.D, a.H, a.W);
    if (a.ws_bytes < wgrad3_f32_workspace_bytes(a.N, a.Cin, a.Cout, a.D, a.H, a.W) || !a.ws) {
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
    // fused data gradient (Wgrad1Args::dg_*): A[m = input channel][k = output channel] fragments of this workgroup's CT channel tiles
    const bool dgrad = a.dg_w != nullptr;
    float dgw[CT][OT * 4];
    if (dgrad) {
#pragma unroll
        for (int q = 0; q < CT; ++q)
#pragma unroll
            for (int ks = 0; ks < OT * 4; ++ks) dgw[q][ks] = a.dg_w[(size_t)(4 * ks + (lane >> 4)) * a.dg_ldw + c0 + q * 16 + (lane & 15)];
    }
    for (long t = blockIdx.x; t < ntot; t += gridDim.x) {
        const int n = (int)(t / nchunk);
        const size_t v0 = (size_t)(t % nchunk) * W1_VC;
        __syncthreads();
        if (a.c16) {
            // voxel-major tensors: one aligned float4 = 4 channels of one voxel; scattered into the [channel][voxel] LDS rows
            constexpr int ND = OT * 16 * W1_VC / 4 / 256, NX = CT * 16 * W1_VC / 4 / 256;
            float4 dv[ND], xv[NX];
            const int xCB0 = a.x1 ? a.C0 >> 4 : a.Cin >> 4, xCB1 = a.x1 ? (a.Cin - a.C0) >> 4 : 1;
            const float* xsecond = a.x1 ? a.x1 : a.x;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const int e4 = tid + j * 256, q = e4 & 3, r = (e4 >> 2) % W1_VC, p = e4 / (4 * W1_VC);
                const bool ok = v0 + r < V;
                const float4 t4 = *reinterpret_cast<const float4*>(a.dy + (((size_t)n * (a.Cout >> 4) + (o0 >> 4) + p) * V + (ok ? v0 + r : 0)) * 16 + 4 * q);
                dv[j] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int e4 = tid + j * 256, q = e4 & 3, r = (e4 >> 2) % W1_VC, p = e4 / (4 * W1_VC);
                const bool ok = v0 + r < V;
                const size_t vv = ok ? v0 + r : 0;
                size_t off;
                if (a.s2d) {                             // x = fine tensor of a 2x2x2 stride-2 conv: channel block (tap, cbf) at the fine voxel
                    const int CBf = a.Cin >> 7, kb = (c0 >> 4) + p, tap = kb / CBf, cbf = kb - tap * CBf;
                    const int xc = (int)(vv % a.Wc);
                    const size_t rr = vv / a.Wc;
                    const int yc = (int)(rr % a.Hc), zc = (int)(rr / a.Hc);
                    const size_t fvx = ((size_t)(2 * zc + (tap >> 2)) * (2 * a.Hc) + 2 * yc + ((tap >> 1) & 1)) * (2 * a.Wc) + 2 * xc + (tap & 1);
                    off = (((size_t)n * CBf + cbf) * (V * 8) + fvx) * 16 + 4 * q;
                } else {
                    off = (((size_t)n * (a.Cin >> 4) + (c0 >> 4) + p) * V + vv) * 16 + 4 * q;
                }
                // optional second input tensor (channel blocks >= xCB0): selects of base and block index, no branch around the load
                const int kbx = (c0 >> 4) + p;
                const bool second = kbx >= xCB0;
                const float* xsrc = second ? xsecond : a.x;
                if (!a.s2d) off = (((size_t)n * (second ? xCB1 : xCB0) + (second ? kbx - xCB0 : kbx)) * V + vv) * 16 + 4 * q;
                const float4 t4 = *reinterpret_cast<const float4*>(xsrc + off);
                xv[j] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const int e4 = tid + j * 256, q = e4 & 3, r = (e4 >> 2) % W1_VC, p = e4 / (4 * W1_VC);
                float* d = dys + (p * 16 + 4 * q) * W1_RS + r;
                d[0] = dv[j].x; d[W1_RS] = dv[j].y; d[2 * W1_RS] = dv[j].z; d[3 * W1_RS] = dv[j].w;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int e4 = tid + j * 256, q = e4 & 3, r = (e4 >> 2) % W1_VC, p = e4 / (4 * W1_VC);
                float* d = xs + (p * 16 + 4 * q) * W1_RS + r;
                d[0] = xv[j].x; d[W1_RS] = xv[j].y; d[2 * W1_RS] = xv[j].z; d[3 * W1_RS] = xv[j].w;
            }
        } else if ((V & 3) == 0) {
            // aligned float4 loads, all issued before the first LDS store (row stride 258: 8-byte aligned float2 stores)
            constexpr int Q4 = W1_VC / 4;
            constexpr int ND = OT * 16 * Q4 / 256, NX = CT * 16 * Q4 / 256;
            float4 dv[ND], xv[NX];
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const int e4 = tid + j * 256, ch = e4 / Q4, r = (e4 % Q4) * 4, o = o0 + ch;
                const bool ok = o < a.Cout && v0 + r < V;
                const float4 t4 = *reinterpret_cast<const float4*>(a.dy + (ok ? ((size_t)n * a.Cout + o) * V + v0 + r : 0));   // unconditional, clamped
                dv[j] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int e4 = tid + j * 256, ch = e4 / Q4, r = (e4 % Q4) * 4, c = c0 + ch;
                const bool ok = c < a.Cin && v0 + r < V;
                const float4 t4 = *reinterpret_cast<const float4*>(a.x + (ok ? ((size_t)n * a.Cin + c) * V + v0 + r : 0));
                xv[j] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const int e4 = tid + j * 256, ch = e4 / Q4, r = (e4 % Q4) * 4;
                float2* d = reinterpret_cast<float2*>(dys + ch * W1_RS + r);
                d[0] = make_float2(dv[j].x, dv[j].y); d[1] = make_float2(dv[j].z, dv[j].w);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int e4 = tid + j * 256, ch = e4 / Q4, r = (e4 % Q4) * 4;
                float2* d = reinterpret_cast<float2*>(xs + ch * W1_RS + r);
                d[0] = make_float2(xv[j].x, xv[j].y); d[1] = make_float2(xv[j].z, xv[j].w);
            }
        } else {
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
        if (dgrad) {
            // dx[c][v] = sum_o w[o][c] dy[o][v] on the staged dy tile: wave w owns voxels 64 w .. 64 w + 63 (four 16-voxel tiles); the result
            // tile D[m = channel][n = voxel] leaves each lane with 4 consecutive channels of one voxel = one float4 of the voxel-major output
            const int xCB0 = a.x1 ? a.C0 >> 4 : a.Cin >> 4, xCB1 = a.x1 ? (a.Cin - a.C0) >> 4 : 1;
#pragma unroll
            for (int vt = 0; vt < 4; ++vt) {
                const int vl = wave * 64 + vt * 16 + (lane & 15);
                const size_t v = v0 + vl;
#pragma unroll
                for (int q = 0; q < CT; ++q) {
                    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < OT * 4; ++ks)
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(dgw[q][ks], dys[(4 * ks + (lane >> 4)) * W1_RS + vl], d, 0, 0, 0);
                    const int kb = (c0 >> 4) + q;
                    const bool second = kb >= xCB0;
                    if (second) {                        // LeakyReLU backward from the staged x1 values (the activation's OUTPUT, model.py:422)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d[r] = xs[(q * 16 + 4 * (lane >> 4) + r) * W1_RS + vl] > 0.f ? d[r] : d[r] * a.dg_mask_slope;
                    }
                    float* dst = second ? a.dg_y1 : a.dg_y0;
                    const size_t idx = (((size_t)n * (second ? xCB1 : xCB0) + (second ? kb - xCB0 : kb)) * V + v) * 16 + 4 * (lane >> 4);
                    if (v < V && kb < (a.Cin >> 4)) *reinterpret_cast<float4*>(dst + idx) = make_float4(d[0], d[1], d[2], d[3]);
                }
            }
        }
    }
    // fold the four waves' accumulators through LDS (fixed order) and write ONE partial per workgroup
    __syncthreads();
    float* red = smem;                                   // [4][OT*CT][256]
#pragma unroll
    for (int p = 0; p < OT; ++p)
#pragma unroll
        for (int q = 0; q < CT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * OT * CT + p * CT + q) * 4 + r) * 64 + lane] = acc[p][q][r];
    __syncthreads();
    for (int e = tid; e < OT * CT * 256; e += 256) {
        const int pq = e >> 8, r = (e >> 6) & 3, ln = e & 63;
        const float v = (red[((0 * OT * CT + pq) * 4 + r) * 64 + ln] + red[((1 * OT * CT + pq) * 4 + r) * 64 + ln]) +
                        (red[((2 * OT * CT + pq) * 4 + r) * 64 + ln] + red[((3 * OT * CT + pq) * 4 + r) * 64 + ln]);
        const int p = pq / CT, q = pq - p * CT;
        const int c = c0 + q * 16 + (ln & 15), o = o0 + p * 16 + (ln >> 4) * 4 + r;
        if (o < CoP && c < CiP) partials[((size_t)blockIdx.x * CoP + o) * CiP + c] = v;
    }
}

// ---- weight gradient of the 2x2x2 stride-2 conv on voxel-major tensors (x = the FINE tensor, gathered; "input channel" kb*16 + c =
// (tap, fine channel)).  wgrad1_f32_kernel gives every 32-channel group of the 8*Cin gathered channels its own workgroups, and each
// of them re-reads the dy tile (Cin = 16: dy read 4 times, 2 units of traffic for 1.25 of data).  Here a workgroup keeps the dy tile
// of a voxel chunk in LDS and walks W1S_NG = 4 channel groups over it (accumulators for all four in registers), the x tile of the
// next group in flight while the current one is on the matrix cores.
constexpr int W1S_NG = 4;
__global__ __launch_bounds__(256, 2) void wgrad1_s2d_kernel(const Wgrad1Args a, float* __restrict__ partials, int nchunk, int ncgb, int CoP, int CiP) {
    constexpr int OT = 2, CT = 2, NG = W1S_NG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;
    float* xs = smem + OT * 16 * W1_RS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = blockIdx.y / ncgb, cgb = blockIdx.y % ncgb;
    const int o0 = og * OT * 16, cbase = cgb * NG * CT * 16;
    const size_t V = a.V;
    const int CBf = a.Cin >> 7;                          // channel blocks of the fine tensor
    const int Hf = 2 * a.Hc, Wf = 2 * a.Wc;
    f32x4 acc[NG][OT][CT];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int p = 0; p < OT; ++p)
#pragma unroll
            for (int q = 0; q < CT; ++q) acc[g][p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int abase = (lane & 15) * W1_RS + (lane >> 4) + wave * 64;
    const int q4 = tid & 3, r0 = tid >> 2;               // a thread's float4 units: channel quad q4 of voxels r0 + 64*i, block p
    constexpr int NU = 8;                                // units per thread and operand: (p = j >> 2, i = j & 3)
    const long ntot = (long)a.N * nchunk;
    for (long t = blockIdx.x; t < ntot; t += gridDim.x) {
        const int n = (int)(t / nchunk);
        const size_t v0 = (size_t)(t % nchunk) * W1_VC;
        size_t fv[4];                                    // fine corner voxel of the 4 coarse voxels of this thread
        bool okv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t v = v0 + r0 + 64 * i;
            okv[i] = v < V;
            const size_t vv = okv[i] ? v : 0;
            const int xc = (int)(vv % a.Wc);
            const size_t rr = vv / a.Wc;
            const int yc = (int)(rr % a.Hc), zc = (int)(rr / a.Hc);
            fv[i] = ((size_t)(2 * zc) * Hf + 2 * yc) * Wf + 2 * xc;
        }
        float4 dv[NU], xv[NU];
        auto load_x = [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const int p = j >> 2, i = j & 3;
                const int kb = ((cbase + g * CT * 16) >> 4) + p, tap = kb / CBf, cbf = kb - tap * CBf;
                const size_t toff = ((size_t)(tap >> 2) * Hf + ((tap >> 1) & 1)) * Wf + (tap & 1);
                const float4 t4 = *reinterpret_cast<const float4*>(a.x + (((size_t)n * CBf + cbf) * (V * 8) + fv[i] + toff) * 16 + 4 * q4);
                xv[j] = okv[i] ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto store_x = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                float* d = xs + ((j >> 2) * 16 + 4 * q4) * W1_RS + r0 + 64 * (j & 3);
                d[0] = xv[j].x; d[W1_RS] = xv[j].y; d[2 * W1_RS] = xv[j].z; d[3 * W1_RS] = xv[j].w;
            }
        };
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int p = j >> 2, i = j & 3;
            const size_t v = okv[i] ? v0 + r0 + 64 * i : 0;
            const float4 t4 = *reinterpret_cast<const float4*>(a.dy + (((size_t)n * (a.Cout >> 4) + (o0 >> 4) + p) * V + v) * 16 + 4 * q4);
            dv[j] = okv[i] ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        load_x(0);
        __syncthreads();                                 // the previous chunk's last group is off the LDS tiles
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            float* d = dys + ((j >> 2) * 16 + 4 * q4) * W1_RS + r0 + 64 * (j & 3);
            d[0] = dv[j].x; d[W1_RS] = dv[j].y; d[2 * W1_RS] = dv[j].z; d[3 * W1_RS] = dv[j].w;
        }
        store_x();
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_x(g + 1);
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
                        acc[g][p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[p], bf[q], acc[g][p][q], 0, 0, 0);
            }
            if (g + 1 < NG) {
                __syncthreads();
                store_x();
                __syncthreads();
            }
        }
    }
    // fold the four waves' accumulators through LDS (fixed order), one channel group at a time: ONE partial per workgroup
    float* red = smem;                                   // [4][OT*CT][256]
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < OT; ++p)
#pragma unroll
            for (int q = 0; q < CT; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((wave * OT * CT + p * CT + q) * 4 + r) * 64 + lane] = acc[g][p][q][r];
        __syncthreads();
        for (int e = tid; e < OT * CT * 256; e += 256) {
            const int pq = e >> 8, r = (e >> 6) & 3, ln = e & 63;
            const float v = (red[((0 * OT * CT + pq) * 4 + r) * 64 + ln] + red[((1 * OT * CT + pq) * 4 + r) * 64 + ln]) +
                            (red[((2 * OT * CT + pq) * 4 + r) * 64 + ln] + red[((3 * OT * CT + pq) * 4 + r) * 64 + ln]);
            const int p = pq / CT, q = pq - p * CT;
            const int c = cbase + g * CT * 16 + q * 16 + (ln & 15), o = o0 + p * 16 + (ln >> 4) * 4 + r;
            if (o < CoP && c < CiP) partials[((size_t)blockIdx.x * CoP + o) * CiP + c] = v;
        }
    }
}
static bool wgrad1_s2d_usable(int Cin, int Cout) { return Cout % 32 == 0 && Cin % (W1S_NG * 32) == 0; }
static int wgrad1_s2d_nbx(int N, int Cin, int Cout, size_t V) {
    const int ngroups = (Cout / 32) * (Cin / (W1S_NG * 32));
    long nbx = 512 / ngroups;                            // two resident workgroups per CU
    if (nbx < 1) nbx = 1;
    const long ntot = (long)N * (long)((V + W1_VC - 1) / W1_VC);
    return (int)(nbx > ntot ? ntot : nbx);
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
    long nbx = 512 / c.ngroups;                    // two resident workgroups per CU: one round of blocks, half the partials of 1024
    if (nbx < 1) nbx = 1;
    const long ntot = (long)N * c.nchunk;
    if (nbx > ntot) nbx = ntot;
    c.nbx = (int)nbx;
    return c;
}

size_t wgrad1_workspace_bytes(int N, int Cin, int Cout, size_t V) {
    const W1Choice c = wgrad1_choose(N, Cin, Cout, V);
    int nbx = c.nbx;
    if (wgrad1_s2d_usable(Cin, Cout)) { const int n2 = wgrad1_s2d_nbx(N, Cin, Cout, V); if (n2 > nbx) nbx = n2; }   // either kernel
    return (size_t)nbx * round_up(Cout, 16) * round_up(Cin, 16) * sizeof(float);
}

template <int OT, int CT>
static int wgrad1_cfg(const Wgrad1Args& a, const W1Choice& c, hipStream_t s) {
    static PerDevice attr_done;
    const size_t lds = (size_t)(OT + CT) * 16 * W1_RS * sizeof(float);
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1_f32_kernel<OT, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad1)");
        attr_done.set();
    }
    const int CoP = round_up(a.Cout, 16), CiP = round_up(a.Cin, 16);
    hipLaunchKernelGGL((wgrad1_f32_kernel<OT, CT>), dim3(c.nbx, c.ngroups), dim3(256), lds, s, a, (float*)a.ws, c.nchunk, c.ncg, CoP, CiP);
    RU_CHECK_LAUNCH("wgrad1_f32_kernel");
    return wgrad_reduce_launch((const float*)a.ws, c.nbx, 1, CoP, CiP, a.Cout, a.Cin, a.dw, a.ldw, 1, a.tap_split, s, 0, a.defer);
}

int wgrad1_launch(const Wgrad1Args& a, hipStream_t s) {
    RU_REQUIRE(a.N > 0 && a.Cin > 0 && a.Cout > 0 && a.V > 0 && a.ldw >= a.Cin, "wgrad1: bad shape");
    RU_REQUIRE(!a.c16 || (a.Cin % 16 == 0 && a.Cout % 16 == 0), "wgrad1: C16 tensors need channel counts that are multiples of 16");
    RU_REQUIRE(!a.s2d || (a.c16 && a.Cin % 128 == 0 && (size_t)a.Dc * a.Hc * a.Wc == a.V), "wgrad1: bad stride-2 geometry");
    RU_REQUIRE(!a.x1 || (a.c16 && !a.s2d && a.C0 > 0 && a.C0 < a.Cin && a.C0 % 16 == 0), "wgrad1: a second input tensor needs voxel-major tensors and a split at a multiple of 16");
    const W1Choice c = wgrad1_choose(a.N, a.Cin, a.Cout, a.V);
    RU_REQUIRE(!a.dg_w || (a.c16 && !a.s2d && a.Cout == c.ot * 16 && a.Cin % (c.ct * 16) == 0 && a.dg_y0 && (a.dg_y1 || !a.x1) && a.dg_ldw >= a.Cin),
               "wgrad1: the fused data gradient needs voxel-major tensors, Cout <= 32 and whole channel tiles");
    if (!a.ws || a.ws_bytes < wgrad1_workspace_bytes(a.N, a.Cin, a.Cout, a.V)) {
        set_error("wgrad1: workspace too small");
        return RU_ENOMEM;
    }
    if (a.s2d && wgrad1_s2d_usable(a.Cin, a.Cout)) {
        static PerDevice attr_done;
        const size_t lds = (size_t)4 * 16 * W1_RS * sizeof(float);
        if (!attr_done.get()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1_s2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(wgrad1_s2d)");
            attr_done.set();
        }
        const int nbx = wgrad1_s2d_nbx(a.N, a.Cin, a.Cout, a.V), ncgb = a.Cin / (W1S_NG * 32);
        hipLaunchKernelGGL(wgrad1_s2d_kernel, dim3(nbx, (a.Cout / 32) * ncgb), dim3(256), lds, s, a, (float*)a.ws, c.nchunk, ncgb, a.Cout, a.Cin);
        RU_CHECK_LAUNCH("wgrad1_s2d_kernel");
        return wgrad_reduce_launch((const float*)a.ws, nbx, 1, a.Cout, a.Cin, a.Cout, a.Cin, a.dw, a.ldw, 1, a.tap_split, s, 0, a.defer);
    }
    if (c.ot == 2 && c.ct == 2) return wgrad1_cfg<2, 2>(a, c, s);
    if (c.ot == 2) return wgrad1_cfg<2, 1>(a, c, s);
    if (c.ct == 2) return wgrad1_cfg<1, 2>(a, c, s);
    return wgrad1_cfg<1, 1>(a, c, s);
}

}  // namespace ru
