// conv3_wz32mx.hpp -- conv3_wz32_kernel (Winograd F(2,3) along z, matrix waves on 32x32 MFMA tiles; conv3_wz32.hpp) with the fp16 + MX-fp8 product scheme of
// conv3_mx.hpp: the FORWARD convolutions of the 32..128-channel levels (model.py:72-73 as used by model.py:89-91) when the caller declares its input an activation
// tensor (Conv3Args::products == 2, RU_MX).  Per transformed plane, 16-channel chunk and row pair:
//     three bf16 products (conv3_wz32_kernel):   9 taps x 3 v_mfma_f32_32x32x16_bf16                                  = 27 MFMA time units
//     here:                                      9 v_mfma_f32_32x32x16_f16 + 5 v_mfma_scale_f32_32x32x64_f8f6f4      =  9 + 5 x 2 = 19 units
//   A scaled MFMA's K = 64 is two taps x 16 channels x both cross terms: lane (row = l & 31, k-group kg = l >> 5) holds 32 bytes = [tap slot 0 | tap slot 1], kg = 0
//   reads the e4m3(lo * 2^11) section of the image against e4m3(G * 2^8) weights, kg = 1 the e4m3(value) section against e4m3(G_lo * 2^19); the nine taps pair as
//   (0,0)+(0,1), (0,2)+(1,0), (1,1)+(1,2), (2,0)+(2,1), (2,2)+phantom (wz32mx_pair_tap).  A cross fragment belongs to ONE row pair (its two slots are arbitrary
//   taps), so the cross terms do not share fragments between row pairs as the fp16 steps do: 27 + 20 x 2 = 67 ds_read_b128 per item against 54.
//   Same staging waves (wz_stage_waves<.., MX = true>: sections 0 / 1 fp16 halves, 2 e4m3 lo, 3 e4m3 value), tile, scratch protocol, combine and statistics as
//   conv3_wz32_kernel; the scales, ranges and saturation rules are conv3_mx.hpp's.
#pragma once
#include "conv3_wz.hpp"

namespace ru {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// devtools bit 128: matrix wave 0 of every workgroup adds s_memtime section sums here (cycles): [0] item setup, [1] fragment steps 0-8, [2] steps 9-17,
// [3] steps 18-26 + the last row pair's scratch write, [4] barrier, [5] items, [6] tail after the loop, [7] workgroups
static __device__ unsigned long long wz32mx_prof[8];

__global__ __launch_bounds__(512, 2) void conv3_wz32mx_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx, int nchunk) {
    constexpr int HY = WZ_HY, HX = WZ_HX, HVOLP = WZ_HVOLP, BUF = WZ_BUF;
#ifdef RU_SB2_DBG
    constexpr int dbg = RU_SB2_DBG;          // (the staging waves' devtools bits; the matrix waves honour 4 = no MFMAs, 8 = no combine, 16 = no weight refills)
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    float* scratch = smem + 2 * BUF * 4;
    float* stat_lds = scratch + WZ_SCRATCH_FLOATS;           // [generation 2][wave 4][channel 32][sum, sum2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3;
    const int cog32 = blockIdx.y;
    const int D = a.D, H = a.H, W = a.W;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;      // XCD-compact tile order (conv3_sb2_kernel)
    // column walk (wz_stage_waves): when the tiles divide evenly, a workgroup walks down z through its own columns -- the z overlap of neighbouring F(2,3) tiles (two of
    // four input planes) is then re-read from the XCD's L2 instead of fetched by two XCDs (counter traffic 1.5x the algorithmic bytes at 32 channels in the slab order)
    const bool zcol = ntile % G == 0;
    const int nsteps = zcol ? ntile / G : (swz < ntile ? (ntile - swz + G - 1) / G : 0);
    const int nitems = nsteps * nchunk;

    if (producer) {
        wz_stage_waves<dbg, true>(a, lds, rw, lane, swz, G, nitems, nchunk, tiles_per_sample, nty, ntx, zcol ? ntz : 0);
    } else {
        // ---------------------------------------------------------------- matrix waves: wave xi owns transformed plane xi
        const int xi = rw;
        const int rp = (lane >> 4) & 1, kh = lane >> 5;
        const int fb = kh * HVOLP + (xi * HY + rp) * HX + (lane & 15);                   // fragment (h, dx): + h * HX + dx; lo half: + 2 * HVOLP
        u32x4 wm[9];                                     // fp16 fragments, one per tap
        mx_i32x8 wx[5];                                  // cross fragments, one per tap pair
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wfrag), 0, (int)((size_t)gridDim.y * nchunk * WZ32MX_UNITS * 1024), 0x00020000);
        const unsigned wlane = (unsigned)lane * 16u;
        auto wbase = [&](int chunk) { return (unsigned)(((cog32 * nchunk + chunk) * 4 + xi) * WZ32MX_UNITS_XI) * 1024u; };       // scalar byte offset of this wave's 19 units
        auto wload = [&](unsigned base, int unit) __attribute__((always_inline)) {
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, base + (unsigned)(unit * 1024), 0));
        };
        auto wload8 = [&](unsigned base, int p) __attribute__((always_inline)) {
            const u32x4 p0 = wload(base, 9 + 2 * p), p1 = wload(base, 9 + 2 * p + 1);
            return mx_i32x8{(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3], (int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
        };
        {
            const unsigned wb0 = wbase(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) wm[t] = wload(wb0, t);
#pragma unroll
            for (int p = 0; p < 5; ++p) wx[p] = wload8(wb0, p);
        }
        // cross fragment of (row pair t, tap pair p): this lane's two packets of section 2 + kh
        const int xb = (2 + kh) * HVOLP + (xi * HY + rp) * HX + (lane & 15);
        f32x16 acc[4];
        auto mm16 = [](const u32x4& av, const u32x4& wv, const f32x16& c) -> f32x16 {    // operands swapped: D[m = cout][n = voxel]
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mx_f16x8, wv), __builtin_bit_cast(mx_f16x8, av), c, 0, 0, 0);
        };
        auto mm8 = [](const mx_i32x8& av, const mx_i32x8& wv, const f32x16& c) -> f32x16 {
            return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, av, c, 0, 0, 0, MX_SCALE_W, 0, MX_SCALE_ACT);
        };
        // ---- the combining role of this wave: output plane pz, row pairs 2 th and 2 th + 1, all 32 output channels
        const int pz = rw & 1, th = rw >> 1;
        f32x4 s1[4], s2[4];                              // [j]: channels 8 j + 4 kh + r
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int stat_blk = blockIdx.x, stat_nblk = G;
        unsigned flushed = 0;
        int n_acc = -1, pend_n = -1, pend_par = 0, par = 0;
        // A lane holds 16 channel sums (and 16 sums of squares) over ITS voxels; a channel's sum is over the 32 lanes of its K half.  Halving butterfly:
        // at distance 16 a lane keeps j in {0,1} or {2,3} (bit 4) and hands the other eight values over, at 8 it keeps one j (bit 3), at 4 an r pair
        // (bit 2), at 2 one r (bit 1), at 1 both lanes add: 8 + 4 + 2 + 1 + 1 = 16 shuffles per statistic where the plain butterfly needs 80 --
        // with one flush per sample of a workgroup's run the plain form cost 1-1.5 us per sample (batch-8 forward: +4 us per launch).
        auto flush_stats = [&](int n) {
            if (a.stat_partials) {
                float* sc = stat_lds + (par * 4 + rw) * 64;
                const bool b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
                float res[2];
                static_for<2>([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    f32x4 (&v)[4] = q == 0 ? s1 : s2;
                    float k8[2][4], k4[4], k2[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) k8[jj][r] = (b4 ? v[jj + 2][r] : v[jj][r]) + __shfl_xor(b4 ? v[jj][r] : v[jj + 2][r], 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) k4[r] = (b3 ? k8[1][r] : k8[0][r]) + __shfl_xor(b3 ? k8[0][r] : k8[1][r], 8);
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) k2[rr] = (b2 ? k4[rr + 2] : k4[rr]) + __shfl_xor(b2 ? k4[rr] : k4[rr + 2], 4);
                    float k1 = (b1 ? k2[1] : k2[0]) + __shfl_xor(b1 ? k2[0] : k2[1], 2);
                    k1 += __shfl_xor(k1, 1);
                    res[q] = k1;
                });
                if ((lane & 1) == 0) {
                    const int c = 8 * (2 * (int)b4 + (int)b3) + 4 * kh + 2 * (int)b2 + (int)b1;
                    sc[c * 2] = res[0]; sc[c * 2 + 1] = res[1];
                }
            }
            pend_n = n; pend_par = par; par ^= 1;
            flushed |= 1u << (n & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        };
        // lanes 0..31 of wave 0: channel `lane` of the 32, the four waves' rows of one generation summed in wave order
        auto commit_one = [&](const float* sc4w, int n) {
            if (rw == 0 && lane < 32) {
                float u1 = 0.f, u2 = 0.f;
                if (sc4w) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) { u1 += sc4w[w * 64 + lane * 2]; u2 += sc4w[w * 64 + lane * 2 + 1]; }
                }
                const int co = cog32 * 32 + lane;
                stat_publish(a.stat_partials + (((size_t)n * a.Cout + co) * stat_nblk + stat_blk) * 2, u1, u2);
            }
        };
        auto commit_stats = [&]() {
            if (pend_n >= 0) {
                if (a.stat_partials) commit_one(stat_lds + pend_par * 256, pend_n);
                pend_n = -1;
            }
        };
        // The tile whose M accumulators sit in the scratch: output transform + statistics + store, piece by piece inside the matrix loop of the
        // next item (conv3_wz_kernel's scheme).  Piece i = (row pair 2 th + (i >> 2), register quad j = i & 3).
        size_t fbase = 0;
        unsigned frs = 0, fblk = 0;
        bool fok = false;
        int fy = 0;
        f32x4 fm[3];
        const float sg = pz ? -1.f : 1.f;                // plane 0: M0 + (M1 + M2); plane 1: M1 - (M2 + M3)
        auto fin_prepare = [&](int n, int tz, int ty, int tx) {
            if (n != n_acc) {
                if (n_acc >= 0) flush_stats(n_acc);
                n_acc = n;
            }
            const int zz = tz * 2 + pz, xx = tx * 16 + (lane & 15);
            fok = zz < D && xx < W;
            const int zc = zz < D ? zz : 0;
            fbase = ((((size_t)(n * (a.Cout >> 4) + cog32 * 2) * D + zc) * H) * W + (fok ? xx : 0)) * 16 + 4 * kh;
            frs = (unsigned)W * 16u;
            fblk = (unsigned)D * (unsigned)H * (unsigned)W * 16u;        // floats between the two 16-channel blocks (the launch holds D*H*W*64 < 2^31)
            fy = ty * 8 + rp;
        };
        auto fin_load = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, j = i & 3;
            const int tt = 2 * th + (i >> 2);
            const f32x4* S = reinterpret_cast<const f32x4*>(scratch);
            fm[0] = S[(((pz + 0) * 4 + tt) * 4 + j) * 64 + lane];
            fm[1] = S[(((pz + 1) * 4 + tt) * 4 + j) * 64 + lane];
            fm[2] = S[(((pz + 2) * 4 + tt) * 4 + j) * 64 + lane];
        };
        auto fin_row = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, j = i & 3;
            const int tt = 2 * th + (i >> 2);
            f32x4 vv;
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = fm[0][r] + sg * (fm[1][r] + fm[2][r]);
            const int yy = fy + 2 * tt;
            if (!(fok && yy < H)) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[j][r] += vv[r]; s2[j][r] += vv[r] * vv[r]; }
            const size_t idx = fbase + (size_t)((j >> 1) * fblk + (j & 1) * 8u) + (size_t)(unsigned)yy * frs;
            *reinterpret_cast<float4*>(a.y + idx) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        };
        auto put_tile = [&](auto T) __attribute__((always_inline)) {                       // M_xi of row pair t goes to the scratch
            constexpr int t = decltype(T)::value;
            f32x4* S = reinterpret_cast<f32x4*>(scratch);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                S[((xi * 4 + t) * 4 + j) * 64 + lane] = f32x4{acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]};
        };
        __syncthreads();                                 // item 0 is staged
        int cn, ctz, cty, ctx;
        if (zcol) {
            int b = swz * nsteps;
            ctz = b % ntz; b /= ntz;
            ctx = b % ntx; b /= ntx;
            cty = b % nty; cn = b / nty;
        } else {
            int b = swz;
            cn = b / tiles_per_sample; b -= cn * tiles_per_sample;
            ctx = b % ntx; b /= ntx;
            cty = b % nty; ctz = b / nty;
        }
        int gn, gz, gy, gx;
        {
            int b = G;
            gx = b % ntx; b /= ntx;
            gy = b % nty; b /= nty;
            gz = b % ntz; gn = b / ntz;
        }
        bool pending = false;
        int pn = 0, ptz = 0, pty = 0, ptx = 0;
        constexpr int NSTEP = 27;                        // (halo row pair h = 0..8) x (dx = 0..2)
        int chunk = 0;
        constexpr bool prof = (dbg & 128) != 0;
        unsigned long long pt[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
        for (int w = 0; w < nitems; ++w) {
            if constexpr (prof) t0 = __builtin_readcyclecounter();
            const bool last = chunk == nchunk - 1;
            const unsigned wnext = wbase(chunk + 1 < nchunk ? chunk + 1 : 0);
            const u32x4* buf = lds + (w & 1) * BUF;
            commit_stats();
            const bool fin = pending && !(dbg & 8);      // the previous tile is combined and stored under this item's matrix work
            if (fin) fin_prepare(pn, ptz, pty, ptx);
            pending = false;
            // (the accumulators of a tile are not zeroed: the FIRST MFMA of a row pair in the tile's first chunk takes a zero C operand -- an inline constant,
            // one wave-uniform branch per row pair -- instead of 64 v_mov per tile in the matrix wave's stream)
            const bool first_chunk = chunk == 0;
            {
                const bool FIN = fin, LAST = last;
                auto frag_ofs = [&](auto S) __attribute__((always_inline)) { return fb + (decltype(S)::value / 3) * HX + decltype(S)::value % 3; };
                if constexpr (prof) { t1 = __builtin_readcyclecounter(); pt[0] += t1 - t0; t0 = t1; }
                constexpr int RING = 3, AH = RING - 1;
                u32x4 fh[RING];
                mx_i32x8 fx[2];                                 // cross fragments: step ci lives in slot ci % 2, fetched one cross step ahead
                static_for<AH>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    fh[j] = buf[frag_ofs(std::integral_constant<int, j>{})];
                });
                auto fetch_cross = [&](auto CI) __attribute__((always_inline)) {
                    constexpr int ci = decltype(CI)::value;
                    if constexpr (ci < 20) {
                        constexpr int t = ci / 5, p = ci % 5;
                        constexpr int t0_ = wz32mx_pair_tap(p, 0), t1_ = wz32mx_pair_tap(p, 1) >= 0 ? wz32mx_pair_tap(p, 1) : t0_;      // (phantom slot: any valid packet, zero weights)
                        const u32x4 p0 = buf[xb + (2 * t + t0_ / 3) * HX + t0_ % 3];
                        __builtin_amdgcn_sched_barrier(0);
                        const u32x4 p1 = buf[xb + (2 * t + t1_ / 3) * HX + t1_ % 3];
                        __builtin_amdgcn_sched_barrier(0);
                        fx[ci % 2] = mx_i32x8{(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3], (int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
                    }
                };
                fetch_cross(std::integral_constant<int, 0>{});
                static_for<NSTEP>([&](auto S) {
                    constexpr int s = decltype(S)::value, h = s / 3, dx = s % 3, cur = s % RING, nxt = (s + AH) % RING;
                    const u32x4 ah = fh[cur];
                    bool fetched = (s + AH >= NSTEP);
                    auto fetch = [&]() __attribute__((always_inline)) {
                        if constexpr (s + AH < NSTEP) {
                            fh[nxt] = buf[frag_ofs(std::integral_constant<int, (s + AH < NSTEP ? s + AH : 0)>{})];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        fetched = true;
                    };
                    // (row pair, dy) pairs of this fragment: even h: (h/2, 0) and (h/2 - 1, 2); odd h: ((h-1)/2, 1)
                    if constexpr ((dbg & 4) == 0)
                    static_for<2>([&](auto E) {
                        constexpr int e = decltype(E)::value;
                        constexpr int dy = (h & 1) ? (e == 0 ? 1 : -1) : (e == 0 ? 0 : 2);
                        constexpr int t = (h & 1) ? (h - 1) / 2 : (e == 0 ? h / 2 : h / 2 - 1);
                        if constexpr (dy >= 0 && t >= 0 && t < 4) {
                            if constexpr (dx == 0 && dy == 0) {                             // the row pair's first MFMA of this item
                                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                if (first_chunk) acc[t] = mm16(ah, wm[dy * 3 + dx], zero);
                                else acc[t] = mm16(ah, wm[dy * 3 + dx], acc[t]);
                            } else {
                                acc[t] = mm16(ah, wm[dy * 3 + dx], acc[t]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (!fetched) fetch();
                        }
                    });
                    if constexpr ((dbg & 4) == 0) { if (!fetched) fetch(); }
                    // the cross step of this fragment step: row pair t's five tap pairs sit at h = 2t+1 (pairs 0-2) and h = 2t+2 (pairs 3, 4) -- behind the row pair's
                    // first fp16 MFMA (h = 2t: its zero C operand) and in front of its scratch write (h = 2t+3)
                    constexpr int ci = (h & 1) ? ((h - 1) / 2) * 5 + dx : ((h >= 2 && dx < 2) ? (h / 2 - 1) * 5 + 3 + dx : -1);
                    if constexpr (ci >= 0) {
                        constexpr int t = ci / 5, p = ci % 5;
                        if constexpr ((dbg & 4) == 0) {
                            acc[t] = mm8(fx[ci % 2], wx[p], acc[t]);
                            __builtin_amdgcn_sched_barrier(0);
                            fetch_cross(std::integral_constant<int, ci + 1>{});
                        }
                        if constexpr (t == 3 && !(dbg & 16)) {  // the pair's weights are dead for this item: the next chunk's go into the same registers
                            wx[p] = wload8(wnext, p);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // the fp16 weights of tap (dy, dx) are dead for this item after row pair 3: h = 6 + dy
                    if constexpr (h >= 6 && !(dbg & 16)) {      // (devtools bit 16: the weights are never refilled)
                        constexpr int tapd = (h - 6) * 3 + dx;
                        wm[tapd] = wload(wnext, tapd);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // ---- bookkeeping between the MFMAs
                    if constexpr (dx == 0 && h < 8) {           // previous tile, piece h: its three M quads one step ahead of their use
                        if (FIN) fin_load(std::integral_constant<int, (h < 8 ? h : 0)>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (dx == 1 && h < 8) {
                        if (FIN) fin_row(std::integral_constant<int, (h < 8 ? h : 0)>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (prof && (s == 8 || s == 17)) { t1 = __builtin_readcyclecounter(); pt[s == 8 ? 1 : 2] += t1 - t0; t0 = t1; }
                    if constexpr (dx == 2 && (h == 3 || h == 5 || h == 7)) {      // this tile, row pair (h-3)/2: complete since h - 1
                        if (LAST && !(dbg & 8)) put_tile(std::integral_constant<int, (h >= 3 ? (h - 3) / 2 : 0)>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if (LAST && !(dbg & 8)) {
                    put_tile(std::integral_constant<int, 3>{});
                    pending = true;
                    pn = cn; ptz = ctz; pty = cty; ptx = ctx;
                }
            }
            if constexpr (prof) { t1 = __builtin_readcyclecounter(); pt[3] += t1 - t0; t0 = t1; }
            if (++chunk == nchunk) {
                chunk = 0;
                if (zcol) {
                    if (++ctz == ntz) { ctz = 0; if (++ctx == ntx) { ctx = 0; if (++cty == nty) { cty = 0; ++cn; } } }
                } else {
                    ctx += gx; if (ctx >= ntx) { ctx -= ntx; ++cty; }
                    cty += gy; if (cty >= nty) { cty -= nty; ++ctz; }
                    ctz += gz; if (ctz >= ntz) { ctz -= ntz; ++cn; }
                    cn += gn;
                }
            }
            __syncthreads();
            if constexpr (prof) { t1 = __builtin_readcyclecounter(); pt[4] += t1 - t0; }
        }
        if constexpr (prof) t0 = __builtin_readcyclecounter();
        commit_stats();
        if (pending) {                                   // the last tile of this workgroup: nothing left to hide it under
            fin_prepare(pn, ptz, pty, ptx);
            static_for<8>([&](auto I) { fin_load(I); fin_row(I); });
        }
        __syncthreads();                                 // (a flush inside that fin_prepare is in LDS now)
        commit_stats();
        if (n_acc >= 0) flush_stats(n_acc);
        __syncthreads();
        commit_stats();
        if (a.stat_partials && rw == 0) {                // zeros for the samples this workgroup did not touch
            for (int n = 0; n < a.N; ++n)
                if (n >= 32 || !((flushed >> n) & 1u)) commit_one(nullptr, n);
        }
        if constexpr (prof) {
            if (rw == 0 && lane == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) atomicAdd(&wz32mx_prof[i], pt[i]);
                atomicAdd(&wz32mx_prof[5], (unsigned long long)nitems);
                atomicAdd(&wz32mx_prof[6], (unsigned long long)(__builtin_readcyclecounter() - t0));
                atomicAdd(&wz32mx_prof[7], 1ull);
            }
        }
    }
    fin_tail(a.fin, a.stat_partials, smem);
}

}  // namespace ru
