// conv3_sb.hip -- 3x3x3 stride-1 pad-1 convolution (forward and data gradient) on the bf16 matrix cores with
// SPLIT operands: every fp32 value v is carried as hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits together) and
// each product is formed as  hi*hi + lo*hi + hi*lo  in fp32 accumulators -- 3 x v_mfma_f32_16x16x32_bf16 per K-step.
// SURVEY section 7 measured this scheme at max |dp| = 4.7e-5 on the whole network (plain bf16 operands: 3e-2, outside
// the 1e-3 bar), for 16/3 = 5.3x the f32-MFMA rate.  Same call sites as conv3_f32.hip (model.py:72-73,336,348).
//
//   HBM: activations stay NCDHW fp32 (coalesced float4 loads along W); nothing is stored in bf16.
//   GEMM: M = 16 consecutive x voxels per MFMA tile (TZ x TY x 16 voxel tile per workgroup, MT tiles per wave),
//         N = 16 output channels per workgroup (blockIdx.y), K = 2 taps x 16 input channels per K-step
//         (27 taps padded to 28 -> 14 K-steps per 16-channel chunk).
//   LDS image (built while staging: fused affine + LeakyReLU, split, pack, transpose): 16-byte packets of 8 bf16,
//         lds[hl][half][pos] with hl = hi/lo plane, half = channels 0-7 / 8-15, pos = halo-linear voxel.  The A
//         fragment of lane l (row = l&15, k-group g = l>>4) is ONE ds_read_b128 at
//         lds[hl][g&1][pos(tap(2*ks + (g>>1))) + row]: the 16 lanes of a k-group read 16 consecutive packets (256 B,
//         all 64 banks once); the four hardware lane groups of ds_read_b128 each cover 8 rows of one half-plane and the
//         other 8 rows of the second, and the half-planes are a multiple of 256 B apart -> conflict free for every tap.
//   B (weights): pre-packed on the device into per-lane fragment order (hi and lo), loaded once per chunk into
//         registers: 14 K-steps x 2 x 4 VGPRs.
//   C/D: lane holds 4 consecutive x voxels of output channel l&15 -> shared epilogue (conv3_epilogue.hpp).
#include "conv3_sb_common.hpp"
#include "conv3_wz_pack.hpp"
#include "conv3_mx_pack.hpp"

namespace ru {

constexpr int SB1_W_BYTES = SB_KSTEPS * 2 * 64 * 16;      // one (cog, chunk) weight block in LDS (one-stage kernel)
template <int TZ, int TY, bool IN16, bool OUT16>
__global__ __launch_bounds__(256, 2) void conv3_sb_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx, int nchunk) {
    using P = SB<TZ, TY>;
    constexpr int MT = P::MT, HY = P::HY, HX = P::HX, HVOLP = P::HVOLP, NSV = P::NSV, NROW = P::NROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tiles_per_sample = ntz * nty * ntx;
    const int n = b / tiles_per_sample;
    const int tis = b - n * tiles_per_sample;
    b = tis;
    const int tx = b % ntx; b /= ntx;
    const int ty = b % nty;
    const int tz = b / nty;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16;
    const int cog = blockIdx.y;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;

    // A-fragment packet offsets per K-step (hi plane; lo plane = + 2*HVOLP)
    const int mz = (wave * MT) / TY, my0 = (wave * MT) % TY;
    const int kg = lane >> 4;
    int aoff[SB_KSTEPS];
#pragma unroll
    for (int ks = 0; ks < SB_KSTEPS; ++ks) {
        int tap = sb_tap(ks, kg >> 1);
        if (tap < 0) tap = 26;                   // phantom 28th tap: its weights are zero, any valid address will do
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        aoff[ks] = (kg & 1) * HVOLP + ((mz + dz) * HY + my0 + dy) * HX + dx + (lane & 15);
    }

    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;    // neutral constants make the fused transform branch-free
    auto mm = [](const bf16x8& av, const bf16x8& wv, const f32x4& c) -> f32x4 {
        if constexpr (OUT16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, c, 0, 0, 0);     // D[m = cout][n = voxel]
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv, c, 0, 0, 0);
    };

    // ---- Software pipeline over the input-channel chunks (round 4).  The kernel runs where the grid is small -- one or two workgroups per CU,
    // e.g. the 128-channel level of a batch-1 forward -- so nothing but the workgroup itself can hide the L2 latency of a chunk's operands,
    // and with the weight fragments loaded into 112 registers at the top of every chunk that latency (~2 us) was exposed eight times per
    // launch around 0.7 us of matrix work.  Now: the weight fragments of chunk c+1 (28 KB, the same for the four waves) and, for voxel-major
    // input, its halo image are requested BEFORE the matrix loop of chunk c, travel through 7 + 8 registers per thread, and are stored
    // to LDS behind the barrier that ends chunk c; the B fragments are read from LDS (lane-linear 16-byte reads).
    constexpr int WPK = SB_KSTEPS * 2 * 64, WR = WPK / 256;          // 16-byte packets of one (cog, chunk) weight block; per thread
    static_assert(WPK % 256 == 0, "weight block is a whole number of packets per thread");
    u32x4* wl = lds + 4 * HVOLP;                                      // behind the image
    u32x4 wv[WR];
    auto issue_w = [&](int chunk) __attribute__((always_inline)) {
        const u32x4* wp = wfrag + ((size_t)(cog * nchunk + chunk) * (SB_KSTEPS * 2)) * 64 + tid;
#pragma unroll
        for (int j = 0; j < WR; ++j) wv[j] = wp[j * 256];
    };
    auto store_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < WR; ++j) wl[tid + j * 256] = wv[j];
    };
    // C16 input staging, split into issue (global loads -> registers) and store (transform / split -> LDS)
    constexpr int NPOS16 = NROW * HX, NR16 = (NPOS16 + 127) / 128;
    const int hsel = (tid >> 3) & 1;
    const int pslot = (tid >> 4) * 8 + (tid & 7);
    const bool s16 = IN16 && a.in_s16 != 0;          // split form in HBM: hi packet at float offset half*4, lo at 8 + half*4: plain copy
    float4 v16[IN16 ? NR16 : 1][2];
    unsigned vmask = 0;
    size_t vofs[IN16 ? NR16 : 1];
    if constexpr (IN16) {
#pragma unroll
        for (int r = 0; r < NR16; ++r) {
            const int p = r * 128 + pslot;
            const int row = p / HX, xc = p - row * HX;
            const int hz = row / HY, hy = row - hz * HY;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + xc - 1;
            const bool ok = p < NPOS16 && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            vofs[r] = ok ? (size_t)((gz * H + gy) * W + gx) * 16 : 0;
            vmask |= ok ? (1u << r) : 0u;
        }
    }
    auto in16_issue = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (IN16) {
            const float* xb = a.x + ((size_t)(n * nchunk + chunk) * DHW) * 16 + (s16 ? hsel * 4 : hsel * 8);
            const int second = s16 ? 8 : 4;
#pragma unroll
            for (int r = 0; r < NR16; ++r) {
                v16[r][0] = *reinterpret_cast<const float4*>(xb + vofs[r]);
                v16[r][1] = *reinterpret_cast<const float4*>(xb + vofs[r] + second);
            }
        }
    };
    auto in16_store = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (IN16) {
            constexpr int NPOS = NPOS16, NR = NR16;
            float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (xform) {
                const int cofs = n * a.Cin + chunk * 16 + hsel * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c) { sc[c] = a.in_scale[cofs + c]; sh[c] = a.in_shift[cofs + c]; }
            }
            auto body = [&](auto MODE) {                  // one wave-uniform dispatch, then a branch-free unrolled loop
                constexpr int mode = decltype(MODE)::value;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int p = r * 128 + pslot;
                    if ((r + 1) * 128 > NPOS && p >= NPOS) continue;
                    const bool ok = (vmask >> r) & 1u;
                    u32x4 hi, lo;
                    if constexpr (mode == 2) {
                        const u32x4 z = u32x4{0u, 0u, 0u, 0u};
                        hi = ok ? __builtin_bit_cast(u32x4, v16[r][0]) : z;
                        lo = ok ? __builtin_bit_cast(u32x4, v16[r][1]) : z;
                    } else {
                        const float f[8] = {v16[r][0].x, v16[r][0].y, v16[r][0].z, v16[r][0].w, v16[r][1].x, v16[r][1].y, v16[r][1].z, v16[r][1].w};
                        float t[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            if constexpr (mode == 1) {
                                const float u = fmaf(f[c], sc[c], sh[c]);
                                t[c] = ok ? fmaxf(u, u * slope) : 0.f;      // zero padding applies to the ACTIVATED tensor
                            } else {
                                t[c] = ok ? f[c] : 0.f;
                            }
                        }
                        split8(t, hi, lo);
                    }
                    lds[hsel * HVOLP + p] = hi;
                    lds[(2 + hsel) * HVOLP + p] = lo;
                }
            };
            if (s16) body(std::integral_constant<int, 2>{});
            else if (xform) body(std::integral_constant<int, 1>{});
            else body(std::integral_constant<int, 0>{});
        }
    };
    issue_w(0);
    in16_issue(0);

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        if (chunk) __syncthreads();                  // every wave is done with the image and the weight block of the previous chunk
        if constexpr (IN16) {
            // ---- C16 input: converted and stored from the registers in16_issue filled one chunk earlier
            in16_store(chunk);
        } else {
            // ---- NCDHW input (W % 4 == 0): halo row [x0-1, x0+17) = six aligned 16-byte segments [x0-4+4q, +4); slot = (row, q).
            //      Per slot and channel-half, 8 float4 loads (8 channels x 4 voxels) in flight, then transform + split + transpose
            //      into 4 hi and 4 lo packets
            int gv[NSV], lp[NSV];
#pragma unroll
            for (int j = 0; j < NSV; ++j) {
                const int item = tid + j * 256;
                const int row = item / 6, q = item - row * 6;
                const int hz = row / HY, hy = row - hz * HY;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 - 4 + 4 * q;
                const bool slot = item < NROW * 6;
                const bool ok = slot && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && gx >= 0 && gx < W;
                gv[j] = !slot ? -2 : (ok ? (gz * H + gy) * W + gx : -1);
                lp[j] = row * HX + 4 * q - 3;           // packet index of element 0 (elements outside [0,18) of the row are skipped)
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cb = chunk * 16 + half * 8;
                float sc[8], sh[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    sc[c] = 1.f; sh[c] = 0.f;
                    if (xform && cb + c < a.Cin) { sc[c] = a.in_scale[n * a.Cin + cb + c]; sh[c] = a.in_shift[n * a.Cin + cb + c]; }
                }
                float4 v[NSV][8];
#pragma unroll
                for (int j = 0; j < NSV; ++j) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        // UNCONDITIONAL load from a clamped (always valid) address; zeroed by select below.  A conditional load
                        // makes hipcc branch around it and wait vmcnt(0) per load: one load in flight per thread.
                        const int cg = cb + c < a.Cin ? cb + c : a.Cin - 1;
                        v[j][c] = *reinterpret_cast<const float4*>(a.x + ((size_t)n * a.Cin + cg) * DHW + (gv[j] > 0 ? gv[j] : 0));
                    }
                }
#pragma unroll
                for (int j = 0; j < NSV; ++j) {
                    if (gv[j] == -2) continue;
                    const bool inb = gv[j] >= 0;
                    const int q = (tid + j * 256) % 6;
                    const int e0 = q == 0 ? 3 : 0, e1 = q == 5 ? 1 : 4;        // elements of this segment inside the halo row
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e < e0 || e >= e1) continue;
                        float t[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float u = e == 0 ? v[j][c].x : (e == 1 ? v[j][c].y : (e == 2 ? v[j][c].z : v[j][c].w));
                            u = fmaf(u, sc[c], sh[c]);                          // branch-free: (1, 0, slope 1) when no transform
                            u = fmaxf(u, u * slope);                            // LeakyReLU for 0 < slope <= 1
                            t[c] = (inb && cb + c < a.Cin) ? u : 0.f;          // zero padding applies to the ACTIVATED tensor
                        }
                        u32x4 hi, lo;
                        split8(t, hi, lo);
                        lds[half * HVOLP + lp[j] + e] = hi;
                        lds[(2 + half) * HVOLP + lp[j] + e] = lo;
                    }
                }
            }
        }
        store_w();
        __syncthreads();
        if (chunk + 1 < nchunk) { issue_w(chunk + 1); in16_issue(chunk + 1); }      // in flight under this chunk's matrix loop
        // ---- 14 K-steps x MT M-tiles x 3 products
#pragma unroll
        for (int ks = 0; ks < SB_KSTEPS; ++ks) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, wl[(ks * 2 + 0) * 64 + lane]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, wl[(ks * 2 + 1) * 64 + lane]);
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, lds[aoff[ks] + i * HX]);
                al[i] = __builtin_bit_cast(bf16x8, lds[aoff[ks] + 2 * HVOLP + i * HX]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = mm(al[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = mm(ah[i], bl, acc[i]);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = mm(ah[i], bh, acc[i]);
        }
    }
    // ---- epilogue: per-wave statistics partial (block tis*4 + wave), like the persistent kernel
    constexpr int NS = OUT16 ? 4 : 1;
    f32x4 s1, s2;                                   // whole vectors: the per-row update is 2 v_pk_add + 2 v_pk_fma, no packing moves
#pragma unroll
    for (int r = 0; r < NS; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    const SbOut out = sb_out_prepare<OUT16>(a, n, z0 + mz, x0, cog, lane);
    float4 radd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) radd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.add) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int yy = y0 + my0 + i;
            radd[i] = *reinterpret_cast<const float4*>(a.add + ((out.ok && yy < H) ? sb_out_index<OUT16>(a, out, yy) : 0));
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) sb_out_tile<OUT16, NS>(a, out, y0 + my0 + i, acc[i], radd[i], s1, s2);
    // one partial per (tile, sample channel): the four waves' sums meet in LDS (free after the last K-step) and wave 0 publishes them
    if (a.stat_partials) {
        __syncthreads();
        sb_stats_to_lds<OUT16>(s1, s2, smem + 4 + wave * 32, lane);
        __syncthreads();
        if (wave == 0) sb_stats_commit(a, smem + 4, n, cog, tis, tiles_per_sample, lane);
    }
    fin_tail(a.fin, a.stat_partials, smem + 4 + 128);    // RU_FUSE_TAIL_FINALIZE
}

// ------------------------------------------------------------------ few input channels (network input: 4; head gradient: 3)
// conv3_sb2c4_kernel: the same persistent producer/consumer skeleton for Cin <= 4, where padding the channels to 16 would make
// 3/4 of the MFMA work multiplications by zero.  Input: "C4" copy [N][D][H][W][4] (pad_to_c4).  A 16-byte LDS packet holds the 4
// channels of position p AND of p+1, so one A fragment carries TWO dx taps of 4 channels: K-step = 4 tap pairs, tap pair t =
// ((dz,dy) row t>>1, side t&1: dx {0,1} or {2, -}) -> 18 pairs = 5 K-steps instead of 14, 120 MFMAs per item instead of 336; the
// kernel then runs at the speed of its output stores.  No fused input transform (neither caller has one).
constexpr int SB4_KSTEPS = 5;
__host__ __device__ constexpr int sb4_tap(int t, int half) {          // tap index of K-slot half (0: first 4 K values, 1: last 4) of pair t; -1: none
    const int r = t >> 1, side = t & 1;
    if (r >= 9) return -1;
    const int dx = side * 2 + half;
    return dx > 2 ? -1 : r * 3 + dx;
}

template <bool OUT16, bool BST>
__global__ __launch_bounds__(512, 2) void conv3_sb2c4_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx) {
    constexpr int TZ = 4, TY = 8;
    using P = SB<TZ, TY>;
    constexpr int MT = P::MT, HY = P::HY, HX = P::HX, HVOLP = P::HVOLP, NROW = P::NROW;
    constexpr int BUF = 2 * HVOLP;                      // packets per LDS buffer: [hi/lo][pos]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3, ptid = tid & 255;
    const int cog = blockIdx.y;
    const int D = a.D, H = a.H, W = a.W;
    const size_t DHW = (size_t)D * H * W;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int nitems = swz < ntile ? (ntile - swz + G - 1) / G : 0;

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
        constexpr int NPOS = NROW * HX, NR = (NPOS + 255) / 256;
        float4 va[NR], vb[NR];
        unsigned ma = 0, mb = 0;
        auto issue = [&](int item) {
            int n, z0, y0, x0;
            tile_origin(item, n, z0, y0, x0);
            const float4* xb = reinterpret_cast<const float4*>(a.x) + (size_t)n * DHW;
            ma = 0; mb = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int p = r * 256 + ptid;
                const int row = p / HX, xc = p - row * HX;
                const int hz = row / HY, hy = row - hz * HY;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + xc - 1;
                const bool rok = p < NPOS && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H;
                const bool oka = rok && (unsigned)gx < (unsigned)W, okb = rok && xc + 1 < HX && (unsigned)(gx + 1) < (unsigned)W;
                const size_t rb = rok ? (size_t)(gz * H + gy) * W : 0;
                ma |= oka ? (1u << r) : 0u;
                mb |= okb ? (1u << r) : 0u;
                va[r] = xb[rb + (oka ? gx : 0)];                 // unconditional, clamped
                vb[r] = xb[rb + (okb ? gx + 1 : 0)];
            }
        };
        auto store = [&](u32x4* buf) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int p = r * 256 + ptid;
                if ((r + 1) * 256 > NPOS && p >= NPOS) continue;
                const bool oka = (ma >> r) & 1u, okb = (mb >> r) & 1u;
                const float t[8] = {oka ? va[r].x : 0.f, oka ? va[r].y : 0.f, oka ? va[r].z : 0.f, oka ? va[r].w : 0.f,
                                    okb ? vb[r].x : 0.f, okb ? vb[r].y : 0.f, okb ? vb[r].z : 0.f, okb ? vb[r].w : 0.f};
                u32x4 hi, lo;
                split8(t, hi, lo);
                buf[p] = hi;
                buf[HVOLP + p] = lo;
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
                store(lds + ((w + 1) & 1) * BUF);
                if (w + 2 < nitems) issue(w + 2);
            }
            __syncthreads();
        }
        __syncthreads();                                // (the consumers' closing barrier: their last statistics flush)
    } else {
        const int mz = (rw * MT) / TY, my0 = (rw * MT) % TY;
        const int kg = lane >> 4;
        int aoff[SB4_KSTEPS];
#pragma unroll
        for (int ks = 0; ks < SB4_KSTEPS; ++ks) {
            int t = 4 * ks + kg;
            if (t > 17) t = 17;                          // phantom pairs 18, 19: zero weights, any valid address
            const int r = t >> 1, dz = r / 3, dy = r % 3, dxb = (t & 1) * 2;
            aoff[ks] = ((mz + dz) * HY + my0 + dy) * HX + dxb + (lane & 15);
        }
        u32x4 wreg[SB4_KSTEPS][2];
        {
            const u32x4* wp = wfrag + ((size_t)cog * (SB4_KSTEPS * 2)) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < SB4_KSTEPS; ++ks) {
                wreg[ks][0] = wp[(ks * 2 + 0) * 64];
                wreg[ks][1] = wp[(ks * 2 + 1) * 64];
            }
        }
        auto mm = [](const bf16x8& av, const bf16x8& wv, const f32x4& c) -> f32x4 {
            if constexpr (OUT16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv, c, 0, 0, 0);
        };
        constexpr int NS = OUT16 ? 4 : 1;
        f32x4 s1, s2;                                   // whole vectors: the per-row update is 2 v_pk_add + 2 v_pk_fma, no packing moves
#pragma unroll
        for (int r = 0; r < NS; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        const int stat_blk = blockIdx.x, stat_nblk = G;   // one partial per (workgroup, sample): see sb_stats_to_lds / sb_stats_commit
        unsigned flushed = 0;
        int n_acc = -1;
        float* stat_lds = smem + BUF * 8;                // behind the two image buffers
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
        f32x4 acc[MT];
        __syncthreads();                                // item 0 is staged
        int cn, ctz, cty, ctx, gn, gz, gy, gx;           // (sample, tz, ty, tx) of the current tile and of the stride G
        {
            int b = swz;
            cn = b / tiles_per_sample; b -= cn * tiles_per_sample;
            ctx = b % ntx; b /= ntx;
            cty = b % nty; ctz = b / nty;
            b = G;
            gx = b % ntx; b /= ntx;
            gy = b % nty; b /= nty;
            gz = b % ntz; gn = b / ntz;
        }
        for (int w = 0; w < nitems; ++w) {
            const u32x4* buf = lds + (w & 1) * BUF;
            commit_stats();                              // (a flush of the previous item is complete in LDS since that item's barrier)
            // output rows of this tile; the per-row operand of the epilogue (residual, or BST: the forward tensor) is requested now and
            // lands under the MFMAs (it used to be loaded after them, with the stores waiting on it)
            const int n = cn, z0 = ctz * TZ, y0 = cty * TY, x0 = ctx * 16;
            ctx += gx; if (ctx >= ntx) { ctx -= ntx; ++cty; }      // digits of the next tile (+G): scalar adds with carries, no divisions
            cty += gy; if (cty >= nty) { cty -= nty; ++ctz; }
            ctz += gz; if (ctz >= ntz) { ctz -= ntz; ++cn; }
            cn += gn;
            const SbOut out = sb_out_prepare<OUT16>(a, n, z0 + mz, x0, cog, lane);
            float4 radd[MT];
            const float* src = BST ? a.bst_y : a.add;            // BST: the forward tensor of the GroupNorm this gradient enters (Conv3Args::bst_*)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int yy = y0 + my0 + i;
                radd[i] = src ? *reinterpret_cast<const float4*>(src + ((out.ok && yy < H) ? sb_out_index<OUT16>(a, out, yy) : 0))
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, buf[aoff[0] + i * HX]);
                al[i] = __builtin_bit_cast(bf16x8, buf[aoff[0] + HVOLP + i * HX]);
            }
            static_for<SB4_KSTEPS>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                constexpr bool more = ks + 1 < SB4_KSTEPS;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[ks][0]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[ks][1]);
                const int nofs = aoff[more ? ks + 1 : ks];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if constexpr (ks == 0) acc[i] = mm(al[i], bh, f32x4{0.f, 0.f, 0.f, 0.f});      // starts from the zero operand
                    else acc[i] = mm(al[i], bh, acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && i > 0) {
                        al[i - 1] = __builtin_bit_cast(bf16x8, buf[nofs + HVOLP + (i - 1) * HX]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    acc[i] = mm(ah[i], bl, acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && i == 0) {
                        al[MT - 1] = __builtin_bit_cast(bf16x8, buf[nofs + HVOLP + (MT - 1) * HX]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    acc[i] = mm(ah[i], bh, acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        ah[i] = __builtin_bit_cast(bf16x8, buf[nofs + i * HX]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            });
            if (n != n_acc) {
                if (n_acc >= 0) flush_stats(n_acc);
                n_acc = n;
            }
            if constexpr (BST) {
                f32x4 kc[3];
                const float* kp = a.bst_k + (size_t)n * 3 * a.Cout + cog * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 q = *reinterpret_cast<const float4*>(kp + (size_t)t * a.Cout);
                    kc[t] = f32x4{q.x, q.y, q.z, q.w};
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) sb_out_tile_bst(a, out, y0 + my0 + i, acc[i], radd[i], kc, a.bst_slope, s1, s2);
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) sb_out_tile<OUT16, NS>(a, out, y0 + my0 + i, acc[i], radd[i], s1, s2);
            }
            __syncthreads();
        }
        commit_stats();
        if (n_acc >= 0) flush_stats(n_acc);
        __syncthreads();                                // (matched by the producers' closing barrier) the last flush is in LDS
        commit_stats();
        if (a.stat_partials && rw == 0) {                // zeros for the samples this workgroup did not touch
            for (int n = 0; n < a.N; ++n)
                if (n >= 32 || !((flushed >> n) & 1u)) sb_stats_commit(a, nullptr, n, cog, stat_blk, stat_nblk, lane);
        }
    }
    fin_tail(a.fin, a.stat_partials, smem);              // RU_FUSE_TAIL_FINALIZE
}

// weight fragments of the 4-channel kernel: unit (cog*5 + ks)*2 + hl, lane (col, g): pair t = 4*ks + g, elements e < 4: channel e at
// the first tap of the pair, e >= 4: channel e - 4 at the second
__global__ void conv3_sb_pack4_kernel(const float* __restrict__ w, u32x4* __restrict__ wfrag, int Cin_f, int Cout_f, int mode, int ncog) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncog * SB4_KSTEPS * 64) return;
    const int lane = i & 63, ks = (i >> 6) % SB4_KSTEPS, cog = (i >> 6) / SB4_KSTEPS;
    const int col = lane & 15, g = lane >> 4, t = 4 * ks + g;
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int co = cog * 16 + col;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int tap = sb4_tap(t, e >> 2), ci = e & 3;
        float x = 0.f;
        if (tap >= 0 && ci < cin_conv && co < cout_conv)
            x = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
        v[e] = x;
    }
    u32x4 hi, lo;
    split8(v, hi, lo);
    const size_t unit = ((size_t)cog * SB4_KSTEPS + ks) * 2;
    wfrag[(unit + 0) * 64 + lane] = hi;
    wfrag[(unit + 1) * 64 + lane] = lo;
}
size_t conv3_sb4_frag_bytes(int Cout_conv) { return (size_t)cdiv(Cout_conv, 16) * SB4_KSTEPS * 2 * 64 * 16; }
int conv3_sb4_pack_weights(const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, hipStream_t s) {
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    RU_REQUIRE(cin_conv <= 4, "conv3_sb4: at most 4 input channels");
    const int ncog = cdiv(cout_conv, 16);
    hipLaunchKernelGGL(conv3_sb_pack4_kernel, dim3(cdiv(ncog * SB4_KSTEPS * 64, 256)), dim3(256), 0, s, w, (u32x4*)wfrag, Cin_f, Cout_f, mode, ncog);
    RU_CHECK_LAUNCH("conv3_sb_pack4_kernel");
    return RU_OK;
}

// ------------------------------------------------------------------ weight fragments
// unit u = ((cog*nchunk + chunk)*14 + ks)*2 + hl, 64 lanes x 16 bytes each: lane l (col = l&15, k-group g = l>>4) holds,
// for e = 0..7, W[cout = cog*16 + col][cin = chunk*16 + (g&1)*8 + e][tap = sb_tap(ks, g>>1)] (zeros for the phantom tap).
__device__ __forceinline__ void sb_pack_one(const float* __restrict__ w, u32x4* __restrict__ wfrag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog, int i) {
    const int total = ncog * nchunk * SB_KSTEPS * 64;
    if (i >= total) return;
    const int lane = i & 63;
    const int ks = (i >> 6) % SB_KSTEPS;
    const int chunk = ((i >> 6) / SB_KSTEPS) % nchunk;
    const int cog = (i >> 6) / (SB_KSTEPS * nchunk);
    const int col = lane & 15, g = lane >> 4;
    const int tap = sb_tap(ks, g >> 1);
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int co = cog * 16 + col;
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = chunk * 16 + (g & 1) * 8 + e;
        float v = 0.f;
        if (tap >= 0 && ci < cin_conv && co < cout_conv)
            v = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
        t[e] = v;
    }
    u32x4 hi, lo;
    split8(t, hi, lo);
    const size_t unit = ((size_t)(cog * nchunk + chunk) * SB_KSTEPS + ks) * 2;
    wfrag[(unit + 0) * 64 + lane] = hi;
    wfrag[(unit + 1) * 64 + lane] = lo;
}
// head-form unit u = f*2 + hl (f = 0..4), behind the direct fragments: lane l (col = l&15 = 4 dy + co, k-group g = l>>4) holds, for e = 0..7,
// W[cout = co][cin = (g&1)*8 + e][tap = sb_head_tap(f, g>>1, dy)] (conv3_sb_common.hpp, sb_head_shape)
__device__ __forceinline__ void sb_pack_head_one(const float* __restrict__ w, u32x4* __restrict__ hfrag, int Cin_f, int Cout_f, int mode, int i) {
    if (i >= SB_HEAD_KSTEPS * 64) return;
    const int lane = i & 63, f = i >> 6;
    const int col = lane & 15, g = lane >> 4, dy = col >> 2, co = col & 3;
    const int tap = sb_head_tap(f, g >> 1, dy);
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = (g & 1) * 8 + e;
        float v = 0.f;
        if (tap >= 0 && ci < cin_conv && co < cout_conv)
            v = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
        t[e] = v;
    }
    u32x4 hi, lo;
    split8(t, hi, lo);
    hfrag[((size_t)f * 2 + 0) * 64 + lane] = hi;
    hfrag[((size_t)f * 2 + 1) * 64 + lane] = lo;
}
// Every weight is packed in ALL forms where the channel counts allow the Winograd-z kernels (conv3_wz.hpp, conv3_wz32.hpp): the direct fragments, and right
// behind them (conv3_sb_frag_bytes_direct) the transformed ones of the 16x16x32 form, then (wz_frag_bytes further) those of the 32x32x16 form -- which kernel a launch takes depends on its SHAPE, and frozen packs (inference) must serve
// every shape.  Threads [0, direct) pack direct units, [direct, direct + wz) transformed ones, [direct + wz, direct + wz + wz32) the 32x32x16 ones.
// forms: bit 0 = the 16x16x32 Winograd-z fragments, bit 1 = the 32x32x16 ones (SB_FORMS_ALL: op-level packs and inference, whose frozen packs must serve
// every later launch; a training step repacks per forward and packs what that forward's launches take -- sb_pack_forms); bit 2 = the fp16 + MX-fp8 fragments of
// conv3_mx_kernel (16 input channels, whole 16-channel output blocks: shapes that have neither a Winograd-z nor a head form, so they sit right behind the direct ones)
constexpr int SB_FORMS_ALL = 15;                        // (bit 3: the same scheme's Winograd-z fragments, conv3_wz32mx.hpp, behind the two bf16 Winograd-z forms)
__device__ __forceinline__ void sb_pack_both(const float* __restrict__ w, u32x4* __restrict__ wfrag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog, int forms, int i) {
    const int direct = ncog * nchunk * SB_KSTEPS * 64;
    if (i < direct) { if (!(forms & 16)) sb_pack_one(w, wfrag, Cin_f, Cout_f, mode, nchunk, ncog, i); return; }      // (bit 4: the direct fragments are not needed -- conv3_sb_pack_add)
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    if (sb_head_shape(cin_conv, cout_conv)) { sb_pack_head_one(w, wfrag + (size_t)direct * 2, Cin_f, Cout_f, mode, i - direct); return; }
    if (mx_channels_ok(cin_conv, cout_conv)) { if (forms & 4) mx_pack_one(w, wfrag + (size_t)direct * 2, Cin_f, Cout_f, mode, ncog, i - direct, (forms & 32) != 0); return; }     // (bit 5: the gradient-operand variant in the same place)
    if (!wz_channels_ok(cin_conv, cout_conv)) return;
    const int wz = WZ16_FORM ? (cout_conv / 32) * nchunk * 4 * 2 * WZ_KSTEPS * 64 : 0;      // (the fragments keep their places whatever `forms` says: a skipped form leaves its bytes as they are)
    int r = i - direct;
    if (WZ16_FORM && (forms & 1)) {
        if (r < wz) { wz_pack_one(w, reinterpret_cast<wz_u32x4*>(wfrag + (size_t)direct * 2), Cin_f, Cout_f, mode, nchunk, cout_conv / 32, r); return; }
        r -= wz;
    }
    const int wz32 = (cout_conv / 32) * nchunk * 4 * 9 * 64;
    if (forms & 2) {
        if (r < wz32) { wz32_pack_one(w, reinterpret_cast<wz_u32x4*>(wfrag + ((size_t)direct + wz) * 2), Cin_f, Cout_f, mode, nchunk, cout_conv / 32, r); return; }
        r -= wz32;
    }
    if (forms & 8) wz32mx_pack_one(w, reinterpret_cast<wz_u32x4*>(wfrag + ((size_t)direct + wz + wz32) * 2), Cin_f, Cout_f, mode, nchunk, cout_conv / 32, r);
}
static inline int sb_pack_threads(int cin_conv, int cout_conv, int forms) {
    const int nchunk = cdiv(cin_conv, 16), ncog = cdiv(cout_conv, 16);
    return ncog * nchunk * SB_KSTEPS * 64 + (wz_channels_ok(cin_conv, cout_conv) ? (cout_conv / 32) * nchunk * (((WZ16_FORM && (forms & 1)) ? 4 * 2 * WZ_KSTEPS : 0) + ((forms & 2) ? 4 * 9 : 0) + ((forms & 8) ? WZ32MX_UNITS_XI : 0)) * 64 : 0)
         + (sb_head_shape(cin_conv, cout_conv) ? SB_HEAD_KSTEPS * 64 : 0) + (((forms & 4) && mx_channels_ok(cin_conv, cout_conv)) ? ncog * MX_UNITS * 64 : 0);
}
// what the launches of a TRAINING forward + backward read of a weight packed in `mode` (0: forward, 1: data gradient) under the RU_WZ / RU_WZ32 / RU_MX switches
// (read per call, as the launches read them; a toggle BETWEEN a forward's pack and a launch that reads it is not supported -- tools and tests toggle between steps)
static int sb_pack_forms(int mode) {
    if (mode == 1) return conv3_mxg_enabled() ? (4 | 32) : 0;   // data-gradient launches (split-form inputs) take the direct kernels; 16 -> 16: the gradient-operand MX form beside them
    const bool mxon = conv3_mx_enabled();                   // forward convolutions only: gradients never take the fp16 + MX-fp8 scheme
    const char* e = getenv("RU_WZ");
    if (e && *e == '0') return mxon ? 4 : 0;
    if (mxon && conv3_mx_wz_enabled()) return 4 | 8;        // forward launches take the MX kernel of their shape: direct (16 channels) or Winograd-z (32..)
    return (mxon ? 4 : 0) | (conv3_wz32_enabled() ? 2 : 1); // ... or ONE three-product Winograd-z form
}
__global__ void conv3_sb_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ wfrag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog, int forms) {
    sb_pack_both(w, wfrag, Cin_f, Cout_f, mode, nchunk, ncog, forms, blockIdx.x * blockDim.x + threadIdx.x);
}
// all 3x3x3 weights of a network in ONE launch (blockIdx.y = entry): the ~50 pack launches of a training step were 4.5 us each,
// almost all of it launch latency (5 % of a batch-1 forward)
__global__ void conv3_sb_pack_batch_kernel(const SbPackBatch b) {
    const SbPackEntry& e = b.e[blockIdx.y];
    sb_pack_both(e.w, reinterpret_cast<u32x4*>(e.wfrag), e.Cin_f, e.Cout_f, e.mode, e.nchunk, e.ncog, e.forms, blockIdx.x * blockDim.x + threadIdx.x);
}
int conv3_sb_pack_batch(SbPackBatch& b, hipStream_t s) {
    if (b.n == 0) return RU_OK;
    int maxtotal = 0;
    for (int i = 0; i < b.n; ++i) {
        const SbPackEntry& e = b.e[i];
        const int t = sb_pack_threads(e.mode == 0 ? e.Cin_f : e.Cout_f, e.mode == 0 ? e.Cout_f : e.Cin_f, e.forms);
        if (t > maxtotal) maxtotal = t;
    }
    hipLaunchKernelGGL(conv3_sb_pack_batch_kernel, dim3(cdiv(maxtotal, 256), b.n), dim3(256), 0, s, b);
    RU_CHECK_LAUNCH("conv3_sb_pack_batch_kernel");
    b.n = 0;
    return RU_OK;
}
int conv3_sb_switch_signature() {
    const char* e = getenv("RU_WZ");
    return ((e && *e == '0') ? 0 : 1) | (conv3_wz32_enabled() ? 2 : 0) | (conv3_mx_enabled() ? 4 : 0) | (conv3_mx_wz_enabled() ? 8 : 0) | (conv3_mxg_enabled() ? 16 : 0);
}
bool conv3_sb_forward_skips_direct(int N, int Cin, int Cout, int D, int H, int W) {
    if (conv3_sb_uses_wz(N, Cin, Cout, D, H, W, 2)) return true;                    // conv3_wz32mx_kernel or conv3_wz32_kernel
    return conv3_mx_enabled() && conv3_mx_shape_ok(N, Cin, Cout, D, H, W);          // conv3_mx_kernel
}
int conv3_sb_pack_add(SbPackBatch& b, const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, bool all_forms, hipStream_t s, bool skip_direct) {
    if (b.n == RU_PACK_BATCH) { const int rc = conv3_sb_pack_batch(b, s); if (rc) return rc; }
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    SbPackEntry& e = b.e[b.n++];
    e.w = w; e.wfrag = wfrag; e.Cin_f = Cin_f; e.Cout_f = Cout_f; e.mode = mode; e.nchunk = cdiv(cin_conv, 16); e.ncog = cdiv(cout_conv, 16);
    e.forms = all_forms ? SB_FORMS_ALL : (sb_pack_forms(mode) | ((skip_direct && mode == 0) ? 16 : 0));
    return RU_OK;
}

size_t conv3_sb_frag_bytes_direct(int Cin_conv, int Cout_conv) {
    return (size_t)cdiv(Cout_conv, 16) * cdiv(Cin_conv, 16) * SB_KSTEPS * 2 * 64 * 16;
}
size_t conv3_sb_frag_bytes(int Cin_conv, int Cout_conv) {          // direct fragments + (32..: the Winograd-z fragments | <= 4 couts: the head form) behind them
    return conv3_sb_frag_bytes_direct(Cin_conv, Cout_conv) + wz_frag_bytes(Cin_conv, Cout_conv) + wz32_frag_bytes(Cin_conv, Cout_conv) + wz32mx_frag_bytes(Cin_conv, Cout_conv)
         + (sb_head_shape(Cin_conv, Cout_conv) ? (size_t)SB_HEAD_KSTEPS * 2 * 64 * 16 : 0) + mx_frag_bytes(Cin_conv, Cout_conv);
}
// The inference head: the last Residual block's output x + lrelu(norm2(conv2)) (model.py:112-116) is formed in the head conv's staging instead of a pass of its
// own (Conv3Args::in_res).  Exists in the head-form variant of the persistent kernel only: the same tests as conv3_sb_launch's way there.  RU_HEAD_RES=0: off (A/B).
bool conv3_sb_head_takes_residual(int N, int Cin, int Cout, int D, int H, int W) {
    const char* e = getenv("RU_HEAD_RES");
    if (e && *e == '0') return false;
    if (!conv3_sb_head_form_enabled() || !sb_head_shape(Cin, Cout) || Cin % 16 != 0 || (W & 3) != 0) return false;
    if ((size_t)D * H * W * 64 >= ((size_t)1 << 31)) return false;
    return sb_use_v2(sb_choose(N, Cout, D, H, W));
}
bool conv3_sb_head_form_enabled() {
    const char* e = getenv("RU_HEAD_FORM");
    return !(e && e[0] == '0');
}

int conv3_sb_pack_weights(const float* w, void* wfrag, int Cin_f, int Cout_f, int mode, hipStream_t s, bool grad_operand) {
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int nchunk = cdiv(cin_conv, 16), ncog = cdiv(cout_conv, 16);
    const int forms = SB_FORMS_ALL | (grad_operand ? 32 : 0);    // (the MX fragments of a weight exist in ONE variant: activation or gradient operand)
    const int total = sb_pack_threads(cin_conv, cout_conv, forms);
    hipLaunchKernelGGL(conv3_sb_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, (u32x4*)wfrag, Cin_f, Cout_f, mode, nchunk, ncog, forms);
    RU_CHECK_LAUNCH("conv3_sb_pack_kernel");
    return RU_OK;
}

// number of statistics partials per (sample, channel) the kernel chosen for this shape writes
// The Winograd-z kernels (conv3_wz32.hpp, conv3_wz32mx.hpp) take the voxel-major FORWARD convolutions of 32 and more channels whose shape fills the chip;
// RU_WZ=0 keeps every shape on the direct kernels (same-box A/B).  ONE rule for the launch and for the partial count the engine sizes.  Split-form inputs
// (the data-gradient convolutions: gn_bwd_apply16's hi / lo packets) stay on the direct kernel: there the staging is a global -> LDS DMA copy with no VALU
// work at all, while the z transform has to re-join, transform and re-split every value -- measured 120 / 143 / 104 us against 93 / 107 / 79 in round 5.
bool conv3_sb_uses_wz(int N, int Cin, int Cout, int D, int H, int W, int products) {
    const char* e = getenv("RU_WZ");                    // read per call: tests and tools switch it inside one process
    const bool off = e && *e == '0';
    return !off && products != 1 && conv3_wz_shape_ok(N, Cin, Cout, D, H, W);
}
int conv3_sb_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W, int products) {
    if (conv3_sb_uses_wz(N, Cin, Cout, D, H, W, products)) return (int)wz_grid_x(N, Cout, D, H, W);
    const SBChoice c = sb_choose(N, Cout, D, H, W);
    if (sb_use_v2(c)) return (int)sb2_grid_x(N, Cout, D, H, W);        // persistent kernel: one per workgroup
    return cdiv(D, c.tz) * cdiv(H, c.ty) * cdiv(W, 16);                 // one-stage kernel: one per tile
}

bool conv3_sb_bst_usable(int N, int Cout, int D, int H, int W) { return sb_use_v2(sb_choose(N, Cout, D, H, W)); }

template <int TZ, int TY, bool IN16, bool OUT16>
static int sb_cfg(const Conv3Args& a, hipStream_t s) {
    using P = SB<TZ, TY>;
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_sb_kernel<TZ, TY, IN16, OUT16>), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS_BYTES + SB1_W_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_sb)");
        attr_done.set();
    }
    const int ntz = cdiv(a.D, TZ), nty = cdiv(a.H, TY), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)((long)a.N * ntz * nty * ntx), (unsigned)cdiv(a.Cout, 16));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.kind == 1 && a.fin.nblk == ntz * nty * ntx && a.fin.N == a.N && a.fin.C == a.Cout),
               "conv3_sb: tail descriptor does not match the launch");
    hipLaunchKernelGGL((conv3_sb_kernel<TZ, TY, IN16, OUT16>), grid, dim3(256), P::LDS_BYTES + SB1_W_BYTES, s, a, (const u32x4*)a.wfrag, ntz, nty, ntx, cdiv(a.Cin, 16));
    RU_CHECK_LAUNCH("conv3_sb_kernel");
    return RU_OK;
}

bool conv3_sb4_usable(int N, int Cin, int Cout, int D, int H, int W) {
    return Cin <= 4 && sb_use_v2(sb_choose(N, Cout, D, H, W));
}

template <bool OUT16, bool BST = false>
static int sb2c4_cfg(const Conv3Args& a, hipStream_t s) {
    using P = SB<4, 8>;
    static PerDevice attr_done;
    constexpr int LDS = 2 * 2 * P::HVOLP * 16 + SB_STAT_LDS_FLOATS * 4;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_sb2c4_kernel<OUT16, BST>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_sb2c4)");
        attr_done.set();
    }
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_sb2c4: at most 32 samples per call when statistics are requested");
    dim3 grid((unsigned)sb2_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)cdiv(a.Cout, 16));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)LDS),
               "conv3_sb2c4: tail descriptor does not match the launch");
    hipLaunchKernelGGL((conv3_sb2c4_kernel<OUT16, BST>), grid, dim3(512), LDS, s, a, (const u32x4*)a.wfrag, cdiv(a.D, 4), cdiv(a.H, 8), cdiv(a.W, 16));
    RU_CHECK_LAUNCH("conv3_sb2c4_kernel");
    return RU_OK;
}

int conv3_sb_launch(const Conv3Args& a, hipStream_t s) {
    RU_REQUIRE(!(a.sigmoid && a.out_c16), "conv3_sb: the fused sigmoid exists for NCDHW output only");
    RU_REQUIRE(!a.bst_y || (a.bst_k && a.stat_partials && (a.in_c16 || a.in_c4) && a.out_c16 && !a.bias && !a.sigmoid && (!a.add || !a.in_c4) &&
                            conv3_sb_bst_usable(a.N, a.Cout, a.D, a.H, a.W)),
               "conv3_sb: fused GroupNorm-backward statistics need the persistent voxel-major kernel, a partial buffer and no bias / activation");
    RU_REQUIRE(!a.in_res || (a.in_c16 && !a.out_c16 && !a.in_c4 && !a.add && a.in_scale && conv3_sb_head_takes_residual(a.N, a.Cin, a.Cout, a.D, a.H, a.W)),
               "conv3_sb: a residual of the input is staged by the head-form kernel only (conv3_sb_head_takes_residual)");
    if (a.in_c4) {
        RU_REQUIRE(a.Cin <= 4 && !a.in_scale, "conv3_sb: the 4-channel kernel takes Cin <= 4 and no fused input transform");
        RU_REQUIRE(!a.out_c16 || a.Cout % 16 == 0, "conv3_sb: C16 output needs Cout %% 16 == 0");
        const SBChoice c4 = sb_choose(a.N, a.Cout, a.D, a.H, a.W);
        RU_REQUIRE(sb_use_v2(c4), "conv3_sb: the 4-channel kernel needs at least 256 (4,8,16) tiles x cout groups");
        if (a.bst_y) return sb2c4_cfg<true, true>(a, s);
        return a.out_c16 ? sb2c4_cfg<true>(a, s) : sb2c4_cfg<false>(a, s);
    }
    RU_REQUIRE((a.W & 3) == 0 || (a.in_c16 && a.out_c16), "conv3_sb: W must be a multiple of 4 for NCDHW tensors");
    RU_REQUIRE(!a.in_c16 || a.Cin % 16 == 0, "conv3_sb: C16 input needs Cin %% 16 == 0");
    RU_REQUIRE(!a.in_s16 || (a.in_c16 && !a.in_scale), "conv3_sb: a split-form input is voxel-major and has no fused transform");
    RU_REQUIRE(!a.out_c16 || a.Cout % 16 == 0, "conv3_sb: C16 output needs Cout %% 16 == 0");
    SBChoice c = sb_choose(a.N, a.Cout, a.D, a.H, a.W);
    RU_REQUIRE(!(a.bias && a.out_c16 && a.stat_partials), "conv3_sb: bias + voxel-major output + statistics is not a path of the network");
    if (sb_use_v2(c) && a.bias && a.out_c16) c = SBChoice{2, 8};
#ifdef RU_SB2_DBG
    if ((RU_SB2_DBG & 2048) && !a.stat_partials) c = SBChoice{2, 8};      // tools: time the one-stage kernel on a shape the persistent kernel would take
    if ((RU_SB2_DBG & 4096) && !a.stat_partials) c = SBChoice{2, 4};
#endif      // (no engine path: the persistent kernel has the bias for NCDHW output only)
    if (a.in_c16 && a.out_c16 && !a.in_s16 && !a.bias && !a.sigmoid && !a.bst_y && !a.add && conv3_sb_uses_wz(a.N, a.Cin, a.Cout, a.D, a.H, a.W, a.products))
        return conv3_wz_launch(a, static_cast<const char*>(a.wfrag) + conv3_sb_frag_bytes_direct(a.Cin, a.Cout), s);
    // Conv3Args::in_g16: the input is a gradient in the operand form of the MX scheme (the caller asked conv3_mxg_usable before it wrote the tensor that way)
    if (a.in_g16) {
        RU_REQUIRE(a.in_s16 && a.out_c16 && !a.bias && !a.sigmoid && !a.in_res && conv3_mxg_usable(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
                   "conv3_sb: a gradient-operand input is taken by conv3_mx_kernel<GRAD> only (16 -> 16 channels, persistent-kernel shapes)");
        return conv3_mx_launch(a, static_cast<const char*>(a.wfrag) + conv3_sb_frag_bytes_direct(a.Cin, a.Cout), s);
    }
    // Conv3Args::products == 2: the caller's input is an ACTIVATION tensor and it asks for the fp16 + MX-fp8 product scheme where a kernel for the shape exists
    // (conv3_mx.hpp: the 16-channel level); everywhere else the request means three products
    if (a.products == 2 && a.in_c16 && a.out_c16 && !a.in_s16 && !a.bias && !a.sigmoid && !a.add && !a.bst_y && !a.in_res && conv3_mx_enabled() &&
        conv3_mx_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W))
        return conv3_mx_launch(a, static_cast<const char*>(a.wfrag) + conv3_sb_frag_bytes_direct(a.Cin, a.Cout), s);
    if (sb_use_v2(c)) {
        if (a.in_c16 && a.out_c16) return a.products == 1 ? conv3_sb2_launch_c16_p1(a, s) : conv3_sb2_launch_c16(a, s);
        return conv3_sb2_launch_mixed(a, s);             // (three products whatever a.products says: the NCDHW-side variants have no one-product form)
    }
    if (c.ty == 8) {
        if (a.in_c16) return a.out_c16 ? sb_cfg<2, 8, true, true>(a, s) : sb_cfg<2, 8, true, false>(a, s);
        return a.out_c16 ? sb_cfg<2, 8, false, true>(a, s) : sb_cfg<2, 8, false, false>(a, s);
    }
    if (c.ty == 2) {
        if (a.in_c16) return a.out_c16 ? sb_cfg<2, 2, true, true>(a, s) : sb_cfg<2, 2, true, false>(a, s);
        return a.out_c16 ? sb_cfg<2, 2, false, true>(a, s) : sb_cfg<2, 2, false, false>(a, s);
    }
    if (a.in_c16) return a.out_c16 ? sb_cfg<2, 4, true, true>(a, s) : sb_cfg<2, 4, true, false>(a, s);
    return a.out_c16 ? sb_cfg<2, 4, false, true>(a, s) : sb_cfg<2, 4, false, false>(a, s);
}

}  // namespace ru
