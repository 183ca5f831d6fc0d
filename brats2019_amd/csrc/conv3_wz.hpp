// conv3_wz.hpp -- the reduced-product form of the split-bf16 3x3x3 convolution for the 32..128-channel levels (round 5):
// Winograd F(2,3) along z, the (y, x) taps direct.
//
//   reference arithmetic: model.py:72-73 (nn.Conv3d 3x3x3, stride 1, padding 1, no bias) as used by model.py:89-91 (Residual.conv1 / conv2)
//   and its data gradient (autograd of the same op: a 3x3x3 convolution with mirrored taps and exchanged channel roles).
//
// Two output planes z0, z0+1 from the four input planes d0..d3 = z0-1 .. z0+2 (per voxel position and channel, after the fused input
// transform):   U0 = d0 - d2,  U1 = d1 + d2,  U2 = d2 - d1,  U3 = d1 - d3          (input transform, fp32, then split hi / lo)
//               G0 = g0,  G1 = (g0 + g1 + g2) / 2,  G2 = (g0 - g1 + g2) / 2,  G3 = g2   (g_dz = the 3x3 (dy, dx) slice of the weight; packed once)
//               M_xi = conv2d_(y,x)(U_xi, G_xi)   -- four 9-tap convolutions instead of two 27-tap ones: 36 instead of 54 tap products
//               out(z0) = M0 + M1 + M2,   out(z0+1) = M1 - M2 - M3
// Numerics (tools/winograd_gate.py, profiles/r05_winograd_gate.txt): 1.2x the error of the direct split-bf16 convolution (2.7e-5 against
// 2.4e-5 of the output RMS at worst), the split applied AFTER the fp32 transforms.
//
// Same persistent producer / consumer skeleton as conv3_sb2_kernel (conv3_sb_common.hpp), re-cut around what the transform changes:
//   * tile (2 z, 8 y, 16 x) and 32 output channels per workgroup pass.  Consumer wave xi owns TRANSFORMED plane xi: its weights are 5 K-steps
//     (9 taps) x 2 cout groups = 80 VGPRs (the direct kernel: 14 K-steps, 112 VGPRs for ONE group), its accumulators M_xi for both groups 64 --
//     so one staged image feeds 32 output channels: per output channel half the staging work and half the A-fragment reads of the direct
//     kernel (20 fragment pairs per item for 240 MFMAs; there 50 for 336).
//   * K-steps pair taps as there: K-steps 0-2 = (dy, dx 0 | dx 1), read once per halo row and used by the three output rows that share it;
//     K-step 3 = (dy 0 | dy 1) of dx 2, K-step 4 = (dy 2 | phantom) of dx 2 -- both served by ONE fragment (rows r | r+1 at dx 2).
//   * the output transform needs all four waves' M: a wave leaves its accumulators in an LDS scratch (64 KB) at the end of a tile's last chunk;
//     behind the item barrier wave w combines plane (w & 1) of cout group (w >> 1), row by row, and runs the row epilogue (statistics /
//     residual / GroupNorm-backward sums / store) of the direct kernel.
//   * staging: a lane owns (halo position, 4 channels) and loads the FOUR planes of it (one float4 each), applies the fused affine + LeakyReLU
//     (or re-joins the hi / lo halves of a split-form input), transforms along z in registers, splits, and writes 8 bytes per plane and half
//     (ds_write_b64).  12 wave-rounds cover 6 position blocks x 2 channel halves exactly (94 % of the lanes busy).
#pragma once
#include "conv3_sb_common.hpp"
#include "conv3_wz_pack.hpp"
#include "conv3_mx_pack.hpp"

namespace ru {

// devtools bit 128: consumer wave 0 of every workgroup adds s_memtime section sums here (cycles): [0] item setup, [1] rows 0-4, [2] rows 5-9 + row 7's
// scratch write, [3] barrier, [4] items, [5] tail after the loop, [6] workgroups, [7] staging wave 0: cycles from item barrier to item barrier spent in store + issue
static __device__ unsigned long long wz_prof[8];
// The staging waves (waves 4..7 of a workgroup) of the Winograd-z kernels (conv3_wz32_kernel, conv3_wz32mx_kernel; conv3_wz_kernel here in devtools builds): the
// transformed, split image of item w+1 is written while the matrix waves work on item w; one __syncthreads per item, two closing ones.  Plain float32 or
// fused-transform input (the forward convolutions: the split-form / data-gradient routes of round 5 are retired).
// MX (conv3_wz32mx_kernel, round 6): the image is written in the operand formats of the fp16 + MX-fp8 product scheme (conv3_mx.hpp) -- sections 0 / 1 fp16 halves
// (the bf16 hi sections' layout), section 2 e4m3(lo * 2^11), section 3 e4m3(value), 16 channels of a position per 16-byte packet -- plain or fused-transform input only.
template <int dbg, bool MX = false>
// zcol_ntz > 0: the COLUMN walk -- workgroup swz owns the tiles q in [swz * nsteps, (swz + 1) * nsteps) of the z-fastest order q = ((n * nty + ty) * ntx + tx) * ntz + tz, so
// consecutive tiles of a workgroup are z neighbours (the two input planes they share were loaded by the same CU nchunk items earlier: L2 hits) and the workgroups of an XCD
// cover adjacent columns; 0: tile swz + step * G of the x-fastest order (a step's tiles form z slabs, the z neighbour belongs to another XCD: both fetch the planes).
__device__ __forceinline__ void wz_stage_waves(const Conv3Args& a, u32x4* lds, int rw, int lane, int swz, int G, int nitems, int nchunk,
                                               int tiles_per_sample, int nty, int ntx, int zcol_ntz = 0) {
    if constexpr (MX) mx_set_saturating_conversions();
    constexpr int HX = WZ_HX, HVOLP = WZ_HVOLP, BUF = WZ_BUF;
    const int D = a.D, H = a.H, W = a.W;
    const size_t DHW = (size_t)D * H * W;
    const int nsteps_wg = nitems / nchunk;
    auto tile_origin = [&](int step, int& n, int& z0, int& y0, int& x0) {
        int tz, ty, tx;
        if (zcol_ntz > 0) {
            int b = swz * nsteps_wg + step;
            tz = b % zcol_ntz; b /= zcol_ntz;
            tx = b % ntx; b /= ntx;
            ty = b % nty; n = b / nty;
        } else {
            int b = swz + step * G;
            n = b / tiles_per_sample;
            b -= n * tiles_per_sample;
            tx = b % ntx; b /= ntx;
            ty = b % nty;
            tz = b / nty;
        }
        z0 = tz * 2; y0 = ty * 8; x0 = tx * 16;
    };
    // ---------------------------------------------------------------- producers
    const bool xform = a.in_scale != nullptr;
    const float slope = xform ? a.in_slope : 1.f;
    // lane = (position within a block of 16, channel quad): the four lanes of a position load its 64 bytes as ONE contiguous line piece
    // (a first version gave a lane pair half a voxel and two wave-rounds per block: every load touched 32 lines and used half of each,
    // and the staging ran at the speed of the L1 / address pipeline -- 7 900 cycles per item against the consumers' 6 500, tools/wz_sections.py)
    const int quad = lane & 3, q0 = quad & 1, hf = quad >> 1;
    int dlt[3], pyx[3], ppos[3];
    bool pin[3];
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
        const int pb = rd * 4 + rw;                                  // 12 wave-rounds = 12 blocks of 16 halo positions (180 of 192 slots used)
        const int p = pb * 16 + (lane >> 2);
        const int hy = p / HX, xc = p - hy * HX;
        pin[rd] = p < WZ_PLANE;
        ppos[rd] = p;
        pyx[rd] = hy | (xc << 8);
        dlt[rd] = (hy * W + xc) * 64 + quad * 16;
    }
    // TWO register sets of loads in flight: the loads of item w+3 are issued when item w+1 has been converted, and consumed two item
    // barriers later.  With one set (issue(w+2) right before the barrier, store(w+2) right behind it) the staging waves -- the pole of this
    // kernel: ~400 VALU instructions per item beside a wave that issues 240 MFMAs -- sat out a full L2 / HBM round trip every item.
    float4 vv[2][3][4];
    float4 sc4s[2] = {}, sh4s[2] = {};                                      // the item's scale / shift of this lane's channel quad (one copy per register set)
    unsigned okmasks[2] = {0u, 0u};                                  // bit rd: (y, x) of this lane's position is inside the volume
    bool zok0s[2] = {true, true}, zok3s[2] = {true, true};           // planes z0-1 / z0+2 inside the volume (wave-uniform)
    auto issue = [&](auto SET, int item) {
        constexpr int set = decltype(SET)::value;
        auto& v = vv[set];
        auto& sc4 = sc4s[set];
        auto& sh4 = sh4s[set];
        unsigned& okmask = okmasks[set];
        bool& zok0 = zok0s[set];
        bool& zok3 = zok3s[set];
        if constexpr ((dbg & 2) != 0) return;
        const int step = item / nchunk, chunk = item - step * nchunk;
        int n, z0, y0, x0;
        tile_origin(step, n, z0, y0, x0);
        const float* xb = a.x + ((size_t)(n * nchunk + chunk) * DHW) * 16;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(DHW * 64), 0x00020000);
        const int zm1 = z0 - 1, ym1 = y0 - 1, xm1 = x0 - 1;
        const int base = ((zm1 * H + ym1) * W + xm1) * 64;
        const int pstride = H * W * 64;
        zok0 = zm1 >= 0; zok3 = z0 + 2 < D;
        okmask = 0;
        static_for<3>([&](auto R) __attribute__((always_inline)) {
            constexpr int rd = decltype(R)::value;
            const int gy = ym1 + (pyx[rd] & 0xff), gx = xm1 + (pyx[rd] >> 8);
            const bool ok = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W) & pin[rd];
            okmask |= ok ? (1u << rd) : 0u;
            static_for<4>([&](auto PZ) __attribute__((always_inline)) {
                constexpr int pz = decltype(PZ)::value;
                const bool okz = ok & ((unsigned)(zm1 + pz) < (unsigned)D);
                const unsigned ofs = okz ? (unsigned)(base + dlt[rd] + pz * pstride) : 0x80000000u;
                v[rd][pz] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ofs, 0, 0));
            });
        });
        if (xform) {
            const int cofs = n * a.Cin + chunk * 16 + quad * 4;
            sc4 = *reinterpret_cast<const float4*>(a.in_scale + cofs);
            sh4 = *reinterpret_cast<const float4*>(a.in_shift + cofs);
        }
    };
    auto store = [&](auto SET, u32x4* buf) {
        constexpr int set = decltype(SET)::value;
        auto& v = vv[set];
        auto& sc4 = sc4s[set];
        auto& sh4 = sh4s[set];
        const unsigned okmask = okmasks[set];
        const bool zok0 = zok0s[set], zok3 = zok3s[set];
        if constexpr ((dbg & 1) != 0) {
            if constexpr ((dbg & 2) == 0) {
                float acc0 = 0.f;
#pragma unroll
                for (int rd = 0; rd < 3; ++rd)
#pragma unroll
                    for (int pz = 0; pz < 4; ++pz) acc0 += v[rd][pz].x;
                if (acc0 == 12345.678f) buf[0] = u32x4{1u, 2u, 3u, 4u};
            }
            return;
        }
        uint2* b2 = reinterpret_cast<uint2*>(buf);
        const float mz0 = zok0 ? 1.f : 0.f, mz3 = zok3 ? 1.f : 0.f;
        const float s[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, t[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        const float t0[4] = {t[0] * mz0, t[1] * mz0, t[2] * mz0, t[3] * mz0}, t3[4] = {t[0] * mz3, t[1] * mz3, t[2] * mz3, t[3] * mz3};
        auto body = [&](auto MODE) __attribute__((always_inline)) {
            constexpr int mode = decltype(MODE)::value;                      // 0 plain fp32, 1 fused affine + LeakyReLU
#pragma unroll
            for (int rd = 0; rd < 3; ++rd) {
                if (!pin[rd]) continue;
                const int o = (hf * HVOLP + ppos[rd]) * 2 + q0;             // uint2 index of transformed plane 0, hi; lo: + 2*HVOLP*2
                float d[4][4];
                if constexpr (mode == 1) {
                    if (!((okmask >> rd) & 1u)) {                            // outside the volume in (y, x): the ACTIVATED tensor is zero-padded
                        const uint2 z = make_uint2(0u, 0u);
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi) {
                            b2[o + xi * WZ_PLANE * 2] = z;
                            if constexpr (MX) {
                                unsigned* b1 = reinterpret_cast<unsigned*>(buf);
                                b1[(2 * HVOLP + xi * WZ_PLANE + ppos[rd]) * 4 + quad] = 0u;
                                b1[(3 * HVOLP + xi * WZ_PLANE + ppos[rd]) * 4 + quad] = 0u;
                            } else {
                                b2[o + (2 * HVOLP + xi * WZ_PLANE) * 2] = z;
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int pz = 0; pz < 4; ++pz) {
                        const float f[4] = {v[rd][pz].x, v[rd][pz].y, v[rd][pz].z, v[rd][pz].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            // planes 0 / 3 outside the volume (wave-uniform): their loads returned zeros and the shift is masked, so the activated value is zero
                            const float u = fmaf(f[c], s[c], pz == 0 ? t0[c] : (pz == 3 ? t3[c] : t[c]));
                            d[pz][c] = fmaxf(u, u * slope);
                        }
                    }
                } else {
#pragma unroll
                    for (int pz = 0; pz < 4; ++pz) { d[pz][0] = v[rd][pz].x; d[pz][1] = v[rd][pz].y; d[pz][2] = v[rd][pz].z; d[pz][3] = v[rd][pz].w; }
                }
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    float u[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        u[c] = xi == 0 ? d[0][c] - d[2][c] : (xi == 1 ? d[1][c] + d[2][c] : (xi == 2 ? d[2][c] - d[1][c] : d[1][c] - d[3][c]));
                    if constexpr (MX) {
                        uint2 h16;
                        unsigned l8, x8;
                        mx_split4(u, h16, l8, x8);
                        b2[o + xi * WZ_PLANE * 2] = h16;
                        unsigned* b1 = reinterpret_cast<unsigned*>(buf);                 // sections 2 / 3: dword `quad` of the position's packet
                        b1[(2 * HVOLP + xi * WZ_PLANE + ppos[rd]) * 4 + quad] = l8;
                        b1[(3 * HVOLP + xi * WZ_PLANE + ppos[rd]) * 4 + quad] = x8;
                    } else {
                    uint2 hi, lo;
                    split_pair(u[0], u[1], hi.x, lo.x);
                    split_pair(u[2], u[3], hi.y, lo.y);
                    b2[o + xi * WZ_PLANE * 2] = hi;
                    b2[o + (2 * HVOLP + xi * WZ_PLANE) * 2] = lo;
                    }
                }
            }
        };
        if (xform) body(std::integral_constant<int, 1>{});
        else body(std::integral_constant<int, 0>{});
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    if (nitems > 0) issue(S0, 0);
    if (nitems > 1) issue(S1, 1);
    if (nitems > 0) {
        store(S0, lds);
        if (nitems > 2) issue(S0, 2);
    }
    __syncthreads();
    unsigned long long ppt = 0;
    for (int w = 0; w < nitems; w += 2) {            // item w+1 lives in set 1, item w+2 in set 0
        unsigned long long t0 = 0;
        if constexpr ((dbg & 128) != 0) t0 = __builtin_readcyclecounter();
        if (w + 1 < nitems) {
            store(S1, lds + BUF);
            if (w + 3 < nitems) issue(S1, w + 3);
        }
        if constexpr ((dbg & 128) != 0) ppt += __builtin_readcyclecounter() - t0;
        __syncthreads();
        if (w + 1 >= nitems) break;
        if constexpr ((dbg & 128) != 0) t0 = __builtin_readcyclecounter();
        if (w + 2 < nitems) {
            store(S0, lds);
            if (w + 4 < nitems) issue(S0, w + 4);
        }
        if constexpr ((dbg & 128) != 0) ppt += __builtin_readcyclecounter() - t0;
        __syncthreads();
    }
    if constexpr ((dbg & 128) != 0) { if (rw == 0 && lane == 0) atomicAdd(&wz_prof[7], ppt); }
    __syncthreads();                                 // closing barriers of the consumers: the last tile's combine, the last statistics flush
    __syncthreads();
}

template <bool BST, bool ADD>
__global__ __launch_bounds__(512, 2) void conv3_wz_kernel(const Conv3Args a, const u32x4* __restrict__ wfrag, int ntz, int nty, int ntx, int nchunk) {
    constexpr int HY = WZ_HY, HX = WZ_HX, HVOLP = WZ_HVOLP, BUF = WZ_BUF, MT = 8;
    // devtools builds only (python -m brats2019_amd.build --dbg <bits>; results are wrong): 1 = the staging waves skip transform / split / LDS stores,
    // 2 = they skip the global loads, 4 = the matrix waves skip their MFMAs and fragment reads, 8 = they skip the combine (scratch + row epilogue)
#ifdef RU_SB2_DBG
    constexpr int dbg = RU_SB2_DBG;
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    float* scratch = smem + 2 * BUF * 4;
    float* stat_lds = scratch + WZ_SCRATCH_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3;
    const int cog32 = blockIdx.y;
    const int H = a.H;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;      // XCD-compact tile order (conv3_sb2_kernel)
    const int nsteps = swz < ntile ? (ntile - swz + G - 1) / G : 0;
    const int nitems = nsteps * nchunk;

    if (producer) {
        wz_stage_waves<dbg>(a, lds, rw, lane, swz, G, nitems, nchunk, tiles_per_sample, nty, ntx);
    } else {
        // ---------------------------------------------------------------- consumers
        const int xi = rw;
        const int kg = lane >> 4, slot = kg >> 1, half = kg & 1;
        const int fbA = half * HVOLP + (xi * HY) * HX + slot + (lane & 15);              // K-steps 0-2: halo row r, (dx 0 | dx 1)
        const int fbB = half * HVOLP + (xi * HY + slot) * HX + 2 + (lane & 15);          // K-steps 3 / 4: rows (r | r+1) at dx 2
        const int fbB2 = half * HVOLP + (xi * HY) * HX + 2 + (lane & 15);                // rows 8, 9: both slots row r (slot 1 meets zero weights)
        u32x4 wreg[2][WZ_KSTEPS][2];
        // weight fragments through a buffer descriptor: lane offset in a VGPR (constant), the unit's offset scalar -- no 64-bit VALU address
        // arithmetic in the matrix wave's stream (as plain pointer loads the 20 refill loads of an item cost ~27 cycles each there)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wfrag), 0, (int)((size_t)gridDim.y * nchunk * WZ_UNITS * 1024), 0x00020000);
        const unsigned wlane = (unsigned)lane * 16u;
        auto wbase = [&](int chunk) { return (unsigned)(((cog32 * nchunk + chunk) * 4 + xi) * (2 * WZ_KSTEPS * 2)) * 1024u; };      // scalar byte offset of this wave's 20 units
        auto wload = [&](unsigned base, int g, int ks, int hl) __attribute__((always_inline)) {
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, base + (unsigned)(((g * WZ_KSTEPS + ks) * 2 + hl) * 1024), 0));
        };
        {
            const unsigned wb0 = wbase(0);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int ks = 0; ks < WZ_KSTEPS; ++ks) {
                    wreg[g][ks][0] = wload(wb0, g, ks, 0);
                    wreg[g][ks][1] = wload(wb0, g, ks, 1);
                }
        }
        f32x4 acc[2][MT];
        auto mm = [](const bf16x8& av, const bf16x8& wv, const f32x4& c) -> f32x4 {      // operands swapped: D[m = cout][n = voxel]
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, c, 0, 0, 0);
        };
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        // ---- the combining role of this wave: output plane pz of cout group og
        const int pz = rw & 1, og = rw >> 1;
        const int cog16 = cog32 * 2 + og;
        f32x4 kc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        int kc_n = -1;
        auto need_kc = [&](int n) __attribute__((always_inline)) {
            if constexpr (BST) {
                if (n != kc_n) {
                    const float* kp = a.bst_k + (size_t)n * 3 * a.Cout + cog16 * 16 + 4 * (lane >> 4);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const float4 q = *reinterpret_cast<const float4*>(kp + (size_t)t * a.Cout);
                        kc[t] = f32x4{q.x, q.y, q.z, q.w};
                    }
                    kc_n = n;
                }
            }
        };
        const int stat_blk = blockIdx.x, stat_nblk = G;
        unsigned flushed = 0;
        int n_acc = -1, pend_n = -1, pend_par = 0, par = 0;
        auto flush_stats = [&](int n) {
            if (a.stat_partials) sb_stats_to_lds<true>(s1, s2, stat_lds + (par * 4 + rw) * 32, lane);
            pend_n = n; pend_par = par; par ^= 1;
            flushed |= 1u << (n & 31);
            s1 = f32x4{0.f, 0.f, 0.f, 0.f}; s2 = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // lanes 0..31 of wave 0: lane>>4 = cout group, its two waves' rows (2g, 2g+1) of one generation summed in wave order
        auto commit_one = [&](const float* sc4w, int n) {
            if (rw == 0 && lane < 32) {
                const int gs = lane >> 4, c = lane & 15;
                float u1 = 0.f, u2 = 0.f;
                if (sc4w) {
                    u1 = sc4w[(2 * gs) * 32 + c * 2] + sc4w[(2 * gs + 1) * 32 + c * 2];
                    u2 = sc4w[(2 * gs) * 32 + c * 2 + 1] + sc4w[(2 * gs + 1) * 32 + c * 2 + 1];
                }
                const int co = cog32 * 32 + gs * 16 + c;
                stat_publish(a.stat_partials + (((size_t)n * a.Cout + co) * stat_nblk + stat_blk) * 2, u1, u2);
            }
        };
        auto commit_stats = [&]() {
            if (pend_n >= 0) {
                if (a.stat_partials) commit_one(stat_lds + pend_par * 128, pend_n);
                pend_n = -1;
            }
        };
        // The tile whose M accumulators sit in the scratch (complete behind the barrier that ended its last item): output transform + row
        // epilogue, ROW BY ROW inside the matrix loop of the next item -- fin_load(i) fetches the three M rows (and the row operands of the
        // epilogue) one fragment step ahead of fin_row(i), so the LDS / global latency and the row store hide under that item's MFMAs (the
        // first version ran the eight rows back to back before the item: 34-45 % of the kernel, tools/wz_ablate.sh).
        SbOut fso{};
        int fy = 0;
        f32x4 fm[3];
        float4 fradd2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, frbst2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};   // row i lives in slot i & 1
        const int ia = pz, ib = pz + 1, ic = pz + 2;                      // plane 0: M0 + M1 + M2; plane 1: M1 - M2 - M3
        const float sg = pz ? -1.f : 1.f;
        auto fin_prepare = [&](int n, int tz, int ty, int tx) {
            if (n != n_acc) {
                if (n_acc >= 0) flush_stats(n_acc);
                n_acc = n;
            }
            need_kc(n);
            fso = sb_out_prepare<true>(a, n, tz * 2 + pz, tx * 16, cog16, lane);
            fy = ty * 8;
        };
        auto fin_load = [&](int i) __attribute__((always_inline)) {
            const f32x4* S = reinterpret_cast<const f32x4*>(scratch);
            fm[0] = S[((ia * 2 + og) * MT + i) * 64 + lane];
            fm[1] = S[((ib * 2 + og) * MT + i) * 64 + lane];
            fm[2] = S[((ic * 2 + og) * MT + i) * 64 + lane];
        };
        // the row operands of the epilogue (BST: forward tensor y, ADD: residual) are requested a FULL row ahead of their use (two register slots):
        // one fragment step (~300 cycles) covers an LDS read, not the L2 hit the staging waves' prefetch turns these loads into
        auto fin_load_ops = [&](int i) __attribute__((always_inline)) {
            if constexpr ((ADD || BST) && !(dbg & 512)) {     // (devtools bit 512: the row operands are not loaded -- what do these loads cost?)
                const int yy = fy + i;
                const size_t ri = (fso.ok && yy < H) ? sb_out_index<true>(a, fso, yy) : 0;
                if constexpr (ADD) fradd2[i & 1] = *reinterpret_cast<const float4*>(a.add + ri);
                if constexpr (BST) frbst2[i & 1] = *reinterpret_cast<const float4*>(a.bst_y + ri);
            }
        };
        // (component by component: as whole-vector expressions these lower to v_pk_add_f32 / v_pk_fma_f32, and inside the matrix wave a packed-f32
        // instruction costs a whole MFMA issue slot -- tools/coissue_probe.hip -- where two plain instructions cost a third of one)
        auto fin_row = [&](int i) __attribute__((always_inline)) {
            f32x4 vv;
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = fm[0][r] + sg * (fm[1][r] + fm[2][r]);      // plane 0: M0 + (M1 + M2);  plane 1: M1 - (M2 + M3)
            if constexpr ((dbg & 32) != 0) { s1[0] += vv[0] + vv[1] + vv[2] + vv[3]; return; }          // (devtools bit 32: the row is combined but neither counted nor stored)
            const int yy = fy + i;
            if (!(fso.ok && yy < H)) return;
            const float4 fradd = fradd2[i & 1], frbst = frbst2[i & 1];
            if constexpr (ADD) { vv[0] += fradd.x; vv[1] += fradd.y; vv[2] += fradd.z; vv[3] += fradd.w; }      // residual first: the sums are those of the STORED tensor
            if constexpr (BST) {                         // sb_out_tile_bst's arithmetic: u = y*k1 + k2, dh = u > thr ? d : d*slope, S1 += dh, S2' += dh*u
                const float yv[4] = {frbst.x, frbst.y, frbst.z, frbst.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = yv[r] * kc[0][r] + kc[1][r];
                    const float dh = u > kc[2][r] ? vv[r] : vv[r] * a.bst_slope;
                    s1[r] += dh;
                    s2[r] += dh * u;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[r] += vv[r]; s2[r] += vv[r] * vv[r]; }
            }
            *reinterpret_cast<float4*>(a.y + sb_out_index<true>(a, fso, yy)) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        };
        __syncthreads();                                 // item 0 is staged
        int cn, ctz, cty, ctx;
        {
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
        constexpr int NSTEP = 20;                        // (halo row r = 0..9) x (fragment A: dx 0|1, fragment B: dx 2)
        int chunk = 0;
        constexpr bool prof = (dbg & 128) != 0;
        unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
        for (int w = 0; w < nitems; ++w) {
            if constexpr (prof) t0 = __builtin_readcyclecounter();
            const bool last = chunk == nchunk - 1;
            const unsigned wnext = wbase(chunk + 1 < nchunk ? chunk + 1 : 0);
            const u32x4* buf = lds + (w & 1) * BUF;
            commit_stats();
            const bool fin = pending && !(dbg & (8 | 256));      // the previous tile is combined and stored under this item's matrix work
            // (devtools bit 256: no combine and no scratch writes -- compile-time, so the FIN / LAST tests vanish from the stream -- but the accumulators
            // are kept alive by one store after the loop: what the matrix loop costs without its row bookkeeping)
            if (fin) { fin_prepare(pn, ptz, pty, ptx); fin_load_ops(0); }
            pending = false;
            if (chunk == 0) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // (FIN: the previous tile is combined and stored between this item's MFMAs; LAST: this item completes its tile and M_xi goes to the
            // scratch row by row.  Run-time, wave-uniform tests: three compile-time copies of the body -- one branch per item instead of 26 --
            // were built and SPILLED (238-366 VGPRs to scratch memory: the allocator keeps the weights of all copies apart), 2.4x slower.)
            {
                const bool FIN = fin, LAST = last && !(dbg & 256);
                auto frag_ofs = [&](auto S) __attribute__((always_inline)) {
                    constexpr int r = decltype(S)::value / 2, f = decltype(S)::value % 2;
                    return (f == 0 ? fbA : (r < 8 ? fbB : fbB2)) + r * HX;
                };
                if constexpr (prof) { t1 = __builtin_readcyclecounter(); pt[0] += t1 - t0; t0 = t1; }
                constexpr int RING = 3, AH = RING - 1;
                bf16x8 fh[RING], fl[RING];
                static_for<AH>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    fh[j] = __builtin_bit_cast(bf16x8, buf[frag_ofs(std::integral_constant<int, j>{})]);
                    fl[j] = __builtin_bit_cast(bf16x8, buf[frag_ofs(std::integral_constant<int, j>{}) + 2 * HVOLP]);
                });
                static_for<NSTEP>([&](auto S) {
                    constexpr int s = decltype(S)::value, r = s / 2, f = s % 2, cur = s % RING, nxt = (s + AH) % RING;
                    const bf16x8 ah = fh[cur], al = fl[cur];
                    bool fetched = (s + AH >= NSTEP);
                    auto fetch = [&]() __attribute__((always_inline)) {
                        if constexpr (s + AH < NSTEP) {
                            const int o = frag_ofs(std::integral_constant<int, (s + AH < NSTEP ? s + AH : 0)>{});
                            fh[nxt] = __builtin_bit_cast(bf16x8, buf[o]);
                            __builtin_amdgcn_sched_barrier(0);
                            fl[nxt] = __builtin_bit_cast(bf16x8, buf[o + 2 * HVOLP]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        fetched = true;
                    };
                    // (tile, K-step) pairs of this step: A: tiles r, r-1, r-2 with K-steps 0, 1, 2; B: tile r with K-step 3, tile r-2 with K-step 4
                    if constexpr ((dbg & 4) == 0)
                    static_for<3>([&](auto PR) {
                        constexpr int pr = decltype(PR)::value;
                        static_for<3>([&](auto E) {
                            constexpr int e = decltype(E)::value;
                            constexpr int ks = f == 0 ? 2 - e : (e == 0 ? 4 : (e == 1 ? 3 : -1));
                            constexpr int i = f == 0 ? r - (2 - e) : (e == 0 ? r - 2 : (e == 1 ? r : -1));
                            if constexpr (ks >= 0 && i >= 0 && i < MT) {
                                static_for<2>([&](auto GG) {
                                    constexpr int g = decltype(GG)::value;
                                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wreg[g][ks][0]);
                                    const bf16x8 bl = __builtin_bit_cast(bf16x8, wreg[g][ks][1]);
                                    if constexpr (pr == 0) acc[g][i] = mm(al, bh, acc[g][i]);
                                    else if constexpr (pr == 1) acc[g][i] = mm(ah, bl, acc[g][i]);
                                    else acc[g][i] = mm(ah, bh, acc[g][i]);
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (!fetched) fetch();
                                });
                            }
                        });
                    });
                    if constexpr ((dbg & 4) == 0) { if (!fetched) fetch(); }
                    // the weights of a K-step are dead for this item after their last tile: the next chunk's go into the same registers
                    constexpr int ksd = f == 0 ? (r >= 7 ? r - 7 : -1) : (r == 7 ? 3 : (r == 9 ? 4 : -1));
                    if constexpr (ksd >= 0 && !(dbg & 16)) {    // (devtools bit 16: the weights are never refilled)
    #pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            wreg[g][ksd][0] = wload(wnext, g, ksd, 0);
                            wreg[g][ksd][1] = wload(wnext, g, ksd, 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // ---- row bookkeeping between the MFMAs
                    if constexpr (f == 0 && r < MT) {           // previous tile, row r: operands one step ahead of their use
                        if (FIN) { fin_load(r); if constexpr (r + 1 < MT) fin_load_ops(r + 1); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (f == 1 && r < MT) {
                        if (FIN) fin_row(r);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (prof && s == 9) { t1 = __builtin_readcyclecounter(); pt[1] += t1 - t0; t0 = t1; }
                    if constexpr (f == 0 && r >= 3) {           // this tile, row r-3: complete since row r-1, its MFMAs have drained -- M_xi goes to the scratch
                        if (LAST && !(dbg & (8 | 64))) {      // (devtools bit 64: no scratch writes of rows 0-6)
                            f32x4* S = reinterpret_cast<f32x4*>(scratch);
                            S[((xi * 2 + 0) * MT + (r - 3)) * 64 + lane] = acc[0][r - 3];
                            S[((xi * 2 + 1) * MT + (r - 3)) * 64 + lane] = acc[1][r - 3];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if (LAST && !(dbg & 8)) {                    // row 7 (row 6 went out in step r = 9)
                    f32x4* S = reinterpret_cast<f32x4*>(scratch);
                    S[((xi * 2 + 0) * MT + 7) * 64 + lane] = acc[0][7];
                    S[((xi * 2 + 1) * MT + 7) * 64 + lane] = acc[1][7];
                    pending = true;
                    pn = cn; ptz = ctz; pty = cty; ptx = ctx;
                }
            }
            if constexpr (prof) { t1 = __builtin_readcyclecounter(); pt[2] += t1 - t0; t0 = t1; }
            if (++chunk == nchunk) {
                chunk = 0;
                ctx += gx; if (ctx >= ntx) { ctx -= ntx; ++cty; }
                cty += gy; if (cty >= nty) { cty -= nty; ++ctz; }
                ctz += gz; if (ctz >= ntz) { ctz -= ntz; ++cn; }
                cn += gn;
            }
            __syncthreads();
            if constexpr (prof) { t1 = __builtin_readcyclecounter(); pt[3] += t1 - t0; pt[4] += 1; }
        }
        if constexpr (prof) t0 = __builtin_readcyclecounter();
        if constexpr ((dbg & 256) != 0) {
            f32x4 sink = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < MT; ++i) sink += acc[g][i];
            *reinterpret_cast<f32x4*>(a.y + (size_t)(blockIdx.x * 4 + rw) * 256 + lane * 4) = sink;
            pending = false;
        }
        commit_stats();
        if (pending && !(dbg & 8)) {                     // the last tile of this workgroup: nothing left to hide it under
            fin_prepare(pn, ptz, pty, ptx);
#pragma unroll
            for (int i = 0; i < MT; ++i) { fin_load_ops(i); fin_load(i); fin_row(i); }
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
                pt[5] = __builtin_readcyclecounter() - t0;
#pragma unroll
                for (int i = 0; i < 6; ++i) atomicAdd(&wz_prof[i], pt[i]);
                atomicAdd(&wz_prof[6], 1ull);
            }
        }
    }
    fin_tail(a.fin, a.stat_partials, smem);
}

}  // namespace ru
