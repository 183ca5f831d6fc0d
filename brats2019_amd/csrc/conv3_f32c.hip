// conv3_f32c.hip -- the exact-f32 3x3x3 convolution on VOXEL-MAJOR tensors (round 5): BASELINE configs[1] (fp32 forward, batch 1) through the
// engine's fused voxel-major flow instead of the round-1 NCDHW flow.
//
//   reference arithmetic: model.py:72-73,336,348 (nn.Conv3d 3x3x3, stride 1, padding 1) on v_mfma_f32_16x16x4_f32 -- every product an exact f32
//   multiply, the sum a k-ordered fmaf chain (MI355X_MICROARCH.md): the same instruction and the same class of rounding as conv3_f32_kernel.
//
// Why a second exact-f32 kernel: conv3_f32_kernel (NCDHW) reaches 50-66 % of the f32 MFMA peak and forces the whole exact-f32 forward through the
// NCDHW kernels -- stand-alone statistics passes, a space-to-depth tensor, the 1x1 convolutions on the fine grid: 28 % of that forward is non-conv
// launches the voxel-major flow fuses away.  This kernel is the persistent producer / consumer skeleton of conv3_sb2_kernel with f32 operands:
//   * 512 threads, one workgroup per CU; waves 4-7 stage the halo image of the NEXT item (global float4 -> fused affine + LeakyReLU -> LDS, channel-major
//     [16 channels][CS] so that an operand read is 64 consecutive dwords per k-group, CS == 16 (mod 32): conflict-free), double-buffered;
//   * waves 0-3 own TZ*TY/4 output rows each; weights as per-lane fragments in registers (27 float4 = 108 VGPRs per 16-channel chunk: tap t, input
//     channels 4q + (lane >> 4), output channel lane & 15), refilled for the next chunk group by group as their taps finish;
//   * per (dz, dx) and 4-channel group the MT + 2 halo-row operands are read once and serve the three dy taps: (MT + 2) ds_read_b32 per 3 MT MFMAs;
//     the matrix instruction issues every 32 cycles, so the wave's stream is one LDS read and one MFMA per slot -- the kernel is bound by the matrix pipe;
//   * output voxel-major (operands exchanged: D[m = cout][n = voxel], a lane stores 4 consecutive couts of a voxel) or NCDHW with bias + sigmoid (the head);
//     input voxel-major or NCDHW (the stem: its 4 channels are ONE 4-channel k-group, no padded matrix work).
// GroupNorm statistics partials: one per (workgroup, sample), as in conv3_sb2_kernel.  Forward only (the exact-f32 TRAINING path keeps the NCDHW engine).
#include "conv3_sb_common.hpp"

namespace ru {

template <int TZ, int TY>
struct F32C {
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = 18;
    static constexpr int HVOL = HZ * HY * HX;
    static constexpr int CS = ((HVOL - 16 + 31) / 32) * 32 + 16;      // >= HVOL, == 16 (mod 32)
    static constexpr int MT = TZ * TY / 4;                             // output rows per consumer wave (all in one z plane)
    static constexpr int BUF_FLOATS = 16 * CS;
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4 + SB_STAT_LDS_FLOATS * 4;
    static constexpr int NR = (HVOL + 255) / 256;                      // staging rounds: 256 halo positions per round
    static_assert(MT >= 1 && TY % MT == 0, "a wave's rows must lie in one z plane");
    static_assert(CS >= HVOL && CS % 32 == 16, "bank layout");
};

struct F32CChoice { int tz, ty; };
static inline long f32c_blocks(int N, int Cout, int D, int H, int W, int tz, int ty) {
    return (long)N * cdiv(D, tz) * cdiv(H, ty) * cdiv(W, 16) * cdiv(Cout, 16);
}
static inline F32CChoice f32c_choose(int N, int Cout, int D, int H, int W) {
    const int ncu = sb_ncu();
    if (f32c_blocks(N, Cout, D, H, W, 4, 8) >= ncu) return {4, 8};
    if (f32c_blocks(N, Cout, D, H, W, 2, 8) >= ncu) return {2, 8};
    return {2, 4};
}
static inline long f32c_grid_x(int N, int Cout, int D, int H, int W, int tz, int ty) {
    const int ncu = sb_ncu(), ncog = cdiv(Cout, 16);
    const long ntile = (long)N * cdiv(D, tz) * cdiv(H, ty) * cdiv(W, 16);
    long gx = ncu / (ncog < ncu ? ncog : ncu);
    if (gx < 1) gx = 1;
    return gx > ntile ? ntile : gx;
}

// weight fragments: unit (cog, chunk, tap) = 64 lanes x float4: lane l (col = l & 15, kq = l >> 4) holds, for q = 0..3,
// W[cout = cog*16 + col][cin = chunk*16 + 4q + kq][tap] (mode 1: mirrored taps, exchanged channel roles) -- zeros beyond the real channels
// HEAD form (at most 4 output channels and one input chunk: conv_output, model.py:348): a 16-wide N tile for 3 channels wastes 13/16 of the matrix
// work, so the N columns carry (dy, cout) pairs instead -- column n = 4 dy + co.  The operand row of halo row R then feeds, in ONE MFMA per (dz, dx) and
// 4-channel group, the three output rows R, R-1, R-2 (taps dy = 0, 1, 2): 9 * 4 * (MT + 2) MFMAs per wave and item instead of 27 * 4 * MT (360 for 864);
// the three contributions of an output row sit in three accumulators' columns (co, dy) and are summed across lanes in the epilogue (two 4-lane shifts).
// Fragment unit (g9 = dz*3 + dx, q): lane l (n = l & 15, kq = l >> 4) holds W[co = n & 3][cin = 4q + kq][dz][dy = n >> 2][dx], zero for dy = 3 or co >= Cout.
static inline bool f32c_head_form(int Cin_conv, int Cout_conv) { return Cout_conv <= 4 && Cin_conv <= 16; }
__global__ void conv3_f32c_pack_head_kernel(const float* __restrict__ w, float* __restrict__ wh, int Cin_f, int Cout_f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 36 * 64) return;
    const int lane = i & 63, u = i >> 6, g9 = u >> 2, q = u & 3;
    const int n = lane & 15, kq = lane >> 4, dy = n >> 2, co = n & 3, ci = 4 * q + kq;
    float v = 0.f;
    if (dy < 3 && co < Cout_f && ci < Cin_f) v = w[((size_t)co * Cin_f + ci) * 27 + (g9 / 3) * 9 + dy * 3 + g9 % 3];
    wh[i] = v;
}
__global__ void conv3_f32c_pack_kernel(const float* __restrict__ w, float4* __restrict__ wfr, int Cin_f, int Cout_f, int mode, int nchunk, int ncog) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncog * nchunk * 27 * 64) return;
    const int lane = i & 63;
    int u = i >> 6;
    const int tap = u % 27; u /= 27;
    const int chunk = u % nchunk;
    const int cog = u / nchunk;
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    const int co = cog * 16 + (lane & 15), kq = lane >> 4;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ci = chunk * 16 + 4 * q + kq;
        v[q] = 0.f;
        if (ci < cin_conv && co < cout_conv) v[q] = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
    }
    wfr[i] = make_float4(v[0], v[1], v[2], v[3]);
}
size_t conv3_f32c_frag_bytes(int Cin_conv, int Cout_conv) { return (size_t)cdiv(Cout_conv, 16) * cdiv(Cin_conv, 16) * 27 * 64 * 16; }
int conv3_f32c_pack_weights(const float* w, void* wfr, int Cin_f, int Cout_f, int mode, hipStream_t s) {
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    if (mode == 0 && f32c_head_form(cin_conv, cout_conv)) {
        hipLaunchKernelGGL(conv3_f32c_pack_head_kernel, dim3(9), dim3(256), 0, s, w, (float*)wfr, Cin_f, Cout_f);
        RU_CHECK_LAUNCH("conv3_f32c_pack_head_kernel");
        return RU_OK;
    }
    const int nchunk = cdiv(cin_conv, 16), ncog = cdiv(cout_conv, 16);
    const int total = ncog * nchunk * 27 * 64;
    hipLaunchKernelGGL(conv3_f32c_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, (float4*)wfr, Cin_f, Cout_f, mode, nchunk, ncog);
    RU_CHECK_LAUNCH("conv3_f32c_pack_kernel");
    return RU_OK;
}

template <int TZ, int TY, bool IN16, bool OUT16, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv3_f32c_kernel(const Conv3Args a, const float4* __restrict__ wfr, int ntz, int nty, int ntx, int nchunk) {
    static_assert(!HEAD || (IN16 && !OUT16), "head form: voxel-major in, NCDHW out");
    using P = F32C<TZ, TY>;
    constexpr int HY = P::HY, HX = P::HX, HVOL = P::HVOL, CS = P::CS, MT = P::MT, BUF = P::BUF_FLOATS, NR = P::NR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stat_lds = smem + 2 * BUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int rw = wave & 3;
    const int ptid = tid & 255;
    const int cog = blockIdx.y;
    const int D = a.D, H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    const int tiles_per_sample = ntz * nty * ntx;
    const int ntile = a.N * tiles_per_sample;
    const int G = gridDim.x;
    const int swz = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int nsteps = swz < ntile ? (ntile - swz + G - 1) / G : 0;
    const int nitems = nsteps * nchunk;
    constexpr int kgroups = IN16 ? 4 : 1;                // k-groups of 4 input channels per chunk (NCDHW input: the stem, Cin <= 4)
    auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
        int b = tile;
        n = b / tiles_per_sample;
        b -= n * tiles_per_sample;
        const int tx = b % ntx; b /= ntx;
        const int ty = b % nty;
        const int tz = b / nty;
        z0 = tz * TZ; y0 = ty * TY; x0 = tx * 16;
    };

    if (producer) {
        // ---------------------------------------------------------------- producers: halo position p = r*256 + ptid, all 16 channels of the chunk
        const bool xform = a.in_scale != nullptr;
        const float slope = xform ? a.in_slope : 1.f;
        int pk[NR], dlt[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int p = r * 256 + ptid;
            const int row = p / HX, xc = p - row * HX;
            const int hz = row / HY, hy = row - hz * HY;
            pk[r] = hz | (hy << 8) | (xc << 16);
            dlt[r] = IN16 ? ((hz * H + hy) * W + xc) * 64 : ((hz * H + hy) * W + xc) * 4;      // byte offset relative to the halo origin
        }
        float4 v[NR][4];                                   // [round][channel quad]
        float4 scq[4], shq[4];                             // fused input transform of the item in flight: requested with its loads, one item ahead
        unsigned vmask = 0;
        int n_cur = 0, chunk_cur = 0;
        // head form: the residual of the input (Conv3Args::in_res: the last Residual block's x, model.py:114) -- its rows come through a ring of RD rounds
        // requested inside store(), RD rounds ahead of their use (conv3_sb2_kernel's head form, same reason: registers)
        constexpr int RD = 2;
        float4 rq[HEAD ? RD : 1][4];
        const bool res = HEAD && a.in_res != nullptr;
        __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0, 0x00020000);
        int res_base = 0;
        auto issue = [&](int item) {
            const int step = item / nchunk, chunk = item - step * nchunk;
            int n, z0, y0, x0;
            tile_origin(swz + step * G, n, z0, y0, x0);
            n_cur = n; chunk_cur = chunk;
            const int zm1 = z0 - 1, ym1 = y0 - 1, xm1 = x0 - 1;
            vmask = 0;
            if (xform && IN16) {                               // (read at store time these 8 loads were an exposed L2 round trip per item)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    scq[q] = *reinterpret_cast<const float4*>(a.in_scale + n * a.Cin + chunk * 16 + q * 4);
                    shq[q] = *reinterpret_cast<const float4*>(a.in_shift + n * a.Cin + chunk * 16 + q * 4);
                }
            }
            if constexpr (IN16) {
                const float* xb = a.x + ((size_t)(n * nchunk + chunk) * DHW) * 16;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(DHW * 64), 0x00020000);
                const int base = ((zm1 * H + ym1) * W + xm1) * 64;
                if constexpr (HEAD) {
                    if (res) {
                        res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in_res + ((size_t)(n * nchunk + chunk) * DHW) * 16), 0, (int)(DHW * 64), 0x00020000);
                        res_base = base;
                    }
                }
                static_for<NR>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    const int gz = zm1 + (pk[r] & 0xff), gy = ym1 + ((pk[r] >> 8) & 0xff), gx = xm1 + ((pk[r] >> 16) & 0xff);
                    bool ok = ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                    if constexpr ((r + 1) * 256 > HVOL) ok = ok & (r * 256 + ptid < HVOL);
                    const unsigned ofs = ok ? (unsigned)(base + dlt[r]) : 0x80000000u;
                    vmask |= ok ? (1u << r) : 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[r][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ofs, q * 16, 0));
                });
            } else {
                // NCDHW input (the stem): one scalar per channel and position; channels beyond Cin are never read by the matrix loop's k-groups
                // except as the zero fill of an incomplete last group
                const int base = ((zm1 * H + ym1) * W + xm1) * 4;
                static_for<NR>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = decltype(R)::value;
                    const int gz = zm1 + (pk[r] & 0xff), gy = ym1 + ((pk[r] >> 8) & 0xff), gx = xm1 + ((pk[r] >> 16) & 0xff);
                    bool ok = ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                    if constexpr ((r + 1) * 256 > HVOL) ok = ok & (r * 256 + ptid < HVOL);
                    vmask |= ok ? (1u << r) : 0u;
                    const size_t pos = ok ? (size_t)((base + dlt[r]) >> 2) : 0;
                    float t[4];                                                // only the k-group the matrix loop reads (channels 0-3) is staged
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = a.x[((size_t)n * a.Cin + (j < a.Cin ? j : 0)) * DHW + pos];      // (clamped: unconditional loads)
                    v[r][0] = make_float4(t[0], t[1], t[2], t[3]);
                });
            }
        };
        auto store = [&](float* buf) {
            float sc[16], sh[16];
            if (xform && IN16) {                               // voxel-major input: whole 16-channel chunks (Cin % 16 == 0)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sc[q * 4] = scq[q].x; sc[q * 4 + 1] = scq[q].y; sc[q * 4 + 2] = scq[q].z; sc[q * 4 + 3] = scq[q].w;
                    sh[q * 4] = shq[q].x; sh[q * 4 + 1] = shq[q].y; sh[q * 4 + 2] = shq[q].z; sh[q * 4 + 3] = shq[q].w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int cg = chunk_cur * 16 + c;
                    const bool cok = cg < a.Cin;
                    sc[c] = cok ? 1.f : 0.f; sh[c] = 0.f;                           // channels beyond Cin: exact zeros
                    if (xform && cok) { sc[c] = a.in_scale[n_cur * a.Cin + cg]; sh[c] = a.in_shift[n_cur * a.Cin + cg]; }
                }
            }
            auto res_load = [&](auto RR) __attribute__((always_inline)) {             // round rr of the item in the registers -> ring slot rr % RD
                constexpr int rr = decltype(RR)::value;
                if constexpr (HEAD && rr < NR) {
                    if (res) {
                        const unsigned ofs = ((vmask >> rr) & 1u) ? (unsigned)(res_base + dlt[rr]) : 0x80000000u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) rq[rr % RD][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, ofs, q * 16, 0));
                    }
                }
            };
            static_for<RD>([&](auto J) { res_load(J); });
            static_for<NR>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value;
                const int p = r * 256 + ptid;
                if (!((r + 1) * 256 > HVOL && p >= HVOL)) {
                    const bool live = (vmask >> r) & 1u;                            // the zero padding applies to the ACTIVATED tensor (and to the sum)
#pragma unroll
                    for (int q = 0; q < kgroups; ++q) {
                        const float f[4] = {v[r][q].x, v[r][q].y, v[r][q].z, v[r][q].w};
                        float g[4] = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (HEAD) {
                            if (res) { g[0] = rq[r % RD][q].x; g[1] = rq[r % RD][q].y; g[2] = rq[r % RD][q].z; g[3] = rq[r % RD][q].w; }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float u = fmaf(f[j], sc[q * 4 + j], sh[q * 4 + j]);
                            float t = fmaxf(u, u * slope);
                            if constexpr (HEAD) { if (res) t = g[j] + t; }          // (gn_apply16_kernel's order: x + activation)
                            buf[(q * 4 + j) * CS + p] = live ? t : 0.f;
                        }
                    }
                }
                res_load(std::integral_constant<int, r + RD>{});                    // into the slot this round has just read
            });
        };
        if (nitems > 0) {
            issue(0);
            store(smem);
            if (nitems > 1) issue(1);
        }
        __syncthreads();
        for (int w = 0; w < nitems; ++w) {
            if (w + 1 < nitems) {
                store(smem + ((w + 1) & 1) * BUF);
                if (w + 2 < nitems) issue(w + 2);
            }
            __syncthreads();
        }
        __syncthreads();                                 // (the consumers' closing barrier: their last statistics flush)
    } else {
        // ---------------------------------------------------------------- consumers
        const int mz = (rw * MT) / TY, my0 = (rw * MT) % TY;
        const int kq = lane >> 4;
        const int abase = kq * CS + (mz * HY + my0) * HX + (lane & 15);          // + 4q*CS + (dz*HY + r)*HX + dx
        float4 wreg[HEAD ? 1 : 27];
        float wh[HEAD ? 36 : 1];                         // head form: one float per (dz, dx) group and 4-channel k-group
        auto wptr = [&](int chunk) { return wfr + ((size_t)(cog * nchunk + chunk) * 27) * 64 + lane; };
        if constexpr (HEAD) {
            const float* wp = reinterpret_cast<const float*>(wfr) + lane;
#pragma unroll
            for (int u = 0; u < 36; ++u) wh[u] = wp[u * 64];
        } else {
            const float4* wp = wptr(0);
#pragma unroll
            for (int t = 0; t < 27; ++t) wreg[t] = wp[t * 64];
        }
        f32x4 acc[HEAD ? MT + 2 : MT];                   // head form: one accumulator per HALO row (its columns are (dy, cout) pairs)
        auto mm = [](float av, float wv, const f32x4& c) -> f32x4 {
            if constexpr (OUT16) return __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av, c, 0, 0, 0);      // D[m = cout][n = voxel]
            else return __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, c, 0, 0, 0);
        };
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        const int stat_blk = blockIdx.x, stat_nblk = G;
        unsigned flushed = 0;
        int n_acc = -1, pend_n = -1, pend_par = 0, par = 0;
        auto flush_stats = [&](int n) {
            if (a.stat_partials) sb_stats_to_lds<OUT16>(s1, s2, stat_lds + (par * 4 + rw) * 32, lane);
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
        int chunk = 0;
        for (int w = 0; w < nitems; ++w) {
            const bool last = chunk == nchunk - 1;
            const float4* wnext = wptr(chunk + 1 < nchunk ? chunk + 1 : 0);
            const float* buf = smem + (w & 1) * BUF;
            commit_stats();
            if (chunk == 0) {
#pragma unroll
                for (int i = 0; i < (HEAD ? MT + 2 : MT); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // (dz, dx) groups x 4-channel k-groups = KQ * 9 blocks: the MT + 2 operand rows of a block are read once and serve its three dy taps.
            // The rows of block b+1 are requested BEFORE block b's MFMAs (two register sets): issued right in front of their first use, the LDS
            // latency of every block was exposed -- 17 % of the matrix loop at 8 rows per wave, 40 % at 2 (the 128-channel level of a batch-1 forward).
            constexpr int KQ = IN16 ? 4 : 1;                 // k-groups of 4 input channels per chunk (NCDHW input: the stem's 4 channels)
            constexpr int NB = 9 * KQ;
            float fr[2][MT + 2];
            auto load_block = [&](auto B, float (&dst)[MT + 2]) __attribute__((always_inline)) {
                constexpr int b = decltype(B)::value, g9 = b / KQ, q = b % KQ, dz = g9 / 3, dx = g9 % 3;
#pragma unroll
                for (int r = 0; r < MT + 2; ++r) dst[r] = buf[abase + q * 4 * CS + (dz * HY + r) * HX + dx];
            };
            load_block(std::integral_constant<int, 0>{}, fr[0]);
            static_for<NB>([&](auto B) {
                constexpr int b = decltype(B)::value, g9 = b / KQ, q = b % KQ, dz = g9 / 3, dx = g9 % 3;
                if constexpr (b + 1 < NB) load_block(std::integral_constant<int, (b + 1 < NB ? b + 1 : 0)>{}, fr[(b + 1) & 1]);
                if constexpr (HEAD) {                    // one MFMA per halo row: columns (dy, co) take its contribution to output rows R, R-1, R-2
#pragma unroll
                    for (int r = 0; r < MT + 2; ++r) acc[r] = mm(fr[b & 1][r], wh[g9 * 4 + q], acc[r]);
                } else {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float4 w4 = wreg[dz * 9 + dy * 3 + dx];
                    const float wv = q == 0 ? w4.x : (q == 1 ? w4.y : (q == 2 ? w4.z : w4.w));
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = mm(fr[b & 1][i + dy], wv, acc[i]);
                }
                }
                if constexpr (q == KQ - 1 && !HEAD) {
                    if (nchunk > 1) {                    // the three taps of this group are dead for this item: the next chunk's go into their registers
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) wreg[dz * 9 + dy * 3 + dx] = wnext[(dz * 9 + dy * 3 + dx) * 64];
                    }
                }
            });
            if (last) {
                const int n = cn;
                if (n != n_acc) {
                    if (n_acc >= 0) flush_stats(n_acc);
                    n_acc = n;
                }
                const SbOut so = sb_out_prepare<OUT16>(a, n, ctz * TZ + mz, ctx * 16, cog, lane);
                const int ybase = cty * TY + my0;
                const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if constexpr (HEAD) {
                        // output row i = column (co, 0) of halo row i + column (co, 1) of halo row i+1 + column (co, 2) of halo row i+2: the lanes of
                        // columns 4 dy + co hold them, four and eight lanes up (same 16-lane group: same four x positions)
                        f32x4 v = acc[i];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (v[r] + sb_row_shl<4>(acc[i + 1][r])) + sb_row_shl<8>(acc[i + 2][r]);
                        sb2_out_row<OUT16, false>(a, so, ybase + i, v, zero4, s1, s2);      // (lanes of columns >= Cout are masked by so.ok)
                    } else {
                        sb2_out_row<OUT16, false>(a, so, ybase + i, acc[i], zero4, s1, s2);
                    }
                }
            }
            if (++chunk == nchunk) {
                chunk = 0;
                ctx += gx; if (ctx >= ntx) { ctx -= ntx; ++cty; }
                cty += gy; if (cty >= nty) { cty -= nty; ++ctz; }
                ctz += gz; if (ctz >= ntz) { ctz -= ntz; ++cn; }
                cn += gn;
            }
            __syncthreads();
        }
        commit_stats();
        if (n_acc >= 0) flush_stats(n_acc);
        __syncthreads();
        commit_stats();
        if (a.stat_partials && rw == 0) {                // zeros for the samples this workgroup did not touch
            for (int n = 0; n < a.N; ++n)
                if (n >= 32 || !((flushed >> n) & 1u)) sb_stats_commit(a, nullptr, n, cog, stat_blk, stat_nblk, lane);
        }
    }
}

template <int TZ, int TY, bool IN16, bool OUT16, bool HEAD = false>
static int f32c_cfg(const Conv3Args& a, const void* wfr, hipStream_t s) {
    using P = F32C<TZ, TY>;
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_f32c_kernel<TZ, TY, IN16, OUT16, HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_f32c)");
        attr_done.set();
    }
    const int ntz = cdiv(a.D, TZ), nty = cdiv(a.H, TY), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)f32c_grid_x(a.N, a.Cout, a.D, a.H, a.W, TZ, TY), (unsigned)cdiv(a.Cout, 16));
    hipLaunchKernelGGL((conv3_f32c_kernel<TZ, TY, IN16, OUT16, HEAD>), grid, dim3(512), P::LDS_BYTES, s, a, (const float4*)wfr, ntz, nty, ntx, cdiv(a.Cin, 16));
    RU_CHECK_LAUNCH("conv3_f32c_kernel");
    return RU_OK;
}

// partials per (sample, channel) the launch writes (== the x extent of its grid)
int conv3_f32c_tiles_per_sample(int N, int Cin, int Cout, int D, int H, int W) {
    (void)Cin;
    const F32CChoice c = f32c_choose(N, Cout, D, H, W);
    return (int)f32c_grid_x(N, Cout, D, H, W, c.tz, c.ty);
}

// the 16 -> <= 4 voxel-major-in / NCDHW-out conv takes the head form, whose staging forms Conv3Args::in_res (RU_HEAD_RES=0: never; conv3_sb_head_takes_residual)
bool conv3_f32c_head_takes_residual(int Cin, int Cout, int W) {
    const char* e = getenv("RU_HEAD_RES");
    return !(e && *e == '0') && f32c_head_form(Cin, Cout) && Cin % 16 == 0 && (W & 3) == 0;
}

int conv3_f32c_launch(const Conv3Args& a, const void* wfr, hipStream_t s) {
    RU_REQUIRE(a.in_c16 || a.out_c16, "conv3_f32c: at least one voxel-major side (the NCDHW kernel is conv3_f32_kernel)");
    RU_REQUIRE(!a.in_res || (a.in_c16 && !a.out_c16 && a.in_scale && !a.in_sum_out && f32c_head_form(a.Cin, a.Cout)),
               "conv3_f32c: a residual of the input is staged by the head form only (inference: nothing is written back)");
    RU_REQUIRE(wfr != nullptr, "conv3_f32c: packed weight fragments missing");
    RU_REQUIRE(!a.add && !a.bst_y && !a.in_s16 && !a.in_c4 && !a.fin.ticket, "conv3_f32c: forward convolution only (no residual / fused backward sums / split-form input / tail)");
    RU_REQUIRE(!a.in_c16 || a.Cin % 16 == 0, "conv3_f32c: voxel-major input needs Cin %% 16 == 0");
    RU_REQUIRE(a.in_c16 || a.Cin <= 4, "conv3_f32c: an NCDHW input has at most four channels (the stem: one 4-channel k-group)");
    RU_REQUIRE(!a.out_c16 || (a.Cout % 16 == 0 && !a.bias && !a.sigmoid), "conv3_f32c: voxel-major output needs Cout %% 16 == 0 and has no bias / activation");
    RU_REQUIRE(a.out_c16 || (a.W & 3) == 0, "conv3_f32c: NCDHW output needs W %% 4 == 0");
    RU_REQUIRE(!a.in_c16 || (size_t)a.D * a.H * a.W * 64 < ((size_t)1 << 31), "conv3_f32c: a 16-channel block of the voxel-major input must be smaller than 2 GiB");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_f32c: at most 32 samples per call when statistics are requested");
    const F32CChoice c = f32c_choose(a.N, a.Cout, a.D, a.H, a.W);
#define RU_F32C_CASE(TZ, TY)                                                                                   \
    if (c.tz == TZ && c.ty == TY) {                                                                            \
        if (a.in_c16 && a.out_c16) return f32c_cfg<TZ, TY, true, true>(a, wfr, s);                             \
        if (a.in_c16) return f32c_head_form(a.Cin, a.Cout) ? f32c_cfg<TZ, TY, true, false, true>(a, wfr, s) : f32c_cfg<TZ, TY, true, false>(a, wfr, s); \
        return f32c_cfg<TZ, TY, false, true>(a, wfr, s);                                                       \
    }
    RU_F32C_CASE(4, 8)
    RU_F32C_CASE(2, 8)
    RU_F32C_CASE(2, 4)
#undef RU_F32C_CASE
    set_error("conv3_f32c: no kernel for tile (%d, %d)", c.tz, c.ty);
    return RU_EINVAL;
}

}  // namespace ru
