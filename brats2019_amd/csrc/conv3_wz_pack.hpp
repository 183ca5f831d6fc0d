// conv3_wz_pack.hpp -- fragment conventions of the Winograd-z convolution (conv3_wz.hpp), the device function that packs one weight tensor into them
// and the shape rule of the kernel: what conv3_sb.hip needs (its batched pack kernel packs every 3x3x3 weight of the network in both forms, its launch
// routine routes by shape) without instantiating the kernel template.
#pragma once
#include "conv3_sb_common.hpp"
#include "conv3_mx_pack.hpp"

namespace ru {

typedef unsigned int wz_u32x4 __attribute__((ext_vector_type(4)));

constexpr int WZ_KSTEPS = 5;
constexpr int WZ_HY = 10, WZ_HX = 18;
constexpr int WZ_PLANE = WZ_HY * WZ_HX;                  // 180 halo positions per transformed plane
constexpr int WZ_HVOLP = 4 * WZ_PLANE;                   // packets per [hi/lo][channel half] section: 4 transformed planes
constexpr int WZ_BUF = 4 * WZ_HVOLP;                     // packets per image buffer (46 080 bytes)
constexpr int WZ_SCRATCH_FLOATS = 4 * 2 * 8 * 64 * 4;    // [wave][group][row][lane] float4: the four waves' M accumulators of one tile
constexpr int WZ_LDS_BYTES = 2 * WZ_BUF * 16 + WZ_SCRATCH_FLOATS * 4 + SB_STAT_LDS_FLOATS * 4 + 1024;      // images, M scratch, statistics scratch, 1 KB landing pad of the operand prefetch
constexpr int WZ_UNITS = 4 * 2 * WZ_KSTEPS * 2;          // 16-byte x 64-lane fragment units per (32-cout block, 16-cin chunk): [xi][group][K-step][hi/lo]

// tap dy*3 + dx (or -1: phantom, zero weights) in K-slot `slot` of K-step ks of one transformed plane
__host__ __device__ constexpr int wz_tap(int ks, int slot) {
    if (ks < 3) return ks * 3 + slot;                    // (dy = ks, dx 0) | (dy = ks, dx 1)
    if (ks == 3) return slot * 3 + 2;                    // (dy 0, dx 2) | (dy 1, dx 2)
    return slot == 0 ? 2 * 3 + 2 : -1;                   // (dy 2, dx 2) | phantom
}


// channel counts for which the transformed fragments exist beside the direct ones (whether a SHAPE takes the kernel: conv3_wz_shape_ok)
__host__ __device__ constexpr bool wz_channels_ok(int Cin_conv, int Cout_conv) { return Cin_conv >= 32 && Cin_conv % 16 == 0 && Cout_conv % 32 == 0; }

// thread i of ncog32 * nchunk * 4 * 2 * WZ_KSTEPS * 64: unit u = ((((cog32*nchunk + chunk)*4 + xi)*2 + g)*WZ_KSTEPS + ks)*2 + hl, 64 lanes x 16 bytes;
// lane l (col = l&15, k-group kg = l>>4) holds, for e = 0..7, G_xi[cout = cog32*32 + g*16 + col][cin = chunk*16 + (kg&1)*8 + e][tap = wz_tap(ks, kg>>1)]
// with G_0 = g0, G_1 = (g0 + g1 + g2)/2, G_2 = (g0 - g1 + g2)/2, G_3 = g2 over the dz slices g_dz of the (mode 1: mirrored, channel-exchanged) weight.
__device__ __forceinline__ void wz_pack_one(const float* __restrict__ w, wz_u32x4* __restrict__ wzfrag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog32, int i) {
    const int total = ncog32 * nchunk * 4 * 2 * WZ_KSTEPS * 64;
    if (i >= total) return;
    const int lane = i & 63;
    int u = i >> 6;
    const int ks = u % WZ_KSTEPS; u /= WZ_KSTEPS;
    const int g = u & 1; u >>= 1;
    const int xi = u & 3; u >>= 2;
    const int chunk = u % nchunk;
    const int cog32 = u / nchunk;
    const int col = lane & 15, kg = lane >> 4;
    const int tap2 = wz_tap(ks, kg >> 1);
    const int co = cog32 * 32 + g * 16 + col;
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = chunk * 16 + (kg & 1) * 8 + e;
        float v = 0.f;
        if (tap2 >= 0) {
            float gz[3];
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const int tap = dz * 9 + tap2;
                gz[dz] = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
            }
            v = xi == 0 ? gz[0] : (xi == 3 ? gz[2] : (xi == 1 ? 0.5f * ((gz[0] + gz[2]) + gz[1]) : 0.5f * ((gz[0] + gz[2]) - gz[1])));
        }
        t[e] = v;
    }
    wz_u32x4 hi, lo;
    split_n<4>(t, hi, lo);
    const size_t unit = ((((size_t)(cog32 * nchunk + chunk) * 4 + xi) * 2 + g) * WZ_KSTEPS + ks) * 2;
    wzfrag[(unit + 0) * 64 + lane] = hi;
    wzfrag[(unit + 1) * 64 + lane] = lo;
}

// ---- the 32x32x16 form of the matrix waves (conv3_wz32.hpp): one K-step per tap, 32 output channels per fragment row
constexpr int WZ32_UNITS = 4 * 9 * 2;                    // 16-byte x 64-lane fragment units per (32-cout block, 16-cin chunk): [xi][tap dy*3+dx][hi/lo]
// thread i of ncog32 * nchunk * 4 * 9 * 64: unit u = (((cog32*nchunk + chunk)*4 + xi)*9 + tap)*2 + hl; lane l (row = l&31, K half kh = l>>5) holds,
// for e = 0..7, G_xi[cout = cog32*32 + row][cin = chunk*16 + kh*8 + e][tap] (G_xi as in wz_pack_one)
__device__ __forceinline__ void wz32_pack_one(const float* __restrict__ w, wz_u32x4* __restrict__ frag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog32, int i) {
    const int total = ncog32 * nchunk * 4 * 9 * 64;
    if (i >= total) return;
    const int lane = i & 63;
    int u = i >> 6;
    const int tap2 = u % 9; u /= 9;
    const int xi = u & 3; u >>= 2;
    const int chunk = u % nchunk;
    const int cog32 = u / nchunk;
    const int co = cog32 * 32 + (lane & 31), kh = lane >> 5;
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = chunk * 16 + kh * 8 + e;
        float gz[3];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int tap = dz * 9 + tap2;
            gz[dz] = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
        }
        t[e] = xi == 0 ? gz[0] : (xi == 3 ? gz[2] : (xi == 1 ? 0.5f * ((gz[0] + gz[2]) + gz[1]) : 0.5f * ((gz[0] + gz[2]) - gz[1])));
    }
    wz_u32x4 hi, lo;
    split_n<4>(t, hi, lo);
    const size_t unit = ((((size_t)(cog32 * nchunk + chunk) * 4 + xi) * 9 + tap2)) * 2;
    frag[(unit + 0) * 64 + lane] = hi;
    frag[(unit + 1) * 64 + lane] = lo;
}
static inline size_t wz32_frag_bytes(int Cin_conv, int Cout_conv) {
    return wz_channels_ok(Cin_conv, Cout_conv) ? (size_t)(Cout_conv / 32) * (Cin_conv / 16) * WZ32_UNITS * 64 * 16 : 0;
}

// ---- the fp16 + MX-fp8 product scheme on the 32x32 matrix form (conv3_wz32mx.hpp): per transformed plane nine fp16 tap units and five CROSS units of two
// 16-byte halves -- tap pairs (0,0)+(0,1), (0,2)+(1,0), (1,1)+(1,2), (2,0)+(2,1), (2,2)+phantom in the two slots of v_mfma_scale_f32_32x32x64_f8f6f4
constexpr int WZ32MX_UNITS_XI = 9 + 5 * 2;               // 16-byte x 64-lane units per (32-cout block, 16-cin chunk, transformed plane)
constexpr int WZ32MX_UNITS = 4 * WZ32MX_UNITS_XI;
// tap dy*3 + dx (-1: phantom) in slot `slot` of cross pair p
__host__ __device__ constexpr int wz32mx_pair_tap(int p, int slot) { return 2 * p + slot < 9 ? 2 * p + slot : -1; }
// unit u = ((cog32*nchunk + chunk)*4 + xi)*19 + j.  j < 9: lane l (row = l&31, K half kh = l>>5) holds 8 x fp16 of
// G_xi[cout = cog32*32 + row][cin = chunk*16 + kh*8 + e][tap j] (wz32_pack_one's fragment in fp16); j = 9 + 2 p + slot: lane l (row, k-group kg = l>>5) holds
// 16 x e4m3 over cin = chunk*16 + 0..15 of tap wz32mx_pair_tap(p, slot): kg == 0 ? G * 2^8 : (G - f16(G)) * 2^19 (the weight operands of the two cross terms).
// Thread i of ncog32 * nchunk * 19 * 64 packs its (j, lane) for ALL FOUR transformed planes: the three dz slices of a weight element are loaded once, not once
// per plane (the pack of a training step is a gather of 27-float-strided elements: with one thread per plane it was half of a 150 us launch).
constexpr int WZ32MX_PACK_THREADS_PER_CHUNK = WZ32MX_UNITS_XI * 64;
__device__ __forceinline__ void wz32mx_pack_one(const float* __restrict__ w, wz_u32x4* __restrict__ frag, int Cin_f, int Cout_f, int mode, int nchunk, int ncog32, int i) {
    const int total = ncog32 * nchunk * WZ32MX_PACK_THREADS_PER_CHUNK;
    if (i >= total) return;
    mx_set_saturating_conversions();
    const int lane = i & 63;
    int u = i >> 6;
    const int j = u % WZ32MX_UNITS_XI; u /= WZ32MX_UNITS_XI;
    const int chunk = u % nchunk;
    const int cog32 = u / nchunk;
    const int co = cog32 * 32 + (lane & 31), kg = lane >> 5;
    const bool main_unit = j < 9;
    const int tap2 = main_unit ? j : wz32mx_pair_tap((j - 9) >> 1, (j - 9) & 1);
    const int nel = main_unit ? 8 : 16, ci0 = chunk * 16 + (main_unit ? kg * 8 : 0);
    float gz[16][3];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            float v = 0.f;
            if (e < nel && tap2 >= 0) {
                const int tap = dz * 9 + tap2, ci = ci0 + e;
                v = mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
            }
            gz[e][dz] = v;
        }
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        float g[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            g[e] = xi == 0 ? gz[e][0] : (xi == 3 ? gz[e][2] : (xi == 1 ? 0.5f * ((gz[e][0] + gz[e][2]) + gz[e][1]) : 0.5f * ((gz[e][0] + gz[e][2]) - gz[e][1])));
        wz_u32x4 out;
        if (main_unit) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mx_f16x2 h;
                h[0] = (_Float16)g[2 * c];
                h[1] = (_Float16)g[2 * c + 1];
                out[c] = __builtin_bit_cast(unsigned, h);
            }
        } else {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = kg == 0 ? g[e] * (float)(1 << MX_SWH) : (g[e] - (float)(_Float16)g[e]) * (float)(1 << MX_SWL);
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = mx_cvt4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
        }
        frag[(((size_t)(cog32 * nchunk + chunk) * 4 + xi) * WZ32MX_UNITS_XI + j) * 64 + lane] = out;
    }
}
static inline size_t wz32mx_frag_bytes(int Cin_conv, int Cout_conv) {
    return wz_channels_ok(Cin_conv, Cout_conv) ? (size_t)(Cout_conv / 32) * (Cin_conv / 16) * WZ32MX_UNITS * 64 * 16 : 0;
}

// (the 16x16x32 form's fragments -- conv3_wz_kernel -- exist in devtools builds only: the product library neither packs nor reserves them)
#ifdef RU_SB2_DBG
constexpr bool WZ16_FORM = true;
#else
constexpr bool WZ16_FORM = false;
#endif
static inline size_t wz_frag_bytes(int Cin_conv, int Cout_conv) {
    return (WZ16_FORM && wz_channels_ok(Cin_conv, Cout_conv)) ? (size_t)(Cout_conv / 32) * (Cin_conv / 16) * WZ_UNITS * 64 * 16 : 0;
}

// shapes the kernel takes: voxel-major in and out, several input chunks, whole 32-channel output blocks, an even number of planes, and at least
// one (2,8,16) tile x 32-cout block per CU (below that the one-stage kernel's smaller tiles fill the chip better)
static inline bool conv3_wz_shape_ok(int N, int Cin, int Cout, int D, int H, int W) {
    if (Cin < 32 || Cin % 16 != 0 || Cout % 32 != 0 || (D & 1)) return false;
    if ((size_t)D * H * W * 64 >= ((size_t)1 << 31)) return false;
    const long items = (long)N * (D / 2) * cdiv(H, 8) * cdiv(W, 16) * (Cout / 32);
    return items >= sb_ncu();
}
static inline long wz_grid_x(int N, int Cout, int D, int H, int W) {
    const int ncu = sb_ncu(), ncog = Cout / 32;
    const long ntile = (long)N * (D / 2) * cdiv(H, 8) * cdiv(W, 16);
    long gx = ncu / (ncog < ncu ? ncog : ncu);
    if (gx < 1) gx = 1;
    return gx > ntile ? ntile : gx;
}

int conv3_wz_launch(const Conv3Args& a, const void* wzfrag, hipStream_t s);
int conv3_wz32_launch(const Conv3Args& a, const void* wz32frag, hipStream_t s);      // conv3_wz32.hip: forward form only (no residual, no GroupNorm-backward sums)
bool conv3_wz32_enabled();                               // devtools builds: RU_WZ32=0 selects the 16x16x32 matrix form (same-box A/B); always true in the product library
int conv3_wz32mx_launch(const Conv3Args& a, const void* wz32mxfrag, hipStream_t s);  // conv3_wz32mx.hip: the forward form with fp16 + MX-fp8 products (Conv3Args::products == 2, RU_MX)

}  // namespace ru
