// conv3_mx_pack.hpp -- what conv3_sb.hip needs of the fp16 + MX-fp8 scheme (conv3_mx.hpp has the kernel and the description): constants, conversions, the weight
// fragment packer, the launch rule.
#pragma once
#include "conv3_sb_common.hpp"

namespace ru {

constexpr int MX_SX = 11, MX_SWH = 8, MX_SWL = 19;
static_assert(MX_SX + MX_SWH == MX_SWL, "one pair of hardware scales serves both cross terms");
constexpr int MX_SCALE_ACT = (127 - MX_SX) * 0x01010101;       // E8M0 2^-11 in every byte (op_sel 0 reads byte 0)
constexpr int MX_SCALE_W = (127 - MX_SWH) * 0x01010101;        // 2^-8
// Gradient operands (conv3_mx_kernel<GRAD>, round 6 late): bf16 main term (a gradient's range needs bf16's exponent), cross terms e4m3(lo * 2^(8-e)) * e4m3(w * 2^8) +
// e4m3(g * 2^-e) * e4m3(w_lo * 2^16) (packets 2 / 3 of a voxel: both planes of channels 0-7 / 8-15, so that each channel half is written by one thread) with ONE exponent e per voxel (its 16 channels, both planes: a scale block of the instruction), carried as the E8M0 byte 127 + e
// in a byte plane beside the tensor; lo = g - bf16(g) <= 2^-8 of the voxel's largest value, which e puts into [128, 256).
constexpr int MXG_SX = 8, MXG_SWH = 8, MXG_SWL = 16;
static_assert(MXG_SX + MXG_SWH == MXG_SWL, "one weight-side scale serves both cross terms");
constexpr int MXG_SCALE_W = (127 - MXG_SWL) * 0x01010101;      // 2^-16; the data side's byte is 127 + e per voxel
constexpr int MX_UNITS = 28;                                   // 16-byte x 64-lane units per 16-cout group: 14 fp16 K-steps, 2 x 3 x 2 cross, 2 ninth chain

typedef _Float16 mx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mx_f16x2 __attribute__((ext_vector_type(2)));
typedef int mx_i32x8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr bool mx_channels_ok(int Cin_conv, int Cout_conv) { return Cin_conv == 16 && Cout_conv % 16 == 0; }
static inline size_t mx_frag_bytes(int Cin_conv, int Cout_conv) { return mx_channels_ok(Cin_conv, Cout_conv) ? (size_t)(Cout_conv / 16) * MX_UNITS * 64 * 16 : 0; }

// tap (dz*9 + dy*3 + dx) of cross fragment X(h, .) for row tap dy, tap pair j = g >> 1, slot
__host__ __device__ constexpr int mx_cross_tap(int h, int j, int slot, int dy) {
    const int c = 4 * h + 2 * j + slot;
    return (c / 3) * 9 + dy * 3 + c % 3;
}
// ... of the ninth-chain fragment N: (j, slot) = (0,0), (0,1), (1,0) are dy = 0, 1, 2; (1,1) is a phantom (zero weights)
__host__ __device__ constexpr int mx_ninth_tap(int j, int slot) {
    const int dy = 2 * j + slot;
    return dy < 3 ? 2 * 9 + dy * 3 + 2 : -1;
}

// two floats -> two e4m3 bytes in the low / high half of a dword (round to nearest even; saturating under MODE.FP16_OVFL)
__device__ __forceinline__ unsigned mx_cvt4(float a, float b, float c, float d) {
    // (the first conversion's pass-through operand is `a` itself: its other half is overwritten by the second one, and a zero would cost a v_mov per dword)
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, __builtin_bit_cast(int, a), false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}
__device__ __forceinline__ void mx_set_saturating_conversions() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }    // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL

// 8 values -> 4 dwords of fp16 (RNE), 2 dwords of e4m3(lo * 2^SX) and 2 dwords of e4m3(v).  2.5 VALU per value (the bf16 split: 3): per pair one
// v_cvt_pk_f16_f32, two v_fma_mix_f32 (lo = v - f16(v) straight from the packed half: exact; written as inline assembly because the compiler converts every
// value a second time on its own -- v_cvt_f16_f32 + v_cvt_f32_f16 + v_sub_f32, 5.5 per value), one v_cvt_scalef32_pk_fp8_f32 (divides by its scale operand: the
// 2^SX costs no multiply; bit-identical to multiply + v_cvt_pk_fp8_f32 incl. saturation and subnormals, tools/mx_cvt_probe.hip) and one v_cvt_pk_fp8_f32.
typedef short mx_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mx_split8(const float (&t)[8], u32x4& h16, unsigned (&l8)[2], unsigned (&x8)[2]) {
    float lo[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mx_f16x2 h;
        h[0] = (_Float16)t[2 * c];
        h[1] = (_Float16)t[2 * c + 1];
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        h16[c] = hb;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * c]) : "v"(hb), "v"(t[2 * c]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * c + 1]) : "v"(hb), "v"(t[2 * c + 1]));
    }
    constexpr float inv = 1.f / (float)(1 << MX_SX);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        mx_s16x2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(mx_s16x2, lo[4 * d]), lo[4 * d], lo[4 * d + 1], inv, false);
        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lo[4 * d + 2], lo[4 * d + 3], inv, true);
        l8[d] = __builtin_bit_cast(unsigned, w);
        x8[d] = mx_cvt4(t[4 * d], t[4 * d + 1], t[4 * d + 2], t[4 * d + 3]);
    }
}

// 4 values (the Winograd-z staging: a lane owns a channel quad) -> 2 dwords of fp16, one dword of e4m3(lo * 2^SX), one dword of e4m3(v)
__device__ __forceinline__ void mx_split4(const float (&t)[4], uint2& h16, unsigned& l8, unsigned& x8) {
    float lo[4];
    unsigned hb[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        mx_f16x2 h;
        h[0] = (_Float16)t[2 * c];
        h[1] = (_Float16)t[2 * c + 1];
        hb[c] = __builtin_bit_cast(unsigned, h);
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * c]) : "v"(hb[c]), "v"(t[2 * c]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * c + 1]) : "v"(hb[c]), "v"(t[2 * c + 1]));
    }
    h16 = make_uint2(hb[0], hb[1]);
    constexpr float inv = 1.f / (float)(1 << MX_SX);
    mx_s16x2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(mx_s16x2, lo[0]), lo[0], lo[1], inv, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lo[2], lo[3], inv, true);
    l8 = __builtin_bit_cast(unsigned, w);
    x8 = mx_cvt4(t[0], t[1], t[2], t[3]);
}

// ------------------------------------------------------------------ weight fragments (packed behind the direct ones of the same weight)
// unit u of 16-cout group cog, lane l (col = l & 15 = output channel, k-group g = l >> 4):
//   u < 14            fp16 K-step u: 8 x fp16 of W[co][ci = (g&1)*8 + e][sb_tap(u, g>>1)]                       (the direct kernel's fragment in fp16)
//   u = 14 + (3h+dy)*2 + slot   cross: 16 x e4m3 over ci = 0..15 of tap mx_cross_tap(h, g>>1, slot, dy): (g&1) == 0 ? w * 2^8 : (w - f16(w)) * 2^19
//   u = 26 + slot     ninth chain: the same at tap mx_ninth_tap(g>>1, slot)
// grad: the fragments of the gradient-operand form -- units < 14 hold bf16(w) (the direct kernel's hi fragments), the cross units w * 2^8 | (w - bf16(w)) * 2^16
__device__ __forceinline__ void mx_pack_one(const float* __restrict__ w, u32x4* __restrict__ mxfrag, int Cin_f, int Cout_f, int mode, int ncog, int i, bool grad = false) {
    if (i >= ncog * MX_UNITS * 64) return;
    mx_set_saturating_conversions();
    const int lane = i & 63, unit = (i >> 6) % MX_UNITS, cog = (i >> 6) / MX_UNITS;
    const int col = lane & 15, g = lane >> 4, co = cog * 16 + col;
    const int cin_conv = mode == 0 ? Cin_f : Cout_f, cout_conv = mode == 0 ? Cout_f : Cin_f;
    auto wat = [&](int ci, int tap) -> float {
        if (tap < 0 || ci >= cin_conv || co >= cout_conv) return 0.f;
        return mode == 0 ? w[((size_t)co * Cin_f + ci) * 27 + tap] : w[((size_t)ci * Cin_f + co) * 27 + (26 - tap)];
    };
    u32x4 out;
    if (unit < SB_KSTEPS) {
        const int tap = sb_tap(unit, g >> 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float w0 = wat((g & 1) * 8 + 2 * c, tap), w1 = wat((g & 1) * 8 + 2 * c + 1, tap);
            if (grad) {
                ru_bf16x2 h;
                h[0] = (__bf16)w0; h[1] = (__bf16)w1;
                out[c] = __builtin_bit_cast(unsigned, h);
            } else {
                mx_f16x2 h;
                h[0] = (_Float16)w0; h[1] = (_Float16)w1;
                out[c] = __builtin_bit_cast(unsigned, h);
            }
        }
    } else {
        int tap;
        if (unit < 26) {
            const int u = unit - SB_KSTEPS, slot = u & 1, dy = (u >> 1) % 3, h = u / 6;
            tap = mx_cross_tap(h, g >> 1, slot, dy);
        } else {
            tap = mx_ninth_tap(g >> 1, unit - 26);
        }
        float v[16];
        if (grad) {                                        // the operand's packet 2 + (g & 1) holds e4m3(lo) | e4m3(value) of channels 8 (g & 1) .. + 7: w * 2^8 | w_lo * 2^16 of the same channels
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float x = wat((g & 1) * 8 + k, tap);
                v[k] = x * (float)(1 << MXG_SWH);
                v[8 + k] = (x - (float)(__bf16)x) * (float)(1 << MXG_SWL);
            }
        } else {
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) {
                const float x = wat(ci, tap);
                v[ci] = (g & 1) == 0 ? x * (float)(1 << MX_SWH) : (x - (float)(_Float16)x) * (float)(1 << MX_SWL);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = mx_cvt4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
    mxfrag[((size_t)cog * MX_UNITS + unit) * 64 + lane] = out;
}

// launch rule and entry (conv3_mx.hip)
bool conv3_mx_enabled();                                 // RU_MX=0: every forward convolution keeps the three-product kernels (same-box A/B, parity tests)
bool conv3_mx_wz_enabled();                              // ... and the Winograd-z form of the scheme at 32..128 channels (conv3_wz32mx.hpp); RU_MX=1: the 16-channel kernel only
bool conv3_mx_shape_ok(int N, int Cin, int Cout, int D, int H, int W);
int conv3_mx_launch(const Conv3Args& a, const void* mxfrag, hipStream_t s);
// (conv3_mxg_enabled / conv3_mxg_usable / conv3_mxg_split_launch: ru_common.h -- the engine and the weight gradient use them)
}  // namespace ru
