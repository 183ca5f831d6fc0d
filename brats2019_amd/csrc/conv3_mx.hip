// conv3_mx.hip -- the fp16 + MX-fp8 product scheme of the 16-channel level's forward convolutions (conv3_mx.hpp): the kernel's own translation unit and its
// launch.  conv3_sb_launch routes a launch here when the caller asks for the scheme (Conv3Args::products == 2: the engine's forward convolutions, whose inputs
// are activations -- never gradients, whose magnitudes fp16 and a fixed e4m3 scale do not cover) and the shape takes the persistent kernel.
#include "conv3_mx.hpp"

namespace ru {

bool conv3_mx_enabled() {
    const char* e = getenv("RU_MX");                    // read per call: tests and tools switch it inside one process
    return !(e && *e == '0');
}

bool conv3_mx_wz_enabled() {                          // RU_MX=1: the 16-channel kernel only (same-box A/B of the Winograd-z form); RU_MX=0: neither
    const char* e = getenv("RU_MX");
    return !(e && (*e == '0' || *e == '1'));
}

bool conv3_mx_shape_ok(int N, int Cin, int Cout, int D, int H, int W) {
    return mx_channels_ok(Cin, Cout) && sb_use_v2(sb_choose(N, Cout, D, H, W));
}

bool conv3_mxg_enabled() {                            // RU_MXG=0: the 16-channel data-gradient convolutions keep three bf16 products (same-box A/B, parity tests)
    const char* e = getenv("RU_MXG");
    return !(e && *e == '0');
}

bool conv3_mxg_usable(int N, int Cin, int Cout, int D, int H, int W) {
    return conv3_mxg_enabled() && Cin == 16 && Cout == 16 && conv3_mx_shape_ok(N, Cin, Cout, D, H, W) && (size_t)D * H * W * 64 < ((size_t)1 << 31);
}

template <bool GRAD, bool BST, bool ADD>
static int mx_cfg(const Conv3Args& a, const void* mxfrag, hipStream_t s) {
    using P = SB<4, 8>;
    static PerDevice attr_done;
    constexpr int LDS2 = 2 * P::LDS_BYTES + SB_STAT_LDS_FLOATS * 4 + (GRAD ? 2 * ((P::HVOLP + 15) & ~15) : 0);
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mx_kernel<GRAD, BST, ADD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_mx)");
        attr_done.set();
    }
    const int ntz = cdiv(a.D, 4), nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)sb2_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)(a.Cout / 16));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)LDS2),
               "conv3_mx: tail descriptor does not match the launch");
    hipLaunchKernelGGL((conv3_mx_kernel<GRAD, BST, ADD>), grid, dim3(512), LDS2, s, a, (const u32x4*)mxfrag, ntz, nty, ntx);
    RU_CHECK_LAUNCH("conv3_mx_kernel");
    return RU_OK;
}

int conv3_mx_launch(const Conv3Args& a, const void* mxfrag, hipStream_t s) {
    RU_REQUIRE(a.in_c16 && a.out_c16 && !a.in_c4 && !a.bias && !a.sigmoid && !a.in_res && conv3_mx_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
               "conv3_mx: voxel-major tensors, 16 input channels, whole 16-channel output blocks, no bias / input residual");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_mx: at most 32 samples per call when statistics are requested");
    RU_REQUIRE((size_t)a.D * a.H * a.W * 64 < ((size_t)1 << 31), "conv3_mx: a 16-channel block of the voxel-major input must be smaller than 2 GiB (buffer addressing)");
    if (a.in_g16) {                                       // a gradient in the operand form of the scheme: the data-gradient convolutions of the 16-channel level
        RU_REQUIRE(a.in_s16 && !a.in_scale && a.Cout == 16 && (!a.bst_y || (a.bst_k && a.stat_partials)),
                   "conv3_mx: the gradient form -- operand-form input, no input transform, 16 output channels");
        if (a.bst_y) return a.add ? mx_cfg<true, true, true>(a, mxfrag, s) : mx_cfg<true, true, false>(a, mxfrag, s);
        return a.add ? mx_cfg<true, false, true>(a, mxfrag, s) : mx_cfg<true, false, false>(a, mxfrag, s);
    }
    RU_REQUIRE(!a.in_s16 && !a.add && !a.bst_y, "conv3_mx: the forward form -- plain float32 activations in, no residual, no GroupNorm-backward sums");
    return mx_cfg<false, false, false>(a, mxfrag, s);
}

// fp32 voxel-major gradient -> operand form (one thread per voxel half; the partner lane holds the other 8 channels)
__global__ __launch_bounds__(256) void mxg_split_kernel(const float* __restrict__ x, u32x4* __restrict__ g16, size_t nvox) {
    mx_set_saturating_conversions();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t vox = i >> 1;
    const int hsel = (int)(i & 1);
    const bool ok = vox < nvox;
    float t[8];
    const float4 v0 = ok ? *reinterpret_cast<const float4*>(x + vox * 16 + hsel * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = ok ? *reinterpret_cast<const float4*>(x + vox * 16 + hsel * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    t[0] = v0.x; t[1] = v0.y; t[2] = v0.z; t[3] = v0.w; t[4] = v1.x; t[5] = v1.y; t[6] = v1.z; t[7] = v1.w;
    u32x4 hi;
    float lo[8];
    float m = mxg_hi8(t, hi, lo);
    m = fmaxf(m, __shfl_xor(m, 1));
    unsigned l8[2], x8[2];
    mxg_cvt8(t, lo, m, l8, x8);
    if (!ok) return;
    g16[vox * 4 + hsel] = hi;
    g16[vox * 4 + 2 + hsel] = u32x4{l8[0], l8[1], x8[0], x8[1]};
}
int conv3_mxg_split_launch(const float* x, void* g16, size_t nvox, hipStream_t s) {
    RU_REQUIRE(x && g16 && nvox > 0 && nvox * 2 < ((size_t)1 << 39), "conv3_mxg_split: bad argument");
    hipLaunchKernelGGL(mxg_split_kernel, dim3((unsigned)((nvox * 2 + 255) / 256)), dim3(256), 0, s, x, (u32x4*)g16, nvox);
    RU_CHECK_LAUNCH("mxg_split_kernel");
    return RU_OK;
}

}  // namespace ru

#ifdef RU_SB2_DBG
// tools only (not in include/resunet_hip.h, -DRU_SB2_DBG builds): read and clear the section counters of devtools bit 64
extern "C" int ru_dbg_mx_prof(unsigned long long* out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(ru::mx_prof), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return ru::hip_fail(e, "hipMemcpyFromSymbol(mx_prof)");
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(ru::mx_prof), z, sizeof(z));
    return e == hipSuccess ? RU_OK : ru::hip_fail(e, "hipMemcpyToSymbol(mx_prof)");
}
#endif
