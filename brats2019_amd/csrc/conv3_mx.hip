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

int conv3_mx_launch(const Conv3Args& a, const void* mxfrag, hipStream_t s) {
    using P = SB<4, 8>;
    RU_REQUIRE(a.in_c16 && a.out_c16 && !a.in_s16 && !a.in_c4 && !a.bias && !a.sigmoid && !a.add && !a.bst_y && !a.in_res && conv3_mx_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
               "conv3_mx: the forward form -- voxel-major tensors, 16 input channels, whole 16-channel output blocks, plain float32 activations in, no residual / bias");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_mx: at most 32 samples per call when statistics are requested");
    RU_REQUIRE((size_t)a.D * a.H * a.W * 64 < ((size_t)1 << 31), "conv3_mx: a 16-channel block of the voxel-major input must be smaller than 2 GiB (buffer addressing)");
    static PerDevice attr_done;
    constexpr int LDS2 = 2 * P::LDS_BYTES + SB_STAT_LDS_FLOATS * 4;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_mx)");
        attr_done.set();
    }
    const int ntz = cdiv(a.D, 4), nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)sb2_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)(a.Cout / 16));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)LDS2),
               "conv3_mx: tail descriptor does not match the launch");
    hipLaunchKernelGGL(conv3_mx_kernel, grid, dim3(512), LDS2, s, a, (const u32x4*)mxfrag, ntz, nty, ntx);
    RU_CHECK_LAUNCH("conv3_mx_kernel");
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
