// conv3_wz32mx.hip -- the Winograd-z forward convolution with fp16 + MX-fp8 products (conv3_wz32mx.hpp): the kernel's own translation unit and its launch.
// conv3_wz_launch (conv3_wz.hip) routes the forward form here when the caller declares its input an activation tensor (Conv3Args::products == 2) and RU_MX is on.
#include "conv3_wz32mx.hpp"

namespace ru {

int conv3_wz32mx_launch(const Conv3Args& a, const void* frag, hipStream_t s) {
    RU_REQUIRE(a.in_c16 && a.out_c16 && !a.bias && !a.sigmoid && !a.in_c4 && !a.in_s16 && a.products == 2 && !a.bst_y && !a.add && conv3_wz_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
               "conv3_wz32mx: the forward form on activations -- voxel-major float32 tensors, >= 32 input channels, whole 32-channel output blocks, an even depth, no residual");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_wz32mx: at most 32 samples per call when statistics are requested");
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wz32mx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WZ_LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_wz32mx)");
        attr_done.set();
    }
    const int ntz = a.D / 2, nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)wz_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)(a.Cout / 32));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)WZ_LDS_BYTES),
               "conv3_wz32mx: tail descriptor does not match the launch");
    hipLaunchKernelGGL(conv3_wz32mx_kernel, grid, dim3(512), WZ_LDS_BYTES, s, a, (const u32x4*)frag, ntz, nty, ntx, a.Cin / 16);
    RU_CHECK_LAUNCH("conv3_wz32mx_kernel");
    return RU_OK;
}

}  // namespace ru

#ifdef RU_SB2_DBG
// tools only (not in include/resunet_hip.h, -DRU_SB2_DBG builds): read and clear the section counters of devtools bit 128
extern "C" int ru_dbg_wz32mx_prof(unsigned long long* out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(ru::wz32mx_prof), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return ru::hip_fail(e, "hipMemcpyFromSymbol(wz32mx_prof)");
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(ru::wz32mx_prof), z, sizeof(z));
    return e == hipSuccess ? RU_OK : ru::hip_fail(e, "hipMemcpyToSymbol(wz32mx_prof)");
}
#endif
