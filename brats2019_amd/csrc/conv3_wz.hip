// conv3_wz.hip -- launch of the Winograd-z forward convolutions (F(2,3) along z; conv3_wz.hpp has the staging waves they share): the fp16 + MX-fp8 form
// (conv3_wz32mx.hpp) for launches that declare their input an activation tensor, else the three-product form on 32x32x16 MFMAs (conv3_wz32.hpp).
// The first matrix form of the kernel, conv3_wz_kernel on 16x16x32 MFMAs, is instantiated in devtools builds only (-DRU_SB2_DBG, RU_WZ32=0 there: same-box
// A/B, section counters); its data-gradient variants (residual add / GroupNorm-backward sums: round 5's RU_WZ=2 / 3) were measured slower than the direct
// DMA-staged kernels twice and are retired (profiles/r05_notes.txt, sections 3 and 7).
#include "conv3_wz.hpp"

namespace ru {

#ifdef RU_SB2_DBG
static int wz_cfg16(const Conv3Args& a, const void* wzfrag, hipStream_t s) {
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wz_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WZ_LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_wz)");
        attr_done.set();
    }
    const int ntz = a.D / 2, nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)wz_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)(a.Cout / 32));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)WZ_LDS_BYTES),
               "conv3_wz: tail descriptor does not match the launch");
    hipLaunchKernelGGL((conv3_wz_kernel<false, false>), grid, dim3(512), WZ_LDS_BYTES, s, a, (const u32x4*)wzfrag, ntz, nty, ntx, a.Cin / 16);
    RU_CHECK_LAUNCH("conv3_wz_kernel");
    return RU_OK;
}
#endif

int conv3_wz_launch(const Conv3Args& a, const void* wzfrag, hipStream_t s) {
    RU_REQUIRE(a.in_c16 && a.out_c16 && !a.bias && !a.sigmoid && !a.in_c4 && !a.in_s16 && !a.bst_y && !a.add && a.products != 1 && conv3_wz_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
               "conv3_wz: the forward form -- voxel-major float32 tensors, >= 32 input channels, whole 32-channel output blocks, an even depth, no residual / GroupNorm-backward sums");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_wz: at most 32 samples per call when statistics are requested");
    if (a.products == 2 && conv3_mx_wz_enabled())        // an activation tensor: fp16 + MX-fp8 products (conv3_wz32mx.hpp), fragments behind the three-product forms
        return conv3_wz32mx_launch(a, static_cast<const char*>(wzfrag) + wz_frag_bytes(a.Cin, a.Cout) + wz32_frag_bytes(a.Cin, a.Cout), s);
#ifdef RU_SB2_DBG
    if (!conv3_wz32_enabled()) return wz_cfg16(a, wzfrag, s);
#endif
    return conv3_wz32_launch(a, static_cast<const char*>(wzfrag) + wz_frag_bytes(a.Cin, a.Cout), s);     // matrix waves on 32x32x16 MFMAs (conv3_wz32.hpp)
}

}  // namespace ru

#ifdef RU_SB2_DBG
// tools only (not in include/resunet_hip.h, -DRU_SB2_DBG builds): read and clear the section counters of devtools bit 128
extern "C" int ru_dbg_wz_prof(unsigned long long* out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(ru::wz_prof), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return ru::hip_fail(e, "hipMemcpyFromSymbol(wz_prof)");
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(ru::wz_prof), z, sizeof(z));
    return e == hipSuccess ? RU_OK : ru::hip_fail(e, "hipMemcpyToSymbol(wz_prof)");
}
#endif
