// conv3_wz.hip -- the Winograd-z form of the split-bf16 3x3x3 convolution (conv3_wz.hpp): its four variants (plain / residual add / fused
// GroupNorm-backward sums / both) in their own translation unit, and the launch.
#include "conv3_wz.hpp"

namespace ru {

template <bool BST, bool ADD>
static int wz_cfg(const Conv3Args& a, const void* wzfrag, hipStream_t s) {
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wz_kernel<BST, ADD>), hipFuncAttributeMaxDynamicSharedMemorySize, WZ_LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_wz)");
        attr_done.set();
    }
    const int ntz = a.D / 2, nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)wz_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)(a.Cout / 32));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)WZ_LDS_BYTES),
               "conv3_wz: tail descriptor does not match the launch");
    hipLaunchKernelGGL((conv3_wz_kernel<BST, ADD>), grid, dim3(512), WZ_LDS_BYTES, s, a, (const u32x4*)wzfrag, ntz, nty, ntx, a.Cin / 16);
    RU_CHECK_LAUNCH("conv3_wz_kernel");
    return RU_OK;
}

int conv3_wz_launch(const Conv3Args& a, const void* wzfrag, hipStream_t s) {
    RU_REQUIRE(a.in_c16 && a.out_c16 && !a.bias && !a.sigmoid && !a.in_c4 && a.products != 1 && conv3_wz_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
               "conv3_wz: voxel-major tensors, >= 32 input channels, whole 32-channel output blocks, an even depth, three products");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_wz: at most 32 samples per call when statistics are requested");
    RU_REQUIRE(!a.in_s16 || !a.in_scale, "conv3_wz: a split-form input has no fused transform");
    RU_REQUIRE(!a.bst_y || (a.bst_k && a.stat_partials), "conv3_wz: fused GroupNorm-backward statistics need the coefficients and a partial buffer");
    if (!a.bst_y && !a.add && conv3_wz32_enabled())      // the forward form: matrix waves on 32x32x16 MFMAs (conv3_wz32.hpp), fragments behind these
        return conv3_wz32_launch(a, static_cast<const char*>(wzfrag) + wz_frag_bytes(a.Cin, a.Cout), s);
    if (a.bst_y) return a.add ? wz_cfg<true, true>(a, wzfrag, s) : wz_cfg<true, false>(a, wzfrag, s);
    return a.add ? wz_cfg<false, true>(a, wzfrag, s) : wz_cfg<false, false>(a, wzfrag, s);
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
