// conv3_wz32.hip -- the Winograd-z convolution with its matrix waves on v_mfma_f32_32x32x16_bf16 (conv3_wz32.hpp): the kernel's own translation
// unit and its launch.  conv3_wz_launch (conv3_wz.hip) routes the forward form here (RU_WZ32=0: the 16x16x32 matrix form, same-box A/B).
#include "conv3_wz32.hpp"

namespace ru {

bool conv3_wz32_enabled() {
#ifdef RU_SB2_DBG
    const char* e = getenv("RU_WZ32");                  // devtools builds: RU_WZ32=0 selects the 16x16x32 matrix form (read per call)
    return !(e && *e == '0');
#else
    return true;                                        // the product library has this matrix form only
#endif
}

int conv3_wz32_launch(const Conv3Args& a, const void* wz32frag, hipStream_t s) {
    RU_REQUIRE(a.in_c16 && a.out_c16 && !a.bias && !a.sigmoid && !a.in_c4 && a.products != 1 && !a.bst_y && !a.add && conv3_wz_shape_ok(a.N, a.Cin, a.Cout, a.D, a.H, a.W),
               "conv3_wz32: the forward form -- voxel-major tensors, >= 32 input channels, whole 32-channel output blocks, an even depth, three products, no residual");
    RU_REQUIRE(a.N <= 32 || !a.stat_partials, "conv3_wz32: at most 32 samples per call when statistics are requested");
    RU_REQUIRE(!a.in_s16 || !a.in_scale, "conv3_wz32: a split-form input has no fused transform");
    static PerDevice attr_done;
    if (!attr_done.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wz32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WZ_LDS_BYTES);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv3_wz32)");
        attr_done.set();
    }
    static_assert(2 * 4 * 64 * 4 <= SB_STAT_LDS_FLOATS * 4 + 1024, "the statistics scratch of the 32-channel commit (two generations x four waves x 32 pairs) fits behind the M scratch");
    const int ntz = a.D / 2, nty = cdiv(a.H, 8), ntx = cdiv(a.W, 16);
    dim3 grid((unsigned)wz_grid_x(a.N, a.Cout, a.D, a.H, a.W), (unsigned)(a.Cout / 32));
    RU_REQUIRE(!a.fin.ticket || (a.stat_partials && a.fin.nblk == (int)grid.x && a.fin.N == a.N && a.fin.C == a.Cout && fin_tail_lds_bytes(a.fin) <= (size_t)WZ_LDS_BYTES),
               "conv3_wz32: tail descriptor does not match the launch");
    hipLaunchKernelGGL(conv3_wz32_kernel, grid, dim3(512), WZ_LDS_BYTES, s, a, (const u32x4*)wz32frag, ntz, nty, ntx, a.Cin / 16);
    RU_CHECK_LAUNCH("conv3_wz32_kernel");
    return RU_OK;
}

}  // namespace ru

#ifdef RU_SB2_DBG
// tools only (not in include/resunet_hip.h, -DRU_SB2_DBG builds): read and clear the section counters of devtools bit 128
extern "C" int ru_dbg_wz32_prof(unsigned long long* out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(ru::wz32_prof), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return ru::hip_fail(e, "hipMemcpyFromSymbol(wz32_prof)");
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(ru::wz32_prof), z, sizeof(z));
    return e == hipSuccess ? RU_OK : ru::hip_fail(e, "hipMemcpyToSymbol(wz32_prof)");
}
#endif
