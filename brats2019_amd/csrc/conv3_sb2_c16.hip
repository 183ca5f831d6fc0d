// conv3_sb2_c16.hip -- the persistent split-bf16 3x3x3 convolution kernel (conv3_sb_common.hpp) for voxel-major input AND output: the
// variants the whole-network engine runs (plain / residual add / fused GroupNorm-backward sums, one or several 16-channel input chunks).
#include "conv3_sb_common.hpp"

namespace ru {

int conv3_sb2_launch_c16(const Conv3Args& a, hipStream_t s) { return sb2_cfg<4, 8, true, true>(a, s); }

}  // namespace ru

#ifdef RU_SB2_DBG
// tools only (not in include/resunet_hip.h, -DRU_SB2_DBG builds): read and clear the RU_SB2_DEBUG=64 section counters
extern "C" int ru_dbg_sb2_prof(unsigned long long* out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(ru::sb2_prof), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return ru::hip_fail(e, "hipMemcpyFromSymbol(sb2_prof)");
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(ru::sb2_prof), z, sizeof(z));
    return e == hipSuccess ? RU_OK : ru::hip_fail(e, "hipMemcpyToSymbol(sb2_prof)");
}
#endif
