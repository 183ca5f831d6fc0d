// conv3_sb2_mixed.hip -- the persistent split-bf16 3x3x3 convolution kernel (conv3_sb_common.hpp) with NCDHW tensors on at least one side:
// the head conv of the engine (voxel-major in, NCDHW out + bias + sigmoid) and the op-level C-ABI (ru_conv3d_fwd_p / _l).
#include "conv3_sb_common.hpp"

namespace ru {

int conv3_sb2_launch_mixed(const Conv3Args& a, hipStream_t s) {
    if (a.in_c16) return sb2_cfg<4, 8, true, false>(a, s);
    return a.out_c16 ? sb2_cfg<4, 8, false, true>(a, s) : sb2_cfg<4, 8, false, false>(a, s);
}

}  // namespace ru
