// conv3_sb2_c16_p1.hip -- the persistent 3x3x3 convolution kernel (conv3_sb_common.hpp), voxel-major in and out, with ONE MFMA product per
// operand pair (plain bf16 operands, fp32 accumulate): the data-gradient convolutions of a backward run with gradient precision
// RU_PREC_BF16 (ru_unet_set_grad_precision).  Same variants as conv3_sb2_c16.hip, compiled in their own unit so that the builds overlap.
#include "conv3_sb_common.hpp"

namespace ru {

int conv3_sb2_launch_c16_p1(const Conv3Args& a, hipStream_t s) { return sb2_cfg<4, 8, true, true, 1>(a, s); }

}  // namespace ru
