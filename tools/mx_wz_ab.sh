#!/bin/bash
# same-box A/B of conv3_wz32mx_kernel (Winograd-z forward convs with fp16 + MX-fp8 products) against conv3_wz32_kernel: op-level at the three deep shapes, the
# training step and the forward legs with RU_MX=1 (16-channel MX kernel only) against the default
cd $GRAFT_REPO_ROOT
for shape in "32 64" "64 32" "128 16"; do
  for r in 1 2; do
    RU_CONV_FLAGS=3 python3 tools/conv_time.py $shape 4 20 2>/dev/null | tail -1
    RU_CONV_FLAGS=35 python3 tools/conv_time.py $shape 4 20 2>/dev/null | tail -1
  done
done
bash tools/env_step_ab.sh RU_MX=1 3
bash tools/fwd_ab_env.sh RU_MX=1 4
bash tools/fwd_ab_env.sh RU_MX=1 1
