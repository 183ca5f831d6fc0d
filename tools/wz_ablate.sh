#!/bin/bash
# phase ablation of conv3_wz_kernel (devtools builds: for d in 1 3 4 8 12; do python -m brats2019_amd.build --dbg $d; done)
# bits: 1 staging waves skip transform / split / LDS stores, 2 skip their global loads, 4 matrix waves skip MFMAs + fragment reads, 8 skip the combine
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for shape in "32 64" "64 32" "128 16"; do
  RU_WZ=1 python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  for d in 1 3 4 8 12; do
    [ -f $L/libresunet_hip_dbg$d.so ] && RU_WZ=1 RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
