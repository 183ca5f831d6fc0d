#!/bin/bash
# phase ablation of conv3_mx_kernel (16 -> 16, 4 x 128^3), HIP-event timing; devtools builds: for d in 64 3 4 12; do python -m brats2019_amd.build --dbg $d; done
# bits: 1 staging waves skip convert + LDS stores, 2 skip their loads, 4 matrix waves skip the MFMAs, 8 skip the row stores
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for r in 1 2; do
  RU_CONV_FLAGS=35 python3 $GRAFT_REPO_ROOT/tools/conv_time.py 16 128 4 20 2>/dev/null
  for d in 3 4 12; do
    [ -f $L/libresunet_hip_dbg$d.so ] && RU_CONV_FLAGS=35 RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py 16 128 4 20 2>/dev/null
  done
  RU_CONV_FLAGS=3 python3 $GRAFT_REPO_ROOT/tools/conv_time.py 16 128 4 20 2>/dev/null
  for d in 3 4 12; do
    [ -f $L/libresunet_hip_dbg$d.so ] && RU_CONV_FLAGS=3 RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py 16 128 4 20 2>/dev/null
  done
done
python3 $GRAFT_REPO_ROOT/tools/mx_sections.py 16 128 4
python3 $GRAFT_REPO_ROOT/tools/sb2_sections.py 16 128 4
