#!/bin/bash
# phase ablation of conv3_sb2 at three shapes, HIP-event timing.  The switches are COMPILE-TIME: build the variants first (in the build
# container: for d in 3 11 131 139 8 136; do python -m brats2019_amd.build --dbg $d; done), then on the GPU box: tools/sb2_ablate2.sh
# bits: 1 producers skip transform/split/LDS store, 2 producers skip global loads, 4 consumers skip MFMAs (+ their LDS reads),
#       8 consumers drop the epilogue rows (MFMAs kept), 128 consumers never refill the weights (multi-chunk shapes)
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for shape in "16 128" "32 64" "128 16"; do
  RU_SB2_DEBUG=0 python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  for d in 3 11 131 139 8 136 12 7; do
    [ -f $L/libresunet_hip_dbg$d.so ] && RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
