#!/bin/bash
# How does conv3_sb2's time scale with the number of matrix operations per staged image?  (round 5, before building a reduced-product kernel)
# devtools builds 1048576 / 2097152 / 3145728 drop the last 1 / 2 / 3 fragment families per halo row (336 -> 288 / 216 / 144 MFMAs per item, with
# their LDS reads); staging unchanged.  Build first: for d in 1048576 2097152 3145728; do python -m brats2019_amd.build --dbg $d; done
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for rep in 1 2; do
for shape in "16 128" "32 64" "64 32" "128 16"; do
  python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  for d in 1048576 2097152 3145728; do
    [ -f $L/libresunet_hip_dbg$d.so ] && RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
done
