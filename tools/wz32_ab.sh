#!/bin/bash
# same-box A/B of the two matrix forms of the Winograd-z forward kernel (RU_WZ32=0: 16x16x32 MFMAs, conv3_wz.hpp; 1: 32x32x16, conv3_wz32.hpp) and the
# direct kernel (RU_WZ=0) at the deep-level shapes (HIP events, 20 launches)
for rep in 1 2; do
for shape in "32 64" "64 32" "128 16"; do
  echo -n "direct      "; RU_WZ=0 python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  for m in 0 1; do
    echo -n "RU_WZ32=$m  "; RU_WZ32=$m python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
done
