#!/bin/bash
# same-box A/B of library builds on the conv sweep: tools/ab_libs.sh <lib suffix> [<lib suffix> ...]   ("" = product build)
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
for rep in 1 2; do
  echo "--- product"; python3 $GRAFT_REPO_ROOT/tools/conv_sweep.py 20 2>/dev/null | grep "C="
  for sfx in "$@"; do
    echo "--- $sfx"; RU_LIB_PATH=$L/libresunet_hip_$sfx.so python3 $GRAFT_REPO_ROOT/tools/conv_sweep.py 20 2>/dev/null | grep "C="
  done
done
