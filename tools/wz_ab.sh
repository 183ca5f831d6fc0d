#!/bin/bash
# same-box A/B of the Winograd-z conv kernel against the direct persistent kernel at the deep-level shapes (HIP events, 20 launches, incl. the pack kernel)
for rep in 1 2; do
for shape in "32 64" "64 32" "128 16"; do
  for wz in 0 1; do
    echo -n "RU_WZ=$wz  "; RU_WZ=$wz python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
done
done
