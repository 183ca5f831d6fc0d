#!/bin/bash
# conv-probe A/B of several library builds, interleaved: tools/ab3.sh <name1> <name2> ...  (lib/libresunet_hip_<name>.so; "new" = the default)
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  for o in "$@"; do
    if [ $o = new ]; then unset RU_LIB_PATH; else export RU_LIB_PATH=$R/brats2019_amd/lib/libresunet_hip_$o.so; fi
    echo "--- $o: $(python3 $R/tools/conv_sweep.py 20 2>/dev/null | grep 'C=' | awk '{printf "%s/%s %s us   ", $2, $3, $5}')"
  done
done
