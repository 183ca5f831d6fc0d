#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh <other .so> [bench args]   (interleaved runs; RU_LIB_PATH selects the build)
OTHER=${1:-$GRAFT_REPO_ROOT/brats2019_amd/lib/libresunet_hip_old.so}; shift
for i in 1 2; do
  echo "--- new"; python3 $GRAFT_REPO_ROOT/tools/conv_sweep.py 20 2>/dev/null | grep "C="
  echo "--- other ($OTHER)"; RU_LIB_PATH=$OTHER python3 $GRAFT_REPO_ROOT/tools/conv_sweep.py 20 2>/dev/null | grep "C="
done
for i in 1 2 3; do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new   ms_per_step', d['ms_per_step'])"
  RU_LIB_PATH=$OTHER python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('other ms_per_step', d['ms_per_step'])"
done
