#!/bin/bash
# same-box A/B of several builds of the library: tools/ab2.sh <other1.so> [<other2.so> ...]   (interleaved; RU_LIB_PATH selects the build)
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  echo "--- new"; python3 $R/tools/conv_sweep.py 20 2>/dev/null | grep "C="
  for o in "$@"; do echo "--- other ($o)"; RU_LIB_PATH=$R/$o python3 $R/tools/conv_sweep.py 20 2>/dev/null | grep "C="; done
done
for i in 1 2 3; do
  python3 $R/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new   ms_per_step', d['ms_per_step'])"
  for o in "$@"; do RU_LIB_PATH=$R/$o python3 $R/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o ms_per_step', d['ms_per_step'])"; done
done
