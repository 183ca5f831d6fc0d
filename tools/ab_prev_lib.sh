#!/bin/bash
# same-box A/B of the product library against lib/libresunet_hip_prev.so (a copy of an earlier build): training step and forward legs, interleaved
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "prev step: "; RU_LIB_PATH=$L/libresunet_hip_prev.so python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  echo -n "new  step: "; python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
for B in 8 4 1; do for i in 1 2; do
  RU_LIB_PATH=$L/libresunet_hip_prev.so python3 tools/fwd_probe.py $B bf16x3 40 2>/dev/null | tail -1 | sed "s/^/[prev] /"
  python3 tools/fwd_probe.py $B bf16x3 40 2>/dev/null | tail -1 | sed "s/^/[new]  /"
done; done
