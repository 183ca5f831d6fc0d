#!/bin/bash
# same-box interleaved A/B of an environment switch on the training step: tools/env_step_ab.sh VAR=VALUE [rounds]   (bench.py --no-extras, 20 steps)
R=${2:-3}
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $R); do
  env $1 python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$1] step %.3f ms' % d['ms_per_step'])"
  python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[default] step %.3f ms' % d['ms_per_step'])"
done
