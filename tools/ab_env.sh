#!/bin/bash
# same-box interleaved A/B of an environment switch: tools/ab_env.sh "VAR=value" [rounds]
SET=$1; ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default      ms_per_step', d['ms_per_step'])"
  env $SET python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$SET  ms_per_step', d['ms_per_step'])"
done
