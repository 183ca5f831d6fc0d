#!/bin/bash
# same-box A/B of fusion bits: tools/ab_fusion.sh <mask to switch off> [rounds]   (interleaved bench runs; RU_FUSION_OFF clears RU_FUSE_* bits)
MASK=${1:-8}; ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all on        ms_per_step', d['ms_per_step'])"
  RU_FUSION_OFF=$MASK python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off mask $MASK    ms_per_step', d['ms_per_step'])"
done
