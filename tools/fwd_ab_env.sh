#!/bin/bash
# same-box A/B of an environment switch on the forward legs: tools/fwd_ab_env.sh VAR=VALUE [batch] [precision]   (e.g. RU_FUSION_OFF=64)
# prints ms per forward (tools/fwd_probe.py) alternating off / on, three times
B=${2:-1}; P=${3:-bf16x3}
for i in 1 2 3; do
  env $1 python3 $GRAFT_REPO_ROOT/tools/fwd_probe.py $B $P 40 2>/dev/null | tail -1 | sed "s/^/[$1] /"
  python3 $GRAFT_REPO_ROOT/tools/fwd_probe.py $B $P 40 2>/dev/null | tail -1 | sed "s/^/[default] /"
done
