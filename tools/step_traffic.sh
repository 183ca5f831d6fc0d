#!/bin/bash
# HBM traffic of ONE training step, per kernel: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with traces) over
# `bench.py --steps 2 --warmup 1 --no-extras --probe-steps 0` (3 identical steps), summed per kernel and divided by the steps.
# usage (GPU box): tools/step_traffic.sh <tag>   -> gpurun_out/<tag>_step_traffic.txt
tag=${1:-r03}
out=/tmp/traffic_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extras --probe-steps 0 > $out/$c.log 2>&1
  tail -1 $out/$c.log | cut -c1-160
done
python3 $GRAFT_REPO_ROOT/tools/step_traffic.py $out 3 $GRAFT_REPO_ROOT/gpurun_out/${tag}_step_traffic.txt
head -60 $GRAFT_REPO_ROOT/gpurun_out/${tag}_step_traffic.txt
