#!/bin/bash
# kernels of the reference-surface loop that are NOT the library's (ATen passes on the Trainer path): tools/trainer_kernels.sh [steps]
steps=${1:-10}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trprof -o t -- python3 $GRAFT_REPO_ROOT/tools/trainer_probe.py $steps > /tmp/trprof.log 2>&1
tail -2 /tmp/trprof.log
st=$(find /tmp/trprof -name "*kernel_stats.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $st /tmp/trprof_summary.txt
grep -v "^ru::" /tmp/trprof_summary.txt | head -30
grep "dice_counts\|crit_\|head_grad\|adam" /tmp/trprof_summary.txt
