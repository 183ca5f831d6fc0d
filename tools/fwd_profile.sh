#!/bin/bash
# per-kernel profile of the forward pass alone: tools/fwd_profile.sh <tag> [batch] [precision] [iters]  -> gpurun_out/<tag>_fwd_b<batch>_kernel_stats.txt
tag=${1:-r03}; b=${2:-1}; prec=${3:-bf16x3}; it=${4:-20}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fwdprof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fwdprof_$tag -o t -- python3 $GRAFT_REPO_ROOT/tools/fwd_probe.py $b $prec $it > /tmp/fwdprof_$tag.log 2>&1
tail -2 /tmp/fwdprof_$tag.log
st=$(find /tmp/fwdprof_$tag -name "*kernel_stats.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $st $GRAFT_REPO_ROOT/gpurun_out/${tag}_fwd_b${b}_${prec}_kernel_stats.txt
head -40 $GRAFT_REPO_ROOT/gpurun_out/${tag}_fwd_b${b}_${prec}_kernel_stats.txt
