#!/bin/bash
# rocprofv3 kernel trace of the training step (bench.py --no-extras) -> gpurun_out/<tag>_{kernel_stats,sequence}.txt
# usage (GPU box): tools/profile_step.sh <tag> [steps]
tag=${1:-r02}; steps=${2:-6}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 2 --no-extras $BENCH_ARGS > /tmp/prof_$tag.log 2>&1
tail -1 /tmp/prof_$tag.log | cut -c1-200
st=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); tr=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $st $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_step_kernel_stats.txt
python3 $GRAFT_REPO_ROOT/tools/step_sequence.py $tr $steps $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_step_sequence.txt
head -45 $GRAFT_REPO_ROOT/gpurun_out/${tag}_train_step_kernel_stats.txt
