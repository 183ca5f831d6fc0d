#!/bin/bash
# everything profiles/ holds for a round, in one GPU call: tools/round_profiles.sh <tag>   (outputs under gpurun_out/, copy what is kept to profiles/)
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_stderr.txt
tail -c 600 gpurun_out/${tag}_bench_line.json; echo
bash tools/profile_step.sh $tag 13 > /dev/null 2>&1
RU_SIDE_STREAM=0 bash tools/profile_step.sh ${tag}_serial 13 > /dev/null 2>&1
bash tools/step_traffic.sh $tag > /dev/null 2>&1
tail -3 gpurun_out/${tag}_step_traffic.txt
bash tools/fwd_profile.sh $tag 1 bf16x3 > /dev/null 2>&1
bash tools/fwd_profile.sh $tag 1 f32 > /dev/null 2>&1
bash tools/fwd_profile.sh $tag 8 bf16x3 > /dev/null 2>&1
cp profiles/pmc_conv3_l0.json gpurun_out/pmc_conv3_l0.json
python3 tools/family_table.py --json gpurun_out/pmc_conv3_l0.json > gpurun_out/${tag}_family_table.txt 2>/dev/null
cat gpurun_out/${tag}_family_table.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/benchprof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/benchprof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-power > /tmp/benchprof.log 2>&1
st=$(find /tmp/benchprof -name "*kernel_stats.csv" | head -1); python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $st $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_default_kernel_stats.txt
head -12 $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_default_kernel_stats.txt
