#!/bin/bash
# same-box A/B of conv3_mx_kernel (fp16 + MX-fp8 products) against the three-product kernel: tools/mx_ab.sh [rounds]
#   op-level 16 -> 16 conv at 4 x 128^3 and 1 x 128^3 (HIP events, 20 launches, incl. the pack launch), the training step (bench.py --no-extras), the forward legs
R=${1:-3}
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $R); do
  RU_CONV_FLAGS=3 python3 tools/conv_time.py 16 128 4 20 2>/dev/null | tail -1
  RU_CONV_FLAGS=35 python3 tools/conv_time.py 16 128 4 20 2>/dev/null | tail -1
done
RU_CONV_FLAGS=3 python3 tools/conv_time.py 16 128 1 40 2>/dev/null | tail -1
RU_CONV_FLAGS=35 python3 tools/conv_time.py 16 128 1 40 2>/dev/null | tail -1
for i in $(seq 1 $R); do
  RU_MX=0 python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[RU_MX=0] step %.3f ms' % d['ms_per_step'])"
  python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[default] step %.3f ms' % d['ms_per_step'])"
done
bash tools/fwd_ab_env.sh RU_MX=0 1
bash tools/fwd_ab_env.sh RU_MX=0 4
