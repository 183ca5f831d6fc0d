#!/bin/bash
# sample power / clocks of GPU 0 while a workload runs: tools/power_sample.sh <seconds> <cmd...>   (rocm-smi readings every ~0.2 s)
R=$GRAFT_REPO_ROOT
secs=$1; shift
"$@" > /tmp/ps_work.log 2>&1 &
wp=$!
for i in $(seq 1 $((secs * 4))); do
  rocm-smi -d 0 --showpower --showclocks --showtemp --json 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.load(sys.stdin); c = d[list(d.keys())[0]]
    keys = [k for k in c if any(t in k.lower() for t in ('power', 'sclk', 'mclk', 'junction', 'fclk'))]
    print(' | '.join('%s=%s' % (k.split('(')[0].strip()[-28:], c[k]) for k in keys))
except Exception as e:
    print('err', e)
"
  kill -0 $wp 2>/dev/null || break
  sleep 0.2
done
wait $wp
tail -3 /tmp/ps_work.log
