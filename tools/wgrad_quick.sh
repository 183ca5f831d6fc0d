#!/bin/bash
# usage: tools/wgrad_quick.sh "<layout flags list>" [C] [size]  -- time the L0 16->16 bf16x3 weight gradient (batch 4, 128^3)
cd /tmp && export TMPDIR=/tmp
for f in ${1:-0 3}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wq$f -o a -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py wgrad bf16x3 4 ${2:-16} ${3:-128} 6 $f > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/wq$f/a_kernel_stats.csv')):
    if 'wgrad' in r['Name']: print("flags=%-2s %s avg %.1f us  min %.1f us" % ("$f", r['Name'][:60], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
