#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for d in 0 8 6 1 2 4 3 7 14; do
  RU_WTZ_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wa$d -o a -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py wgrad bf16x3 4 16 128 6 3 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/wa$d/a_kernel_stats.csv')):
    if 'wgrad3_tz' in r['Name']: print("dbg=%-3s avg %.1f us  min %.1f us" % ("$d", float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
