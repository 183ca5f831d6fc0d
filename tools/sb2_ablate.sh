#!/bin/bash
# time the L0 16->16 bf16x3 conv (batch 4, 128^3) with phases switched off (RU_SB2_DEBUG bits)
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3 4 8 12 7 15 16 32 48; do
  RU_SB2_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$d -o a -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py fwd bf16x3 4 16 128 6 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/abl$d/a_kernel_stats.csv')):
    if 'conv3_sb2' in r['Name']: print("dbg=%-3s avg %.1f us  min %.1f us" % ("$d", float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
