#!/bin/bash
# usage: tools/sb2_quick.sh "<dbg list>" [layout flags] [channels] [size]
# quick timing of one bf16x3 conv shape (batch 4; default 16 channels at 128^3) for a few RU_SB2_DEBUG settings: tools/sb2_quick.sh "0 3 12" 3 32 64
cd /tmp && export TMPDIR=/tmp
for d in ${1:-0 3 12}; do
  RU_SB2_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$d -o a -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py fwd bf16x3 4 ${3:-16} ${4:-128} 6 ${2:-0} > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/abl$d/a_kernel_stats.csv')):
    if 'conv3_sb' in r['Name'] and 'pack' not in r['Name']: print("dbg=%-3s %s avg %.1f us  min %.1f us" % ("$d", r['Name'][:40], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
