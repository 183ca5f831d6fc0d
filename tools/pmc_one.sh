#!/bin/bash
# usage: pmc_one.sh "<counters>" <kernel substring> <probe args...>  -- one rocprofv3 --pmc pass, prints per-launch means
set_=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1
rocprofv3 --pmc $set_ --output-format csv -d /tmp/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" > /tmp/pmc1.log 2>&1
python3 - "$kern" <<'PY'
import csv, glob, sys, collections
kern = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print("%-28s %.4e  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
