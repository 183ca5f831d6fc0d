#!/bin/bash
# HBM/L2 traffic counters for one probe config (separate passes, pmc only)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$out/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'pack' in k or 'at::' in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: "%.4g" % (sum(v)/len(v)) for c, v in d.items()})
PY
