#!/bin/bash
# SQ counters of conv3_sb2 for one build / shape: tools/pmc_conv.sh <lib .so or ""> C size   (separate --pmc passes, no tracing)
LIB=$1; C=${2:-32}; S=${3:-64}
[ -n "$LIB" ] && export RU_LIB_PATH=$LIB
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); rm -rf /tmp/pmcc$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcc$i -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_time.py $C $S 4 4 > /tmp/pmcc$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmcc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv3_sb2_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print("%-28s %.4e  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
