#!/bin/bash
# per-kernel same-box A/B of the training step: tools/profile_ab.sh <other .so>  -> prints kernel families (ms per step) for both builds
OTHER=${1:-$GRAFT_REPO_ROOT/brats2019_amd/lib/libresunet_hip_old.so}
cd /tmp && export TMPDIR=/tmp
for v in new other; do
  rm -rf /tmp/pab_$v
  if [ $v = other ]; then export RU_LIB_PATH=$OTHER; else unset RU_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab_$v -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-extras > /tmp/pab_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
def load(v):
    f = glob.glob('/tmp/pab_%s/**/*kernel_stats.csv' % v, recursive=True)[0]
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('void ', '').split('(')[0]
        d[n] = (int(r['Calls']), float(r['TotalDurationNs']) / 8e6)      # 8 steps (2 warm-up + 6)
    return d
a, b = load('new'), load('other')
keys = sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1]))
print("%-78s %6s %9s %9s" % ("kernel (ms per step)", "calls", "new", "other"))
for k in keys[:40]:
    print("%-78s %6d %9.3f %9.3f" % (k[:78], a.get(k, b.get(k))[0] // 8, a.get(k, (0, 0))[1], b.get(k, (0, 0))[1]))
print("%-78s %6s %9.3f %9.3f" % ("TOTAL", "", sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
PY
