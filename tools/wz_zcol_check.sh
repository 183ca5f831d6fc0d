cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_c16.py -q -x -k "winograd" 2>&1 | grep -E "passed|failed"
python -m pytest tests/test_hip_unet.py -q -x -k "128 or forward_conv" 2>&1 | grep -E "passed|failed"
for shape in "32 64" "64 32" "128 16"; do for r in 1 2; do RU_CONV_FLAGS=35 python3 tools/conv_time.py $shape 4 20 2>/dev/null | tail -1; done; done
cd /tmp; export TMPDIR=/tmp
for c in "32 64" "64 32" "128 16"; do
for cnt in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pm; rocprofv3 --pmc $cnt --output-format csv -d /tmp/pm -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py fwd bf16x3 4 $c 6 35 > /dev/null 2>&1; python3 - <<PY
import csv,glob
rows=[r for f in glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'wz32mx' in r.get('Kernel_Name','')]
v=[float(r['Counter_Value']) for r in rows]
print("$c $cnt mean KB", sum(v)/max(1,len(v)), "GB(x2 if fetch)", (2 if "$cnt"=="FETCH_SIZE" else 1)*sum(v)/max(1,len(v))*1024/1e9)
PY
done; done
