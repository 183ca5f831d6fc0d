#!/bin/bash
# phase ablation of conv3_wz32_kernel (devtools builds: for d in 3 4 12; do python -m brats2019_amd.build --dbg $d; done) and per-kernel times of both
# matrix forms without the pack kernel (rocprofv3 kernel trace).  bits: 3 = staging waves idle (matrix waves alone), 4 = no MFMAs / fragment reads
# (staging + combine alone), 12 = staging alone
L=$GRAFT_REPO_ROOT/brats2019_amd/lib
cd /tmp && export TMPDIR=/tmp
for shape in "32 64" "64 32" "128 16"; do
  python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  for d in 3 4 12; do
    [ -f $L/libresunet_hip_dbg$d.so ] && RU_LIB_PATH=$L/libresunet_hip_dbg$d.so RU_SB2_DEBUG=$d python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 2>/dev/null
  done
  for m in 0 1; do
    rm -rf /tmp/rp_$m; RU_WZ32=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$m -o t -- python3 $GRAFT_REPO_ROOT/tools/conv_time.py $shape 4 20 > /dev/null 2>&1
    echo "RU_WZ32=$m kernel table:"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/rp_$m -name "*kernel_stats.csv" | head -1) 2>/dev/null | grep -v "^#" | head -4 | cut -c1-150
  done
done
