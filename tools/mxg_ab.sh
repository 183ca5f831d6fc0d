#!/bin/bash
# kernel time of the 16 -> 16 convolution at 4 x 128^3 in three forms, rocprofv3 --kernel-trace --stats (plain epilogue; the split conversion / packs are separate kernels):
#   flags 67 = gradient operand (conv3_mx_kernel<true,...>), 11 = split-form input, three bf16 products (conv3_sb2_kernel, what the data gradients run today), 35 = forward MX
cd /tmp && export TMPDIR=/tmp
for f in 67 11 35; do
  rm -rf /tmp/mxgp$f
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mxgp$f -o t -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py fwd bf16x3 4 16 128 12 $f > /tmp/mxgp$f.log 2>&1
  st=$(find /tmp/mxgp$f -name "*kernel_stats.csv" | head -1)
  echo "flags $f:"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $st /tmp/mxgp$f.txt > /dev/null 2>&1; grep -E "conv3_mx_kernel|conv3_sb2_kernel|mxg_split" /tmp/mxgp$f.txt | cut -c1-150
done
