#!/bin/bash
# Socket power / clock of GPU 0 under sustained loops of single kernels and of the whole step (rocm-smi sampled every ~0.15 s while the
# loop runs; the first second of samples is dropped).  usage (GPU box): tools/power_table.sh > gpurun_out/<tag>_power_table.txt
R=$GRAFT_REPO_ROOT
sample() {   # sample "<label>" <cmd...>  -> label, median power, median sclk, the workload's own last line
  label=$1; shift
  "$@" > /tmp/pt_work.log 2>&1 &
  wp=$!
  : > /tmp/pt_samples.txt
  while kill -0 $wp 2>/dev/null; do
    rocm-smi -d 0 --showpower --showclocks 2>/dev/null | python3 -c "
import sys, re
t = sys.stdin.read()
p = re.search(r'Power \(W\): ([0-9.]+)', t); c = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t)
if p and c: print(p.group(1), c.group(1))
" >> /tmp/pt_samples.txt
    sleep 0.1
  done
  wait $wp
  python3 - "$label" <<'PY'
import sys, statistics
rows = [tuple(map(float, l.split())) for l in open('/tmp/pt_samples.txt') if l.strip()]
rows = [r for r in rows if r[1] > 500][4:]            # busy samples only, warm-up dropped
last = [l for l in open('/tmp/pt_work.log').read().strip().splitlines() if l.strip()][-1][:110]
if rows:
    print("%-44s %5.0f W  %5.0f MHz  (%2d samples)  | %s" % (sys.argv[1], statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows), len(rows), last))
else:
    print("%-44s no busy samples | %s" % (sys.argv[1], last))
PY
}
echo "# socket power under sustained loops (tools/power_table.sh; rocm-smi, median of the busy samples; idle ~261 W, limit 1400 W)"
sample "3x3x3 conv 16->16, 4 x 128^3 (x3)"   python3 $R/tools/conv_time.py 16 128 4 12000
sample "3x3x3 conv 32->32, 4 x 64^3 (x3)"    python3 $R/tools/conv_time.py 32 64 4 24000
sample "3x3x3 conv 128->128, 4 x 16^3 (x3)"  python3 $R/tools/conv_time.py 128 16 4 80000
sample "training step, batch 4 (default)"    python3 $R/bench.py --steps 250 --warmup 5 --no-extras --probe-steps 0
sample "training step, bf16 gradients"       python3 $R/bench.py --steps 280 --warmup 5 --no-extras --probe-steps 0 --grad-precision bf16
sample "device copy 1 GiB (torch)"           python3 -c "
import torch, time
a = torch.empty(1 << 28, device='cuda'); b = torch.empty_like(a)
for _ in range(5): b.copy_(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 7000
for _ in range(n): b.copy_(a)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('copy: %.2f TB/s read+write' % (2 * a.numel() * 4 * n / dt / 1e12))
"
