"""Forward-only (batch 1) loop for kernel-trace profiling: GPU busy time vs wall."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import parallel as P
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
be = P.HipBackend(precision=prec)
flat = bench.init_params(be)
x, _ = bench.synth(n, 128, 1, torch.device("cuda"))
for _ in range(3): be.forward(flat, x, training=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): be.forward(flat, x, training=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("fwd %s batch %d: %.3f ms wall" % (prec, n, dt * 1e3))
