"""Forward-only loop for kernel-trace profiling (weights frozen: packed once): usage fwd_probe.py [batch] [bf16x3|f32] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import parallel as P
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
it = int(sys.argv[3]) if len(sys.argv) > 3 else 20
be = P.HipBackend(precision=prec)
flat = bench.init_params(be)
be.engine.freeze_params(True)
x, _ = bench.synth(n, 128, 1, torch.device("cuda"))
for _ in range(3): be.forward(flat, x, training=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(it): be.forward(flat, x, training=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
print("fwd %s batch %d: %.3f ms wall per forward (%d + 3 forwards)" % (prec, n, dt * 1e3, it))
