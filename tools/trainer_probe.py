#!/usr/bin/env python3
"""The reference-surface training loop alone (bench.trainer_step_probe) for kernel-trace profiling / host-overhead checks:
usage trainer_probe.py [steps] ; prints wall ms per step of Trainer._train_one_epoch and of DataParallelStep on the same weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import parallel as P
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
be = P.HipBackend(precision="bf16x3")
flat = bench.init_params(be)
x, g = bench.synth(4, 128, 1000, torch.device("cuda"))
st = P.DataParallelStep(be, flat.clone())
for _ in range(3): st.step(x, g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): st.step(x, g)
torch.cuda.synchronize(); print("DataParallelStep: %.3f ms per step" % ((time.perf_counter() - t0) / steps * 1e3))
r = bench.trainer_step_probe(be, flat, x, g, steps, warmup=3)
print("Trainer loop     : %.3f ms per step" % r["ms_per_step"])
