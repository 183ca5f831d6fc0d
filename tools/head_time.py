#!/usr/bin/env python3
"""Head convolution 16 -> 3 (voxel-major in, NCDHW out + bias) at N x size^3: the (row tap, output channel)-column form against the 16-column kernel
(RU_HEAD_FORM=0), HIP events over a loop of launches (each includes its ~5 us weight pack).  usage: head_time.py [size] [N] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import _lib as L
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = L.load()
dev = torch.device("cuda")
x = torch.randn(n, 1, size, size, size, 16, device=dev)
w = torch.randn(3, 16, 3, 3, 3, device=dev) * 0.05
b = torch.randn(3, device=dev)
y = torch.empty(n, 3, size, size, size, device=dev)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, 16, 3, size, size, size, 3), dev)
run = lambda: L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), L.f32(b), L.f32(y), n, 16, 3, size, size, size, 1, L.ptr(ws), ws.numel(), L.stream()), "conv")
for rep in range(2):
    for env in ("0", "1"):
        os.environ["RU_HEAD_FORM"] = env
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("16->3 %d x %d^3 RU_HEAD_FORM=%s: %.1f us" % (n, size, env, e0.elapsed_time(e1) / launches * 1e3))
