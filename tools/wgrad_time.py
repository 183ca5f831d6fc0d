#!/usr/bin/env python3
"""One line: average time of the voxel-major split-bf16 3x3x3 weight gradient at one shape (HIP events).  usage: wgrad_time.py C size [N] [launches]
Honours RU_LIB_PATH (devtools builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import _lib as L
c, size = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = L.load()
dev = torch.device("cuda")
x = torch.randn(n, c // 16, size, size, size, 16, device=dev)
dy = torch.randn(n, c // 16, size, size, size, 16, device=dev)
dw = torch.empty(c, c, 3, 3, 3, device=dev)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
run = lambda: L.check(lib.ru_conv3d_bwd_weight_l(L.f32(x), L.f32(dy), L.f32(dw), n, c, c, size, size, size, 3, L.ptr(ws), ws.numel(), L.stream()), "wgrad")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches):
    run()
e1.record()
torch.cuda.synchronize()
print("wgrad C=%d %d^3 N=%d lib=%s: %.1f us" % (c, size, n, os.path.basename(os.environ.get("RU_LIB_PATH", "product")), e0.elapsed_time(e1) / launches * 1e3))
