#!/usr/bin/env python3
"""One line: average time of the voxel-major split-bf16 3x3x3 conv at one shape (HIP events).  usage: conv_time.py C size [N] [launches]
Honours RU_LIB_PATH / RU_SB2_DEBUG (devtools build) for ablations; RU_CONV_FLAGS = the `flags` of ru_conv3d_fwd_l (default 3: voxel-major in and out;
35 = + bit 5, the input is an activation tensor: conv3_mx_kernel where the shape has it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import _lib as L
c, size = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = L.load()
FLAGS = int(os.environ.get("RU_CONV_FLAGS", "3"))
dev = torch.device("cuda")
x = torch.randn(n, c // 16, size, size, size, 16, device=dev)
x = torch.where(x > 0, x, 0.01 * x)                 # an activation tensor
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
run = lambda: L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), None, L.f32(y), n, c, c, size, size, size, FLAGS, L.ptr(ws), ws.numel(), L.stream()), "conv")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches):
    run()
e1.record()
torch.cuda.synchronize()
print("C=%d %d^3 N=%d flags=%d dbg=%s: %.1f us" % (c, size, n, FLAGS, os.environ.get("RU_SB2_DEBUG", "0"), e0.elapsed_time(e1) / launches * 1e3))
