#!/usr/bin/env python3
"""Time the split-bf16 3x3x3 conv (voxel-major in/out, the engine's kernel) at the four level shapes of the training step with HIP
events on the launch stream, and check each against the exact-f32 kernel.  usage: conv_sweep.py [launches] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import _lib as L, ops

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = L.load()
dev = torch.device("cuda")
for c, size in ((16, 128), (32, 64), (64, 32), (128, 16)):
    x = torch.randn(n, c, size, size, size, device=dev)
    w = torch.randn(c, c, 3, 3, 3, device=dev) * (2.0 / (27 * c)) ** 0.5
    xc = ops.to_c16(x)
    y = torch.empty_like(xc)
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
    run = lambda: L.check(lib.ru_conv3d_fwd_l(L.f32(xc), L.f32(w), None, L.f32(y), n, c, c, size, size, size, 3, L.ptr(ws), ws.numel(), L.stream()), "conv")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / launches * 1e3
    ref = ops.conv3d(x, w, precision="f32")
    err = float((ops.from_c16(y) - ref).abs().max() / ref.abs().max())
    flops = 2.0 * 27 * c * c * n * size ** 3
    mfma_us = flops * 3 * 28 / 27 / 2.5e15 * 1e6
    hbm_us = 2.0 * c * n * size ** 3 * 4 / 8e12 * 1e6
    print("C=%3d %3d^3 N=%d: %7.1f us  (executed-MFMA floor %5.1f us -> %.2f; HBM floor %5.1f us -> %.2f)  rel err vs f32 %.1e" %
          (c, size, n, us, mfma_us, mfma_us / us, hbm_us, hbm_us / us, err))
