#!/usr/bin/env python3
"""Time the split-bf16 3x3x3 conv on SPLIT-FORM voxel-major input (what the data-gradient convs read) at the deep-level shapes and check it
against the same conv on the fp32 form of the input.  usage: conv_sweep_split.py [launches] [N]"""
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
    xs = ops.to_split_c16(xc)
    y = torch.empty_like(xc)
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
    run = lambda: L.check(lib.ru_conv3d_fwd_l(L.f32(xs), L.f32(w), None, L.f32(y), n, c, c, size, size, size, 1 | 2 | 8, L.ptr(ws), ws.numel(), L.stream()), "conv")
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
    ref = ops.conv3d_layout(xc, w, in_c16=True, out_c16=True)
    err = float((y - ref).abs().max() / ref.abs().max())
    print("split input C=%3d %3d^3 N=%d: %7.1f us   max rel diff vs the fp32-form input %.1e" % (c, size, n, us, err))
