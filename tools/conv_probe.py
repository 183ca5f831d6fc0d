#!/usr/bin/env python3
"""Launch one convolution entry point a few times (for rocprofv3 --pmc / --kernel-trace passes).
usage: conv_probe.py [fwd|wgrad] [f32|bf16x3] [N] [C] [size] [launches] [layout flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import _lib as L

kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
c = int(sys.argv[4]) if len(sys.argv) > 4 else 16
size = int(sys.argv[5]) if len(sys.argv) > 5 else 128
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 5
flags = int(sys.argv[7]) if len(sys.argv) > 7 else 0          # fwd only: bit 0 input C16, bit 1 output C16, ... (include/resunet_hip.h; 67 = gradient-operand MX form)
lib = L.load()
dev = torch.device("cuda")
x = torch.randn(n, c, size, size, size, device=dev)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
dw = torch.empty_like(w)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3) + (n * size ** 3 * 64 + 65536 if flags & 64 else 0), dev)
for _ in range(launches):
    if kind == "fwd" and flags:
        L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), None, L.f32(y), n, c, c, size, size, size, flags, L.ptr(ws), ws.numel(), L.stream()), "fwd_l")
    elif kind == "fwd":
        L.check(lib.ru_conv3d_fwd_p(L.f32(x), L.f32(w), None, L.f32(y), n, c, c, size, size, size, 3, L.PRECISIONS[prec], L.ptr(ws), ws.numel(), L.stream()), "fwd")
    elif flags:
        L.check(lib.ru_conv3d_bwd_weight_l(L.f32(x), L.f32(y), L.f32(dw), n, c, c, size, size, size, flags, L.ptr(ws), ws.numel(), L.stream()), "wgrad_l")
    elif prec == "bf16x3":
        L.check(lib.ru_conv3d_bwd_weight_p(L.f32(x), L.f32(y), L.f32(dw), None, n, c, c, size, size, size, 3, L.PRECISIONS[prec], L.ptr(ws), ws.numel(), L.stream()), "wgrad")
    else:
        L.check(lib.ru_conv3d_bwd_weight(L.f32(x), L.f32(y), L.f32(dw), None, n, c, c, size, size, size, 3, L.ptr(ws), ws.numel(), L.stream()), "wgrad")
torch.cuda.synchronize()
print("ok", kind, prec, n, c, size, launches)
