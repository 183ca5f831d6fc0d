#!/usr/bin/env python3
"""Per-section cycle breakdown of conv3_wz32_kernel (-DRU_SB2_DBG=128 build, RU_SB2_EXTRA adds bits): usage wz32_sections.py [C] [size] [N]
Average cycles per item of matrix wave 0 (setup / fragment steps 0-8 / 9-17 / 18-26 / barrier).  RU_SECT_MX=1: conv3_wz32mx_kernel (fp16 + MX-fp8 products)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = 128 | int(os.environ.get("RU_SB2_EXTRA", "0"))
os.environ.setdefault("RU_LIB_PATH", os.path.join(ROOT, "brats2019_amd", "lib", "libresunet_hip_dbg%d.so" % DBG))
sys.path.insert(0, ROOT)
import torch
from brats2019_amd import _lib as L
c = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = L.load()
MX = os.environ.get("RU_SECT_MX", "0") == "1"
fn = lib.ru_dbg_wz32mx_prof if MX else lib.ru_dbg_wz32_prof
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
x = torch.randn(n, c // 16, size, size, size, 16, device=dev)
x = torch.where(x > 0, x, 0.01 * x)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
buf = (ctypes.c_ulonglong * 8)()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), None, L.f32(y), n, c, c, size, size, size, 35 if MX else 3, L.ptr(ws), ws.numel(), L.stream()), "fwd_l")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fn(ctypes.addressof(buf))
v = list(buf)
items, wgs = max(v[5], 1), max(v[7], 1)
tot = sum(v[:5])
print("C=%d size=%d N=%d dbg=%d: %d workgroups, %.1f items each; launch (with the pack) %.1f us; cycles per item (matrix wave 0):" % (c, size, n, DBG, wgs, items / wgs, ms * 1e3))
for i, nm in enumerate(["item setup", "steps 0-8", "steps 9-17", "steps 18-26", "barrier"]):
    print("  %-12s %8.0f  (%4.1f %%)" % (nm, v[i] / items, 100.0 * v[i] / tot))
print("  %-12s %8.0f ; 108 MFMAs x 32 cycles = 3456;  tail per workgroup %.0f cycles" % ("total (mx: 36 x 32 + 20 x 64 = 2432)" if MX else "total", tot / items, v[6] / wgs))
