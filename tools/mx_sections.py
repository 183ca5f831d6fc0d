#!/usr/bin/env python3
"""Per-section cycle breakdown of the conv3_mx_kernel matrix-wave loop (-DRU_SB2_DBG=64 build): usage mx_sections.py [C] [size] [N]
Prints the average cycles per item that consumer wave 0 of a workgroup spends in each section."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = 64 | int(os.environ.get("RU_SB2_EXTRA", "0"))
os.environ.setdefault("RU_LIB_PATH", os.path.join(ROOT, "brats2019_amd", "lib", "libresunet_hip_dbg%d.so" % DBG))   # python -m brats2019_amd.build --dbg <64|extra>
sys.path.insert(0, ROOT)
import torch
from brats2019_amd import _lib as L

c = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = L.load()
fn = lib.ru_dbg_mx_prof
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
x = torch.randn(n, c // 16, size, size, size, 16, device=dev)
x = torch.where(x > 0, x, 0.01 * x)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
buf = (ctypes.c_ulonglong * 8)()
ms = 0.0
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), None, L.f32(y), n, c, c, size, size, size, 35, L.ptr(ws), ws.numel(), L.stream()), "fwd_l")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)                 # (includes the ~2 us weight-pack kernel)
    fn(ctypes.addressof(buf))
v = list(buf)
items, wgs = max(v[5], 1), max(v[7], 1)
names = ["item setup", "rows 0-4", "(unused)", "rows 5-9", "barrier"]
tot = sum(v[:5])
print("C=%d size=%d N=%d: %d workgroups, %.1f items each; cycles per item (consumer wave 0):" % (c, size, n, wgs, items / wgs))
for i, nm in enumerate(names):
    print("  %-12s %8.0f  (%4.1f %%)" % (nm, v[i] / items, 100.0 * v[i] / tot))
print("  %-12s %8.0f ; ideal MFMA time per item 112 x 16 + 56 x 32 = 3584" % ("total", tot / items))
per_wg = items / wgs
print("  launch %.1f us for %.1f items per workgroup = %.2f us per item -> the consumer's %0.f cycles per item are %.2f GHz (s_memtime counts shader cycles)"
      % (ms * 1e3, per_wg, ms * 1e3 / per_wg, tot / items, (tot / items) / (ms * 1e3 / per_wg) / 1e3))
