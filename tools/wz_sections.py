#!/usr/bin/env python3
"""Per-section cycle breakdown of conv3_wz_kernel (-DRU_SB2_DBG=128 build): usage wz_sections.py [C] [size] [N]
Average cycles per item of consumer wave 0 (setup / rows 0-4 / rows 5-9 / barrier) and of staging wave 0 (store + issue)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = 128 | int(os.environ.get("RU_SB2_EXTRA", "0"))
os.environ.setdefault("RU_LIB_PATH", os.path.join(ROOT, "brats2019_amd", "lib", "libresunet_hip_dbg%d.so" % DBG))
os.environ["RU_WZ"] = "1"
sys.path.insert(0, ROOT)
import torch
from brats2019_amd import _lib as L
c = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = L.load()
fn = lib.ru_dbg_wz_prof
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
x = torch.randn(n, c // 16, size, size, size, 16, device=dev)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, size, size, size, 3), dev)
buf = (ctypes.c_ulonglong * 8)()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.ru_conv3d_fwd_l(L.f32(x), L.f32(w), None, L.f32(y), n, c, c, size, size, size, 3, L.ptr(ws), ws.numel(), L.stream()), "fwd_l")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fn(ctypes.addressof(buf))
v = list(buf)
items, wgs = max(v[4], 1), max(v[6], 1)
tot = sum(v[:4])
print("C=%d size=%d N=%d dbg=%d: %d workgroups, %.1f items each; launch %.1f us; cycles per item (consumer wave 0):" % (c, size, n, DBG, wgs, items / wgs, ms * 1e3))
for i, nm in enumerate(["item setup", "rows 0-4", "rows 5-9", "barrier"]):
    print("  %-12s %8.0f  (%4.1f %%)" % (nm, v[i] / items, 100.0 * v[i] / tot))
print("  %-12s %8.0f ; ideal MFMA time per item 240 x 16 = 3840;  tail per workgroup %.0f;  staging wave 0: %.0f cycles per item in store + issue"
      % ("total", tot / items, v[5] / wgs, v[7] / items))
