"""Time the 16->16 3x3x3 conv (bf16x3 and f32) for several extents: ns per output voxel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brats2019_amd import _lib as L
lib = L.load(); dev = torch.device("cuda")
def run(prec, n, c, d, h, w, it=10):
    x = torch.randn(n, c, d, h, w, device=dev); wt = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05; y = torch.empty_like(x)
    ws = L.workspace(lib.ru_conv3d_workspace_bytes(n, c, c, d, h, w, 3), dev)
    f = lambda: L.check(lib.ru_conv3d_fwd_p(L.f32(x), L.f32(wt), None, L.f32(y), n, c, c, d, h, w, 3, L.PRECISIONS[prec], L.ptr(ws), ws.numel(), L.stream()), "c")
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print("%-7s n=%d c=%d %dx%dx%d: %.3f ms  %.3f ns/voxel  %.1f TF alg" % (prec, n, c, d, h, w, ms, ms * 1e6 / (n * d * h * w), 2 * 27 * c * c * n * d * h * w / ms / 1e9), flush=True)
for prec in ("bf16x3", "f32"):
    for shp in ((4, 16, 128, 128, 128), (4, 16, 128, 128, 112), (4, 16, 120, 120, 128), (4, 16, 120, 136, 144), (4, 16, 132, 124, 128), (1, 16, 128, 128, 128), (8, 16, 128, 128, 128)):
        run(prec, *shp)
