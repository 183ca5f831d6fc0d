"""Which torch CPU thread count runs the oracle fastest on this host?  (bench.py cpu_baseline sizing)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import resunet_oracle as O
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = O.to_torch(O.make_params(1337, **O.DEFAULT_CFG))
x = torch.from_numpy(O.make_input(1, size, size, size))
for th in (8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1):
        break
    torch.set_num_threads(th)
    with torch.no_grad():
        O.unet_forward(p, x, **O.DEFAULT_CFG)
        t0 = time.perf_counter(); O.unet_forward(p, x, **O.DEFAULT_CFG); dt = time.perf_counter() - t0
    print("threads %3d  fwd %d^3: %.3f s" % (th, size, dt), flush=True)
