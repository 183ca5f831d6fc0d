"""Eager launches against hipGraph replay of forward + criterion + backward (DataParallelStep.loss_and_grads), batch 4 x 128^3, one process, interleaved.
   python tools/graph_step_probe.py [steps]   -- answers whether the ~260 launches of a step leave host-side or dispatch gaps a graph would close."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import resunet_oracle as O          # tools only: synthetic inputs / parameters
from brats2019_amd import parallel as P

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = torch.from_numpy
be = P.HipBackend(cfg=O.DEFAULT_CFG)
flat = be.new_flat()
for k, v in be.engine.layout.views(flat).items():
    v.copy_(T(O.make_params(3, **O.DEFAULT_CFG)[k]))
st = P.DataParallelStep(be, flat)
x = T(O.make_input(4, 128, 128, 128, seed=3)).cuda()
g = T(O.make_target(4, 128, 128, 128, seed=3)).cuda()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        st.loss_and_grads(x, g)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    st.loss_and_grads(x, g)
torch.cuda.synchronize()

def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for r in range(3):
    e = timed(lambda: st.loss_and_grads(x, g))
    q = timed(graph.replay)
    print("eager %.3f ms   graph replay %.3f ms per forward + criterion + backward (%d steps)" % (e, q, steps))
