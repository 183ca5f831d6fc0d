import os, sys
sys.path.insert(0, "/root/repo")
import torch
from oracle import resunet_oracle as O
from brats2019_amd import model as M
T = torch.from_numpy
net = M.UNet(**O.DEFAULT_CFG); net.set_precision("bf16x3")
net.load_state_dict({k: T(v) for k, v in O.make_params(2024, **O.DEFAULT_CFG).items()}); net.cuda()
x = T(O.make_input(2, 128, 128, 128, seed=2024)).cuda()
for seed in (1, 2, 3, 4, 5):
  w = torch.randn(2, 3, 128, 128, 128, generator=torch.Generator().manual_seed(seed)).cuda() * 1e-3
  res = {}
  for fusion in ((True, True), (False, False)):
    net._get_engine().set_fusion(*fusion)
    net.zero_grad()
    p = net([x])[0]
    (p * w).sum().backward()
    res[fusion] = {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None}
  gb = res[(False, False)]
  for f in ((True, True),):
    ga = res[f]
    rel = sorted(((float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-30)), k) for k in ga), reverse=True)
    print(seed, f, ["%.1e %s" % r for r in rel[:3]])
