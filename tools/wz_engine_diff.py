#!/usr/bin/env python3
"""Whole-network A/B of the Winograd-z kernel inside one process (RU_WZ is read per launch): probabilities, loss and every parameter
gradient with RU_WZ=1 against RU_WZ=0 on the same inputs.  usage: wz_engine_diff.py [N] [D] [H] [W] [fusion bits as 0/1 0/1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import resunet_oracle as O
from brats2019_amd import model as M, loss as L
n, d, h, w = [int(v) for v in (sys.argv[1:5] + ["2", "16", "128", "128"][len(sys.argv) - 1:])][:4]
fus = tuple(bool(int(v)) for v in sys.argv[5:7]) if len(sys.argv) > 6 else None
T = torch.from_numpy
res = {}
for tag in ("0", "1"):
    os.environ["RU_WZ"] = tag
    net = M.UNet(**O.DEFAULT_CFG)
    net.set_precision("bf16x3")
    net.load_state_dict({k: T(v) for k, v in O.make_params(3, **O.DEFAULT_CFG).items()})
    net.cuda().train()
    if fus is not None:
        net._get_engine().set_fusion(*fus)
    x = T(O.make_input(n, d, h, w, seed=3)).cuda()
    g = T(O.make_target(n, d, h, w, seed=3)).cuda()
    out = net([x])
    loss = L.FusedCriterion()(out, [g])
    loss.backward()
    res[tag] = (out[0].detach().clone(), float(loss), {k: q.grad.detach().clone() for k, q in net.named_parameters() if q.grad is not None})
os.environ.pop("RU_WZ", None)
pa, la, ga = res["1"]
pb, lb, gb = res["0"]
print("max |dp| %.3e  loss %.7f vs %.7f" % (float((pa - pb).abs().max()), la, lb))
for k in ga:
    rel = float((ga[k].double() - gb[k].double()).norm() / (gb[k].double().norm() + 1e-30))
    if rel > 1e-3:
        print("  %-44s rel L2 %.3e" % (k, rel))
print("done")
