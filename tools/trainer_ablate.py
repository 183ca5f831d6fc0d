import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, tempfile
from brats2019_amd import parallel as P, model as M, loss as LS, train as TR, metrics as MT
import bench
be = P.HipBackend(precision="bf16x3")
flat = bench.init_params(be)
x, g = bench.synth(4, 128, 1000, torch.device("cuda"))
net = M.UNet(**be.cfg); net.cuda()
with torch.no_grad():
    for (name, p), (_n, v) in zip(net.named_parameters(), be.engine.layout.views(flat).items()): p.copy_(v)
root = tempfile.mkdtemp()
tr = TR.Trainer(name="b", models_root=root, model=net, rewrite=True, connect_tb=False)
crit = [LS.Dice_loss_joint(index=0, priority=1), LS.BCE_Loss(index=0, bg_weight=1e-2)]
opt, sched = tr._make_optimizer(torch.optim.Adam, {"lr": 2e-5, "weight_decay": 1e-6, "amsgrad": True}, torch.optim.lr_scheduler.StepLR, {"step_size": 16000, "gamma": 0.5})
def timeit(label, metric, steps=20):
    res = {"Dice": []}
    run = lambda k: tr._train_one_epoch(crit, opt, [([x], [g])] * k, metric, res, 0, 0, sched)
    run(3); torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    print("%-40s %.3f ms/step  adam launches %s" % (label, (time.perf_counter() - t0) / steps * 1e3, getattr(opt, "last_launches", None)), flush=True)
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    pass
timeit("default (Dice metric)", [MT.Dice(name="Dice")])
timeit("no metric", [])
tr.fuse_criteria = False
timeit("no metric, criteria unfused", [])
tr.fuse_criteria = True
# host cost of one step without GPU work is not separable here; time the pieces on the host
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
tr._train_one_epoch(crit, opt, [([x], [g])] * 10, [], {"Dice": []}, 0, 0, sched); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout); st.sort_stats("cumulative").print_stats(18)
