import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from brats2019_amd import model as M
prec = sys.argv[1]
torch.manual_seed(11)
net = M.UNet(4, [1, 2, 2, 4], [1, 1, 1, 1], [16, 32, 64, 128], 3).cuda()
net.set_precision(prec)
g = torch.Generator().manual_seed(5)
x = torch.randn(4, 4, 128, 128, 128, generator=g).cuda()
w = torch.randn(4, 3, 128, 128, 128, generator=g).cuda() * 1e-3
def run(xs, ws):
    net.zero_grad()
    p = net([xs])[0]
    (p * ws).sum().backward()
    return p.detach().clone(), {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None}
pb, gb = run(x, w)
pb2, gb2 = run(x, w)
print(prec, "rerun identical:", all(torch.equal(gb[k], gb2[k]) for k in gb), float((pb - pb2).abs().max()))
gsum = None; dp = 0
for n in range(4):
    pn, gn = run(x[n:n + 1].contiguous(), w[n:n + 1].contiguous())
    dp = max(dp, float((pn[0] - pb[n]).abs().max()))
    gsum = gn if gsum is None else {k: gsum[k] + gn[k] for k in gsum}
rel = sorted(((float((gb[k] - gsum[k]).norm() / (gsum[k].norm() + 1e-30)), k) for k in gb), reverse=True)
print(prec, os.environ.get("RU_NO_BST"), "dp %.2e" % dp, "worst", ["%.1e %s" % r for r in rel[:4]], "median %.1e" % rel[len(rel)//2][0])
