import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from brats2019_amd import model as M
def rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g).cuda()
torch.manual_seed(3)
net = M.UNet(4, [1,2,2,4],[1,1,1,1],[16,32,64,128],3).cuda()
shapes = [(3,4,24,40,48), (2,4,32,32,32), (1,4,24,40,48), (1,4,32,40,48), (1,4,24,32,48), (1,4,24,40,32)]
if len(sys.argv) > 1 and sys.argv[1] == 'wide':      # shapes that reach the persistent kernels (z-walk order, fused statistics, sample boundaries)
    shapes = [(1,4,128,128,128), (2,4,64,64,64), (5,4,16,64,64), (8,4,16,128,128), (3,4,32,96,80), (2,4,8,256,256), (6,4,16,128,64)]
for shape in shapes:
    x = rand(*shape, seed=9); tgt = (rand(shape[0],3,*shape[2:], seed=10) > 0.3).float()
    res = {}
    for prec in ("f32","bf16x3"):
        net.set_precision(prec); net.zero_grad()
        p = net([x])[0]; ((p-tgt)**2).mean().backward()
        res[prec] = (p.detach().clone(), {n:q.grad.detach().clone() for n,q in net.named_parameters() if q.grad is not None})
    ga, gb = res["f32"][1], res["bf16x3"][1]
    errs = sorted(((float((ga[n]-gb[n]).norm()/(ga[n].norm()+1e-30)), n) for n in ga), reverse=True)
    print(shape, "dp %.1e" % float((res["f32"][0]-res["bf16x3"][0]).abs().max()), " | ".join("%s %.1e" % (n.replace("encoder_convs","enc").replace("decoder_convs","dec").replace(".weight",""), e) for e, n in errs[:6]))
