"""Data-parallel step on 2 CPU processes over gloo: the host logic of brats2019_amd.parallel.DataParallelStep
(shard ownership, all-reduce of the criterion sums BEFORE the gradient, SUM all-reduce of the flat gradient
bucket, Adam on live segments only) with the arithmetic supplied by an oracle-backed CPU backend (test
infrastructure: the product backend is HipBackend and is HIP-only).

Checks, SURVEY 8(e): the 2-rank sharded step equals the reference's global-batch step (what nn.DataParallel
computes on GPU 0, main.py:61 / train.py:201-210); the naive recipe (per-rank Dice, mean of gradients) does not."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import resunet_oracle as O

CFG = dict(depth=3, encoder_layers=[1, 1, 1], decoder_layers=[1, 1, 1], number_of_channels=[8, 16, 16], number_of_outputs=3)
DHW = (8, 8, 8)
SEED = 5


class OracleBackend:
    """Same interface as parallel.HipBackend, arithmetic by the CPU oracle (float32 forward/backward via autograd,
    float64 closed forms for the criterion and Adam)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.spec = O.state_dict_spec(**cfg)
        self.offsets, off = {}, 0
        for k, shape in self.spec:
            n = int(np.prod(shape))
            self.offsets[k] = (off, n, tuple(shape))
            off += n
        self.total = off
        d = cfg["depth"] - 1
        self.dead = {k for k, _ in self.spec if k.startswith("decoder_convs.%d." % d) or k.startswith("decoder_convs1x1.%d." % d)}
        self.live_segments = []
        for k, _ in self.spec:
            o, n, _s = self.offsets[k]
            if k in self.dead:
                continue
            if self.live_segments and self.live_segments[-1][1] == o:
                self.live_segments[-1][1] = o + n
            else:
                self.live_segments.append([o, o + n])

    def unflatten(self, flat, requires_grad=False):
        return {k: flat[o:o + n].view(s).clone().requires_grad_(requires_grad) for k, (o, n, s) in self.offsets.items()}

    def forward(self, flat, x, training=True):
        self._p = self.unflatten(flat, True)
        self._probs = O.unet_forward(self._p, x, **self.cfg)
        return self._probs.detach()

    def criterion_sums(self, probs, target, bg_weight):
        i, u, b = O.np_dice_bce_sums(probs.numpy(), target.numpy(), bg_weight)
        return torch.from_numpy(np.concatenate([i, u, [b]]))

    def criterion_grad(self, probs, target, sums, count, bg_weight, priority):
        s = sums.numpy()
        c = (s.size - 1) // 2
        return torch.from_numpy(O.np_criterion_grad(probs.numpy(), target.numpy(), s[:c], s[c:2 * c], count, bg_weight, priority).astype(np.float32))

    def backward(self, flat, dprobs, grads):
        self._probs.backward(dprobs)
        grads.zero_()
        for k, (o, n, _s) in self.offsets.items():
            if self._p[k].grad is not None:
                grads[o:o + n] = self._p[k].grad.reshape(-1)
        return grads

    def adam(self, flat, grads, m, v, vmax, step, lr, betas, eps, weight_decay):
        for a, b in self.live_segments:
            w2, m2, v2, vm2 = O.np_adam_amsgrad_step(flat[a:b].double().numpy(), grads[a:b].double().numpy(), m[a:b].double().numpy(),
                                                     v[a:b].double().numpy(), vmax[a:b].double().numpy(), step, lr, betas[0], betas[1], eps, weight_decay)
            flat[a:b] = torch.from_numpy(w2).float(); m[a:b] = torch.from_numpy(m2).float()
            v[a:b] = torch.from_numpy(v2).float(); vmax[a:b] = torch.from_numpy(vm2).float()


def flat_params(backend, seed):
    params = O.make_params(seed, **backend.cfg)
    flat = torch.zeros(backend.total)
    for k, (o, n, _s) in backend.offsets.items():
        flat[o:o + n] = torch.from_numpy(params[k]).reshape(-1)
    return flat, params


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brats2019_amd.parallel import DataParallelStep
    backend = OracleBackend(CFG)
    flat, _ = flat_params(backend, SEED)
    x = torch.from_numpy(O.make_input(world, *DHW, seed=SEED))
    g = torch.from_numpy(O.make_target(world, *DHW, seed=SEED))
    sl = DataParallelStep.shard(world, rank, world)
    step = DataParallelStep(backend, flat, lr=1e-3, step_size=1, gamma=0.5)
    loss, dice, bce = step.loss_and_grads(x[sl], g[sl])
    grads1 = step.grads.clone()
    w0 = flat.clone()
    # two optimizer steps exercise Adam state + StepLR(step_size=1): lr 1e-3 then 5e-4
    step2 = DataParallelStep(backend, flat, lr=1e-3, step_size=1, gamma=0.5)
    l1 = step2.step(x[sl], g[sl])[0]
    w1 = flat.clone()
    assert step2.global_step == 1
    l2 = step2.step(x[sl], g[sl])[0]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss=float(loss), dice=float(dice), bce=float(bce), grads=grads1.numpy(),
             w0=w0.numpy(), w1=w1.numpy(), w2=flat.numpy(), l1=float(l1), l2=float(l2))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_rank_sharded_step_equals_global_batch_step(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (dict(np.load(tmp_path / ("rank%d.npz" % r))) for r in range(world))
    # every rank ends with the same loss, gradient bucket and weights
    assert r0["loss"] == r1["loss"] and np.array_equal(r0["grads"], r1["grads"]) and np.array_equal(r0["w2"], r1["w2"])
    # == the reference's global-batch step
    backend = OracleBackend(CFG)
    _, params = flat_params(backend, SEED)
    x, g = O.make_input(world, *DHW, seed=SEED), O.make_target(world, *DHW, seed=SEED)
    _probs, ref_loss, ref_grads = O.forward_backward(params, x, g, **CFG)
    assert abs(float(r0["loss"]) - ref_loss) < 2e-6
    for k, (o, n, s) in backend.offsets.items():
        got = r0["grads"][o:o + n].reshape(s)
        if ref_grads[k] is None:
            assert not got.any(), k                           # dead parameters: zeros in the bucket
            continue
        assert np.abs(got - ref_grads[k]).max() <= 2e-5 * np.abs(ref_grads[k]).max() + 1e-9, k
    # the naive DDP recipe (per-rank Dice, MEAN of gradients) is a different gradient
    naive = None
    for r in range(world):
        _p, _l, gr = O.forward_backward(params, x[r:r + 1], g[r:r + 1], **CFG)
        vec = np.concatenate([(gr[k] if gr[k] is not None else np.zeros(s, np.float32)).ravel() for k, (o, n, s) in backend.offsets.items()])
        naive = vec / world if naive is None else naive + vec / world
    rel = np.abs(naive - r0["grads"]).max() / np.abs(r0["grads"]).max()
    assert rel > 1e-3, "naive recipe unexpectedly matches (rel %.2e)" % rel
    # optimizer: dead segments untouched, live segments moved, second step used the halved lr
    dead_mask = np.ones(backend.total, bool)
    for a, b in backend.live_segments:
        dead_mask[a:b] = False
    assert np.array_equal(r0["w2"][dead_mask], r0["w0"][dead_mask])
    assert (r0["w2"][~dead_mask] != r0["w0"][~dead_mask]).mean() > 0.99
    # replay the first Adam step (lr 1e-3, wd 1e-6, amsgrad) from the verified global gradient with the closed form
    w = r0["w0"].astype(np.float64)
    w1, _m, _v, _vm = O.np_adam_amsgrad_step(w, r0["grads"].astype(np.float64), np.zeros_like(w), np.zeros_like(w), np.zeros_like(w), 1, 1e-3)
    assert np.abs(w1 - r0["w1"])[~dead_mask].max() < 1e-7
    assert abs(float(r0["l1"]) - ref_loss) < 2e-6 and np.isfinite(r0["l2"])
