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

    def criterion_losses(self, sums, count, priority):
        c = (sums.numel() - 1) // 2
        dice = priority * (1.0 - torch.mean(2.0 * (sums[:c] + 1e-6) / (sums[c:2 * c] + 2e-6)))     # loss.py:114-122
        bce = -sums[2 * c] / count                                                                # loss.py:79
        return 0.5 * (dice + bce), dice, bce

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


# ---------------------------------------------------------------------- the reference-surface loop: train.Trainer under a process group
class _OracleUNet(torch.nn.Module):
    """CPU stand-in with the product model's call convention (list in, list out) over the oracle forward: test infrastructure for
    the HOST logic of Trainer (sharding, gradient all-reduce, rank-0 checkpoints); the product model is HIP-only."""

    def __init__(self, cfg, params):
        super().__init__()
        self.cfg = cfg
        self.names = list(params.keys())
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(params[k]).clone()) for k in self.names])

    def cuda(self, *a, **k):
        return self

    def forward(self, x):
        return [O.unet_forward(dict(zip(self.names, self.plist)), x[0], **self.cfg)]


class _OracleCritFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, g, w_dice, w_bce, bg_weight, data_parallel):
        i, u, b = O.np_dice_bce_sums(p.detach().numpy(), g.numpy(), bg_weight)
        sums = torch.from_numpy(np.concatenate([i, u, [b]]))
        world = 1
        if data_parallel and dist.is_initialized():
            dist.all_reduce(sums)
            world = dist.get_world_size()
        c = p.shape[1]
        count = float(p.numel()) * world
        s = sums.numpy()
        _half, dice, bce = O.np_criterion_from_sums(s[:c], s[c:2 * c], s[2 * c], count)
        ctx.save_for_backward(p.detach(), g)
        ctx.cfg = (s, count, w_dice, w_bce, bg_weight)
        return torch.tensor(w_dice * dice + w_bce * bce, dtype=torch.float32)

    @staticmethod
    def backward(ctx, gout):
        p, g = ctx.saved_tensors
        s, count, w_dice, w_bce, bgw = ctx.cfg
        c = p.shape[1]
        both = O.np_criterion_grad(p.numpy(), g.numpy(), s[:c], s[c:2 * c], count, bgw, 1.0)                 # = (dDice + dBCE) / 2
        only_b = O.np_criterion_grad(p.numpy(), g.numpy(), s[:c], np.full(c, 1e150), count, bgw, 1.0)         # Dice term vanishes
        d_bce, d_dice = 2.0 * only_b, 2.0 * (both - only_b)
        return torch.from_numpy((w_dice * d_dice + w_bce * d_bce).astype(np.float32)) * gout, None, None, None, None, None


class _OracleCrit(torch.nn.Module):
    def __init__(self, w_dice, w_bce, bg_weight):
        super().__init__()
        self.w, self.data_parallel = (w_dice, w_bce, bg_weight), False

    def forward(self, x, y):
        return _OracleCritFn.apply(x[0], y[0], self.w[0], self.w[1], self.w[2], self.data_parallel)


class _CountMetric:
    name = "frac"

    def __init__(self):
        self.reset()

    def reset(self):
        self.accumulator, self.samples = 0.0, 0.0

    def update(self, out, tgt):
        self.accumulator += float((out[0] > 0.5).float().mean())
        self.samples += 1

    def get(self):
        return self.accumulator / self.samples


def _trainer_worker(rank, world, port, out_dir):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from brats2019_amd import train as TR
    params = O.make_params(SEED, **CFG)
    net = _OracleUNet(CFG, params)
    tr = TR.Trainer(name="dp", models_root=os.path.join(out_dir, "m%d" % world), model=net, rewrite=True, connect_tb=False)
    tr.state.cuda = False
    batches = []
    for i in range(2):
        batches.append(([torch.from_numpy(O.make_input(2, *DHW, seed=SEED + i))], [torch.from_numpy(O.make_target(2, *DHW, seed=SEED + i))]))
    crit = [_OracleCrit(1.0, 0.0, 1.0), _OracleCrit(0.0, 1.0, 1e-2)]                       # Dice_loss_joint + BCE_Loss(bg 1e-2), main.py:126-128
    m = _CountMetric()
    tr.state.cuda = False
    orig_train = tr.train

    def no_cuda_train(**kw):                      # Trainer.train starts from self.state.cuda, which load/fresh state sets True
        tr.state.cuda = False
        return orig_train(**kw)
    no_cuda_train(criterion=crit, optimizer=torch.optim.Adam, optimizer_params=dict(lr=1e-3, weight_decay=1e-6, amsgrad=True),
                  scheduler=torch.optim.lr_scheduler.StepLR, scheduler_params=dict(step_size=1, gamma=0.5),
                  training_data_loader=batches, evaluation_data_loader=[], split_into_tiles=False, pretrained_weights=None,
                  train_metrics=[m], val_metrics=[], track_metric="none", epoches=1, default_val=0.0,
                  comparator=lambda a, b: False, eval_cpu=False, continue_form_pretraining=False)
    w = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    np.savez(os.path.join(out_dir, "tr_w%d_r%d.npz" % (world, rank)), w=w, metric=np.float64(tr.state.train_metric["frac"][0]),
             ckpt=int(os.path.exists(os.path.join(tr.model_path, "dplast_model.pth"))), steps=tr.state.global_step,
             dp_flags=np.asarray([c.data_parallel for c in crit]))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_trainer_loop_shards_batches_and_sums_gradients(tmp_path):
    """Trainer.train (the reference's loop, train.py:59-127,178-241) under a 2-rank gloo group == the single-process run on the
    global batches: the criteria get `data_parallel`, each rank trains on its slice, gradients are summed before optimizer.step()
    (what nn.DataParallel's reduce_add does, main.py:61), replicas stay identical, metrics are the global-batch values and only
    rank 0 writes the checkpoints."""
    _trainer_worker(0, 1, 0, str(tmp_path))
    ref = dict(np.load(tmp_path / "tr_w1_r0.npz"))
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(tmp_path / ("tr_w2_r%d.npz" % r))) for r in range(2))
    assert np.array_equal(r0["w"], r1["w"])
    assert r0["dp_flags"].all() and not ref["dp_flags"].any()
    assert int(r0["steps"]) == int(ref["steps"]) == 2
    # Adam normalises the step: compare against the reference run elementwise (two steps, lr 1e-3 then 5e-4)
    dw = np.abs(r0["w"] - ref["w"])
    assert dw.max() < 2e-5 and dw.mean() < 2e-7, (dw.max(), dw.mean())
    moved = np.abs(ref["w"] - np.concatenate([v.ravel() for v in O.make_params(SEED, **CFG).values()]))
    assert moved.max() > 1e-3                                 # the run really trained
    assert abs(float(r0["metric"]) - float(ref["metric"])) < 1e-6 and float(r0["metric"]) == float(r1["metric"])
    assert int(r0["ckpt"]) == 1 and int(r1["ckpt"]) == 1 and os.path.exists(tmp_path / "m2" / "dp" / "dplast_model.pth")


def test_bare_bench_spawns_ranks_and_relays_their_exit_code():
    """bench.py started bare with --gpus 2 (WORLD_SIZE unset) must become the launcher: it starts `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py <same arguments>` as a child and hands back its exit code.
    Without a GPU the ranks refuse to run (HIP-only path, no CPU fallback): the launcher must relay that failure, not hide it, and
    no JSON line may appear.  (The success path runs on the GPU box: tests/test_parallel_gpu.py::test_bare_bench_launches_its_own_ranks.)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "[rank0]" in r.stderr and "[rank1]" in r.stderr   # both ranks were started (under torch.distributed.run's elastic agent) ...
    assert "No HIP GPUs are available" in r.stderr or "no ROCm GPU visible" in r.stderr      # ... and refused to run without a GPU
    assert "ChildFailedError" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


# ---------------------------------------------------------------------- a degraded line instead of no line (round-4 verdict, item 7)
def _fallback_worker(rank, world, port, out_dir, inject):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from brats2019_amd import parallel as P
    r, local, w, group, info = P.init_process_groups_with_fallback("nccl", inject_failure=inject, timeout_s=30)
    t = torch.full((3,), float(rank + 1))
    dist.all_reduce(t, group=group)                         # the data collective still works -- over the group the ranks agreed on
    import json
    json.dump({"rank": r, "world": w, "group_is_default": group is None, "info": info, "sum": t.tolist(), "backend": dist.get_backend(group)},
              open(os.path.join(out_dir, "fb%d.json" % rank), "w"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("inject", [True, False], ids=["injected", "no-gpu-here"])
def test_rccl_health_check_falls_back_to_gloo_on_every_rank(tmp_path, inject):
    """parallel.init_process_groups_with_fallback (what bench.py's ranks call under the driver's launcher): when RCCL does not come up --
    injected, or for real on this GPU-less container where `new_group(backend="nccl")` / the health-check all-reduce cannot run -- EVERY rank
    agrees (over the gloo default group) to run the data collectives over gloo, with the first failing rank's error line as the reason."""
    import json
    if not inject and torch.cuda.is_available():
        pytest.skip("RCCL comes up on a GPU box: the real-failure case needs a container without GPUs")
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), str(tmp_path), inject), nprocs=world, join=True)
    rows = [json.load(open(tmp_path / ("fb%d.json" % r))) for r in range(world)]
    for r, row in enumerate(rows):
        assert row["rank"] == r and row["world"] == world and row["group_is_default"] and row["backend"] == "gloo"
        assert row["info"]["backend"] == "gloo" and row["info"]["fallback_reason"].startswith("rank 0: ")
        assert row["sum"] == [3.0, 3.0, 3.0]
    assert rows[0]["info"]["fallback_reason"] == rows[1]["info"]["fallback_reason"]
    if inject:
        assert "injected RCCL failure" in rows[0]["info"]["fallback_reason"]


@pytest.mark.timeout(900)
def test_bare_bench_starts_one_fresh_gloo_set_after_an_rccl_failure():
    """bench.self_launch: the ranks of the first set die over the RCCL backend (RU_BENCH_INJECT_RCCL_FAIL=2: before any GPU call) without a
    JSON line -> the launcher starts exactly ONE fresh set with `--dist-backend gloo` (fresh processes, a new port) and hands the reason to
    it.  Without a GPU that second set refuses to run as well (HIP-only path): the exit code is relayed, the notice and the first error
    line are on stderr, no JSON line appears, and there is no third attempt.  (With a GPU: tests/test_parallel_gpu.py, flagged line.)"""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["RU_BENCH_INJECT_RCCL_FAIL"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode != 0
    assert r.stderr.count("starting ONE fresh set of ranks with --dist-backend gloo") == 1
    assert "injected RCCL failure (RU_BENCH_INJECT_RCCL_FAIL=2)" in r.stderr
    assert "No HIP GPUs are available" in r.stderr or "no ROCm GPU visible" in r.stderr      # the gloo set got as far as the HIP-only path
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    import bench
    assert bench._first_error_line("W0101 warning: x\nTraceback (most recent call last):\n  File a\nRuntimeError: NCCL error in: foo\nmore") == "RuntimeError: NCCL error in: foo"
