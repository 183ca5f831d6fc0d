"""Data-parallel training step: one process per GPU, full replica each, batch sharded over ranks.

Replaces `torch.nn.DataParallel` (main.py:61): instead of per-step replicate / scatter / gather through GPU 0,
every rank owns its shard and two all-reduces per step keep the result identical to the reference's
global-batch step:

  1. all-reduce(SUM) of the [2C+1] float64 criterion partial sums -- Dice_loss_joint sums over the GLOBAL batch
     (loss.py:114-115), so each rank must form d loss/d p with global (I_c, U_c) and the global element count;
  2. all-reduce(SUM, not mean) of the flat parameter-gradient buffer (18 MB fp32 live + zeros of the dead
     parameters); xGMI: one contiguous bucket, RCCL picks ring/direct.

`torch.distributed` is the transport (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).  The
arithmetic is behind `backend`: the product uses `HipBackend` (libresunet_hip.so); tests inject a CPU
backend built on the oracle to check the sharding / reduction logic without a GPU.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


class HipBackend:
    """Arithmetic of the step on the local GPU through the C-ABI (no CPU fallback)."""

    def __init__(self, cfg=None, device=None, precision=None, grad_precision=None):
        from . import _lib as L
        from .engine import UNetEngine, DEFAULT_CFG
        from .model import default_precision
        L.require_gpu()
        self.cfg = dict(cfg or DEFAULT_CFG)
        self.engine = UNetEngine(precision=precision or default_precision(self.cfg["number_of_channels"]), **self.cfg)
        if grad_precision:
            self.engine.set_grad_precision(grad_precision)
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.total = self.engine.layout.total
        # contiguous runs of LIVE parameters: the reference's Adam skips tensors whose grad is None (the never-executed
        # decoder_convs.{depth-1} / decoder_convs1x1.{depth-1}, model.py:420), so they must not even see weight decay
        self.live_segments = []
        for _name, (shape, off, dead) in self.engine.layout.entries.items():
            n = 1
            for d in shape:
                n *= d
            if dead:
                continue
            if self.live_segments and self.live_segments[-1][1] == off:
                self.live_segments[-1][1] = off + n
            else:
                self.live_segments.append([off, off + n])

        self.reduce_runs = merge_runs(self.live_segments)

    def new_flat(self, fill=0.0):
        return torch.full((self.total,), float(fill), dtype=torch.float32, device=self.device)

    def forward(self, flat, x, training=True):
        return self.engine.forward(flat, x, training=training)

    def criterion_sums(self, probs, target, bg_weight):
        from . import ops
        return ops.criterion_sums(probs, target, bg_weight)

    def criterion_losses(self, sums, count, priority):
        """(loss, dice, bce) 0-dim float64 device tensors, loss = (dice + bce) / 2 (train.py:203-205)."""
        from . import ops
        out = ops.criterion_losses(sums, count, priority, 0.5, 0.5)
        return out[0], out[1], out[2]

    def criterion_grad(self, probs, target, sums, count, bg_weight, priority):
        from . import ops
        return ops.criterion_grad(probs, target, sums, count, 0.5, 0.5, bg_weight, priority)

    def backward(self, flat, dprobs, grads):
        return self.engine.backward(flat, dprobs, flat_grads=grads)

    def backward_criterion(self, flat, target, sums, count, bg_weight, priority, grads):
        """criterion gradient + backward in one call: d(loss)/d(probs) is formed inside the head's first backward pass, never written"""
        return self.engine.backward_criterion(flat, target, sums, count, 0.5, 0.5, bg_weight, priority, flat_grads=grads)

    def adam(self, flat, grads, m, v, vmax, step, lr, betas, eps, weight_decay):
        from . import ops
        for a, b in self.live_segments:
            ops.adam_amsgrad_step(flat[a:b], grads[a:b], m[a:b], v[a:b], vmax[a:b], step, lr, betas, eps, weight_decay)


def merge_runs(runs, max_gap=1 << 16):
    """Runs of the flat gradient bucket that travel in the all-reduce: the live runs, joined across dead gaps of at most `max_gap`
    floats (decoder_convs1x1.{depth-1}, 32 768 zeros, is cheaper to send than a third collective); the large dead block
    (decoder_convs.{depth-1}, 0.9 M floats) still splits the bucket.  Dead gradients are zero on every rank, so their sum stays zero and
    the optimizer never reads them (HipBackend.adam walks live_segments)."""
    out = []
    for a, e in runs:
        if out and a - out[-1][1] <= max_gap:
            out[-1][1] = e
        else:
            out.append([a, e])
    return out


class RcclComm:
    """The two collectives of a step through the C-ABI (`ru_comm_*`, `ru_allreduce`): RCCL all-reduce enqueued on torch's current HIP
    stream, in place -- no torch.distributed call in the data path.  The 128-byte RCCL unique id travels once, at set-up: rank 0 makes
    it, `torch.distributed` (any backend, e.g. the gloo group torchrun's environment gives) or the caller broadcasts it."""

    def __init__(self, rank, world, unique_id=None, group=None):
        import ctypes as C
        from . import _lib as L
        L.require_gpu()
        lib = L.load()
        self.rank, self.world = int(rank), int(world)
        if unique_id is None:
            buf = (C.c_char * 128)()
            if self.rank == 0:
                L.check(lib.ru_comm_unique_id(C.cast(buf, C.c_void_p)), "ru_comm_unique_id")
            box = [bytes(buf)]
            if self.world > 1:
                if not (dist.is_available() and dist.is_initialized()):
                    raise RuntimeError("RcclComm: pass unique_id, or initialise torch.distributed so that rank 0 can broadcast it")
                dist.broadcast_object_list(box, src=0, group=group)
            unique_id = box[0]
        self.unique_id = bytes(unique_id)
        h = C.c_void_p()
        idbuf = C.create_string_buffer(self.unique_id, 128)
        L.check(lib.ru_comm_init(C.byref(h), C.cast(idbuf, C.c_void_p), self.rank, self.world), "ru_comm_init")
        self.h = h

    def all_reduce(self, t):
        """in-place SUM over the ranks of a contiguous float32 / float64 device tensor, on the current stream"""
        from . import _lib as L
        if t.dtype not in (torch.float32, torch.float64):
            raise TypeError("RcclComm.all_reduce: float32 or float64, got %s" % t.dtype)
        L.check(L.load().ru_allreduce(self.h, L.ptr(t), t.numel(), 0 if t.dtype == torch.float32 else 1, L.stream()), "ru_allreduce")
        return t

    def all_reduce_many(self, tensors):
        """the all-reduces of `tensors` as ONE RCCL launch (ncclGroupStart / ncclGroupEnd)"""
        from . import _lib as L
        if len(tensors) <= 1:
            for t in tensors:
                self.all_reduce(t)
            return
        lib = L.load()
        L.check(lib.ru_comm_group_begin(), "ru_comm_group_begin")
        try:
            for t in tensors:
                self.all_reduce(t)
        except BaseException:
            lib.ru_comm_group_end()              # close the group, but let the ORIGINAL error travel (its message names the failing call)
            raise
        L.check(lib.ru_comm_group_end(), "ru_comm_group_end")

    def close(self):
        from . import _lib as L
        if getattr(self, "h", None):
            L.load().ru_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DataParallelStep:
    """forward + criterion + backward + gradient all-reduce + Adam(amsgrad) + StepLR, hyper-parameters of
    main.py:126-142 (lr 2e-5, wd 1e-6, amsgrad, StepLR(16000, 0.5) stepped per iteration, train.py:220-223)."""

    def __init__(self, backend, flat_params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6,
                 step_size=16000, gamma=0.5, bg_weight=1e-2, priority=1.0, process_group=None, comm=None):
        self.backend = backend
        self.flat = flat_params
        self.grads = torch.zeros_like(flat_params)
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.vmax = torch.zeros_like(flat_params)
        self.base_lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_size, self.gamma = step_size, gamma
        self.bg_weight, self.priority = bg_weight, priority
        self.group = process_group
        self.global_step = 0
        self.comm = comm                       # RcclComm: the collectives go through ru_allreduce on the kernels' stream
        self.fuse_criterion_grad = True        # the criterion's gradient is formed inside the backward's first pass (ru_unet_backward_criterion)
        self.comm_probe = None                 # bench.py: a list -> every collective of a step is bracketed by an event pair on the current stream
        if comm is not None:
            self.distributed, self.world, self.rank = True, comm.world, comm.rank
        else:
            self.distributed = dist.is_available() and dist.is_initialized()
            self.world = dist.get_world_size(self.group) if self.distributed else 1
            self.rank = dist.get_rank(self.group) if self.distributed else 0

    @staticmethod
    def shard(batch_size, rank, world):
        """Rank r of W owns samples [r*B/W, (r+1)*B/W) of the global batch."""
        if batch_size % world != 0:
            raise ValueError("global batch %d is not divisible by world size %d" % (batch_size, world))
        per = batch_size // world
        return slice(rank * per, (rank + 1) * per)

    def lr(self):
        return self.base_lr * self.gamma ** (self.global_step // self.step_size)

    def loss_and_grads(self, x_shard, target_shard):
        """Forward/backward of this rank's shard with the GLOBAL criterion; leaves the all-reduced (summed)
        gradient in self.grads.  Returns (loss, dice, bce) as 0-dim float64 tensors (identical on all ranks)."""
        b = self.backend
        probs = b.forward(self.flat, x_shard, training=True)
        sums = b.criterion_sums(probs, target_shard, self.bg_weight)
        if self.distributed:
            self._all_reduce(sums)
        count = float(probs.numel()) * self.world
        loss, dice, bce = b.criterion_losses(sums, count, self.priority)
        if self.fuse_criterion_grad and hasattr(b, "backward_criterion"):
            b.backward_criterion(self.flat, target_shard, sums, count, self.bg_weight, self.priority, self.grads)
        else:
            dprobs = b.criterion_grad(probs, target_shard, sums, count, self.bg_weight, self.priority)
            b.backward(self.flat, dprobs, self.grads)
        if self.distributed:
            # only the live runs of the flat bucket travel: the never-executed deepest decoder stage (a third of the 21.7 MB) has zero
            # gradients on every rank.  SUM, not mean: the criterion already carries the global 1/count.
            # The runs go out as ONE grouped collective (ncclGroupStart/End through either transport).
            segs = getattr(b, "reduce_runs", None) or getattr(b, "live_segments", None)
            if segs and sum(e - a for a, e in segs) < 0.9 * self.grads.numel():
                self._all_reduce_many([self.grads[a:e] for a, e in segs])
            else:
                self._all_reduce(self.grads)
        self.last_probs = probs
        return loss, dice, bce

    def _probed(self, what, fn):
        """comm_probe: (name, begin, end) event triples around the collective on the stream of the kernels.  With torch.distributed the
        collective runs on the process group's own stream, which waits for this one and which this one waits for: the pair spans
        exactly the collective (plus the two stream hand-offs)."""
        if self.comm_probe is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.comm_probe.append((what, e0, e1))

    def _all_reduce(self, t):
        if self.comm is not None:
            self._probed("criterion_sums" if t.numel() < 64 else "gradients", lambda: self.comm.all_reduce(t))
        else:
            self._probed("criterion_sums" if t.numel() < 64 else "gradients", lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))

    def _all_reduce_many(self, tensors):
        if self.comm is not None:
            self._probed("gradients", lambda: self.comm.all_reduce_many(tensors))
        else:
            self._probed("gradients", lambda: all_reduce_coalesced(tensors, self.group))

    def step(self, x_shard, target_shard):
        loss, dice, bce = self.loss_and_grads(x_shard, target_shard)
        self.global_step += 1
        # optimizer.step() then scheduler.step() (train.py:220-223): update k uses the lr after k-1 scheduler steps
        lr = self.base_lr * self.gamma ** ((self.global_step - 1) // self.step_size)
        self.backend.adam(self.flat, self.grads, self.m, self.v, self.vmax, self.global_step, lr, self.betas, self.eps, self.weight_decay)
        return loss, dice, bce

    def state_dict(self):
        return dict(m=self.m, v=self.v, vmax=self.vmax, global_step=self.global_step)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.vmax.copy_(sd["vmax"])
        self.global_step = int(sd["global_step"])


def all_reduce_coalesced(tensors, group=None):
    """SUM all-reduce of several tensors as one coalesced collective (ProcessGroup.allreduce_coalesced: one ncclGroup'd launch on RCCL,
    one flattened ring on gloo); every rank must pass the same number of tensors of the same sizes."""
    if len(tensors) == 1 or (tensors[0].is_cuda and dist.get_backend(group) == "gloo"):
        # (ProcessGroupGloo has no coalesced all-reduce for device tensors -- the shared-GPU plumbing tests: one collective per run)
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    cm = getattr(dist, "_coalescing_manager", None)      # private API (its signature moved between torch 2.x releases): optional
    if cm is not None:
        try:
            ctx = cm(group=group)
        except TypeError:
            ctx = None
        if ctx is not None:
            with ctx:
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return
    for t in tensors:                                    # fall-back: one collective per run (same result)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def world_info(group=None):
    """(rank, world) of the initialised process group, (0, 1) otherwise."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_reduce_gradients(model, group=None):
    """What nn.DataParallel's reduce_add does for the reference loop (main.py:61, train.py:210): after `loss.backward()` on this
    rank's shard, SUM the parameter gradients over the ranks (SUM, not mean -- the criterion with `data_parallel` set already
    carries the global 1/count).  Parameters whose grad is None (the never-executed deepest decoder stage) stay None
    on every rank.  When the gradients still alias the flat bucket the HIP executor wrote (`_UNetFn.backward`), the live runs of
    that bucket are reduced in place; otherwise (an accumulation copied them) they are packed into one bucket, reduced, unpacked."""
    rank, world = world_info(group)
    if world <= 1:
        return 0
    net = model.module if hasattr(model, "module") else model
    params = [p for p in net.parameters() if p.grad is not None]
    if not params:
        return 0
    flat = getattr(net, "__dict__", {}).get("_last_flat_grads")
    if flat is not None:
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
        runs = []
        for p in params:
            g = p.grad
            a = g.data_ptr()
            if not (g.dtype == torch.float32 and g.is_contiguous() and lo <= a and a + 4 * g.numel() <= hi):
                runs = None
                break
            a = (a - lo) // 4
            if runs and runs[-1][1] == a:
                runs[-1][1] = a + g.numel()
            else:
                runs.append([a, a + g.numel()])
        if runs:
            runs = merge_runs(runs)              # (the executor zero-fills the bucket: dead gaps are zeros on every rank)
            all_reduce_coalesced([flat[a:e] for a, e in runs], group)
            return sum(e - a for a, e in runs)
    bucket = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(bucket[off:off + n].view_as(p.grad))
        off += n
    return off


def init_process_group_from_env(backend=None):
    """One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 0, 1
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # The host driver of this pool (and of the GPU boxes) supports only dmabuf IPC: with the legacy mode RCCL's intra-node transport set-up
    # and any cross-process device-memory sharing fail with `hipIpcGetMemHandle: invalid argument`.  The image exports the variable
    # already; a launcher that scrubs the environment would lose it, so it is restored here -- before the first HIP call of the rank --
    # and never overridden when the operator set it.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def init_process_groups_with_fallback(prefer="nccl", inject_failure=False, timeout_s=120, device_index=None, allow_single=False):
    """One-shot multi-GPU runs (bench.py under the driver's launcher): a run on a node nobody can look at must not end without a data
    point because RCCL could not come up.  The DEFAULT process group is gloo (control plane: agreement, and the fallback transport); the
    DATA group is RCCL (`backend="nccl"`), created on top and health-checked with one tiny all-reduce on every rank.  The ranks agree
    over gloo whether ALL of them came through; if any did not, every rank uses the gloo group for the data collectives and the line says
    so (`info["fallback_reason"]` = the first failing rank's error line).  Nothing is re-executed and no process is replaced.
    -> (rank, local_rank, world, data_group_or_None_for_the_default_group, info)"""
    import datetime
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not (allow_single and "RANK" in os.environ):       # (allow_single: tests run the group set-up with ONE rank on the one-GPU boxes)
        return 0, 0, 1, None, {"backend": None, "fallback_reason": None}
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # see init_process_group_from_env
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    info = {"backend": "gloo", "fallback_reason": None}
    if prefer != "nccl":
        return rank, local, world, None, info
    err, group = None, None
    try:
        if inject_failure:
            raise RuntimeError("injected RCCL failure (RU_BENCH_INJECT_RCCL_FAIL=1): ncclCommInitRank would have failed here")
        torch.cuda.set_device(local if device_index is None else device_index)      # the health check's tensor and the RCCL communicator live on this rank's GPU
        # the data group keeps torch's normal collective timeout for the run; only the HEALTH CHECK is bounded by timeout_s (a rank that died before the
        # check leaves the others waiting here: they give up after timeout_s and fall back; a watchdog abort later in the run still ends the ranks, and
        # only the bare launcher's fresh gloo set covers that case -- bench.self_launch)
        group = dist.new_group(backend="nccl")
        t = torch.ones(1, device="cuda")
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
        if not work.wait(datetime.timedelta(seconds=timeout_s)):
            raise RuntimeError("RCCL health check: the all-reduce did not complete within %d s" % timeout_s)
        torch.cuda.synchronize()
        if float(t.item()) != float(world):
            raise RuntimeError("RCCL health check: all-reduce of ones over %d ranks returned %r" % (world, float(t.item())))
    except Exception as e:                                          # noqa: BLE001 -- whatever RCCL / torch raise here is the reason we report
        lines = [ln.strip() for ln in str(e).splitlines() if ln.strip()]
        err = ("%s: %s" % (type(e).__name__, lines[0] if lines else ""))[:300]
    ok = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                       # gloo, host tensor
    if int(ok.item()) == 1:
        info["backend"] = "nccl"
        return rank, local, world, group, info
    reasons = [None] * world
    dist.all_gather_object(reasons, err)
    info["fallback_reason"] = next(("rank %d: %s" % (i, r) for i, r in enumerate(reasons) if r), "unknown")
    return rank, local, world, None, info
