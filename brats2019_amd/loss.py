"""Drop-in for the reference's `loss` module on the hot path: `Dice_loss_joint(index, priority)` and
`BCE_Loss(index, bg_weight)` with the reference call convention `forward(x_list, y_list) -> 0-dim tensor`
(loss.py:64-79,98-122), backed by the two-phase HIP criterion kernels:

  phase 1  ru_criterion_sums : per-class sum(p*g), sum(p^2+g) and the BCE log-sum of this rank's shard
  (data-parallel: all-reduce the [2C+1] float64 sums -- the Dice sums couple the GLOBAL batch, loss.py:114-115)
  phase 2  ru_criterion_grad : d loss / d p from the global sums

`FusedCriterion` is the training criterion of main.py:126-128 -- (Dice + BCE(bg 1e-2)) / 2 -- in one pass of
each phase; the separate modules stay available so `criterion=[Dice_loss_joint(), BCE_Loss()]` lists keep working
(train.py:203-205).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def _all_reduce_sums(sums, group):
    import torch.distributed as dist
    if group is not False and dist.is_available() and dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=None if group in (None, True) else group)
        return dist.get_world_size(None if group in (None, True) else group)
    return 1


_HAND_OVER_DEPTH = 0          # process-wide on purpose: autograd runs the backward of device tensors on its own thread


class hand_over_to_network:
    """`with loss.hand_over_to_network(): loss.backward()` -- what the Trainer loop does (train.py:210).  Inside it the backward of a
    criterion whose `pred` is DIRECTLY model.UNet's output does not write d(loss)/d(pred): it describes it to the network's autograd node,
    which forms it inside its first backward pass (ru_unet_backward_criterion).  The price is stated here rather than paid silently: within
    the context d(loss)/d(probs) does not exist as a tensor, so it must not be asked for (autograd.grad(loss, probs), retain_grad, hooks on
    the probabilities -- the two latter are detected and switch the hand-over off).  Outside the context every backward writes the gradient."""

    def __enter__(self):
        global _HAND_OVER_DEPTH
        _HAND_OVER_DEPTH += 1
        return self

    def __exit__(self, *exc):
        global _HAND_OVER_DEPTH
        _HAND_OVER_DEPTH -= 1
        return False


def _observed(pred):
    """somebody wants to SEE d(loss)/d(pred): retain_grad() or a tensor hook on the probabilities"""
    return bool(getattr(pred, "retains_grad", False)) or bool(getattr(pred, "_backward_hooks", None))


def _hand_over(pred, gt, sums, count, w_dice, w_bce, bg_weight, priority, gout):
    """Backward of a criterion whose `pred` is DIRECTLY the output of model.UNet's autograd node, inside `hand_over_to_network()`: instead of
    writing d(loss)/d(pred) (a 100 MB pass that the network's backward reads straight back), describe it to that node -- it forms the
    gradient inside its first pass (ru_unet_backward_criterion) -- and return a zero-stride placeholder of pred's shape.  Returns None when
    the hand-over does not apply (not enabled, another producer of pred, the probabilities are observed, d/d(input) wanted): the caller then
    writes the gradient.  The description never outlives the graph task that created it (a callback at its end clears it), so a backward
    that stops short of the network (autograd.grad(loss, probs)) cannot leak it into the next one."""
    node = pred.grad_fn
    if node is None or not getattr(node, "accepts_criterion", False) or type(node).__name__ != "_UNetFnBackward":
        return None
    if getattr(node, "pending_criterion", None) is not None:
        # a second criterion on the same probabilities (the list evaluated module by module): both gradients are written out -- the
        # first one's placeholder is zeros, so autograd's sum is this tensor
        pc, node.pending_criterion = node.pending_criterion, None
        first = ops.criterion_grad(pred, pc["target"], pc["sums"], pc["count"], pc["w_dice"], pc["w_bce"], pc["bg_weight"], pc["priority"]).mul_(pc["gout"].to(torch.float32))
        return first.add_(ops.criterion_grad(pred, gt, sums, count, w_dice, w_bce, bg_weight, priority).mul_(gout.to(torch.float32)))
    if _HAND_OVER_DEPTH <= 0 or _observed(pred):
        return None
    dummy = torch.zeros((), dtype=pred.dtype, device=pred.device).expand(pred.shape)
    node.pending_criterion = dict(target=gt, sums=sums, count=count, w_dice=w_dice, w_bce=w_bce, bg_weight=bg_weight, priority=priority, gout=gout, dummy=dummy)

    def _clear():                                     # end of THIS graph task: whether or not the network's node ran, nothing stays behind
        node.pending_criterion = None
    torch.autograd.Variable._execution_engine.queue_callback(_clear)
    return dummy


class _CriterionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, w_dice, w_bce, bg_weight, priority, group):
        sums = ops.criterion_sums(pred, gt, bg_weight)
        world = _all_reduce_sums(sums, group)
        count = float(pred.numel()) * world
        out = ops.criterion_losses(sums, count, priority, w_dice, w_bce)
        ctx.save_for_backward(pred, gt, sums)
        ctx.cfg = (count, w_dice, w_bce, bg_weight, priority)
        return out[0].to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        pred, gt, sums = ctx.saved_tensors
        count, w_dice, w_bce, bg_weight, priority = ctx.cfg
        dp = _hand_over(pred, gt, sums, count, w_dice, w_bce, bg_weight, priority, gout)
        if dp is None:
            dp = ops.criterion_grad(pred, gt, sums, count, w_dice, w_bce, bg_weight, priority)
            dp.mul_(gout.to(dp.dtype))
        return dp, None, None, None, None, None, None


class _PairFn(torch.autograd.Function):
    """[Dice_loss_joint, BCE_Loss] evaluated together (train.py:203-205: loss = (dice + bce) / 2): ONE sums pass gives both values,
    ONE gradient pass gives d loss / d p.  Returns (loss, dice, bce); only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, pred, gt, bg_weight, priority, group):
        sums = ops.criterion_sums(pred, gt, bg_weight)
        world = _all_reduce_sums(sums, group)
        count = float(pred.numel()) * world
        out = ops.criterion_losses(sums, count, priority, 0.5, 0.5).to(torch.float32)
        ctx.save_for_backward(pred, gt, sums)
        ctx.cfg = (count, bg_weight, priority)
        loss, dice, bce = out[0], out[1], out[2]
        ctx.mark_non_differentiable(dice, bce)
        return loss, dice, bce

    @staticmethod
    def backward(ctx, gout, _gd, _gb):
        pred, gt, sums = ctx.saved_tensors
        count, bg_weight, priority = ctx.cfg
        dp = _hand_over(pred, gt, sums, count, 0.5, 0.5, bg_weight, priority, gout)
        if dp is None:
            dp = ops.criterion_grad(pred, gt, sums, count, 0.5, 0.5, bg_weight, priority)
            dp.mul_(gout.to(dp.dtype))
        return dp, None, None, None, None


def fuse_criterion_list(criterion):
    """`criterion=[Dice_loss_joint(index, priority), BCE_Loss(index, bg_weight)]` (main.py:126-128, either order, same index) ->
    a callable `(x_list, y_list) -> (loss, [value per list entry])` with loss == sum(values) / 2 exactly as train.py:203-205 forms it,
    computed by one pass of each criterion phase instead of two.  Anything else -> None (the caller evaluates the list as written)."""
    if not isinstance(criterion, (tuple, list)) or len(criterion) != 2:
        return None
    kinds = [type(c) for c in criterion]
    if sorted(k.__name__ for k in kinds) != ["BCE_Loss", "Dice_loss_joint"] or not all(k in (Dice_loss_joint, BCE_Loss) for k in kinds):
        return None
    dice = criterion[0] if kinds[0] is Dice_loss_joint else criterion[1]
    bce = criterion[0] if kinds[0] is BCE_Loss else criterion[1]
    if dice.index != bce.label_index or dice.data_parallel != bce.data_parallel:
        return None
    dice_first = kinds[0] is Dice_loss_joint

    def run(x, y):
        assert x[dice.index].shape == y[dice.index].shape                  # loss.py:71,107
        loss, d, b = _PairFn.apply(x[dice.index], y[dice.index], float(bce.bg_weight), float(dice.priority), dice.data_parallel)
        return loss, ([d, b] if dice_first else [b, d])
    return run


class _LossBase(nn.Module):
    """`data_parallel`: False (reference semantics, single process), or True / a process group: all-reduce the partial
    sums so that every rank forms the GLOBAL-batch loss the reference computes on GPU 0 under nn.DataParallel."""

    def __init__(self):
        super().__init__()
        self.data_parallel = False

    def _run(self, x, y, index, w_dice, w_bce, bg_weight, priority):
        assert x[index].shape == y[index].shape                   # loss.py:71,107
        return _CriterionFn.apply(x[index], y[index], w_dice, w_bce, bg_weight, priority, self.data_parallel)


class Dice_loss_joint(_LossBase):
    """loss.py:98-122: priority * (1 - mean_c 2(I_c+1e-6)/(U_c+2e-6)), sums over batch and space jointly."""

    def __init__(self, index=0, priority=1):
        super().__init__()
        self.index = index
        self.priority = priority

    def forward(self, x, y):
        return self._run(x, y, self.index, 1.0, 0.0, 1.0, float(self.priority))


class BCE_Loss(_LossBase):
    """loss.py:64-79: -mean(g log(p+1e-6) + bg_weight (1-g) log(1+1e-6-p))."""

    def __init__(self, index=0, bg_weight=1):
        super().__init__()
        self.label_index = index
        self.bg_weight = bg_weight

    def forward(self, x, y):
        return self._run(x, y, self.label_index, 0.0, 1.0, float(self.bg_weight), 1.0)


class FusedCriterion(_LossBase):
    """(Dice_loss_joint(priority) + BCE_Loss(bg_weight)) / 2 == train.py:203-205 applied to main.py:126-128."""

    def __init__(self, index=0, priority=1, bg_weight=1e-2):
        super().__init__()
        self.index, self.priority, self.bg_weight = index, priority, bg_weight

    def forward(self, x, y):
        return self._run(x, y, self.index, 0.5, 0.5, float(self.bg_weight), float(self.priority))
