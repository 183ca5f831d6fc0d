"""Drop-in for the reference's `model` module on the hot path: same public classes (`UNet`, `Residual`,
`conv`, `Trilinear`), same constructor signatures, same state_dict() keys / shapes / order, same call
convention (`forward(x)` takes a LIST and reads x[0]; returns a LIST with the sigmoid probabilities,
model.py:407-433) -- but `UNet.forward` is ONE call into libresunet_hip.so (engine.py), not a chain of
torch.nn ops.  The nn.Conv3d / nn.GroupNorm objects below only OWN the parameters (so checkpoints,
optimizers and `weight_init` keep working); their storage is aliased onto one flat device buffer that
the HIP executor reads directly.

Reference checkpoints pickle whole module objects under the module name `model` (train.py:320-324);
`brats2019_amd.compat.install_aliases()` registers this module under that name so they un-pickle into
the classes here.  A UNet restored that way never ran __init__, so everything `forward` needs is
derived lazily from the attributes the reference's __init__ sets (model.py:313-318).

HIP-only: calling forward with CPU tensors raises (no CPU fallback; the CPU restatement is oracle/).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .engine import UNetEngine

LEAKY_SLOPE = 1e-2


def default_precision(number_of_channels):
    """The benchmarked split-bf16 engine needs channel counts divisible by 16 (voxel-major C16 layout); it is what a drop-in user
    gets for main.py:56-59.  Other configurations run the exact-f32 NCDHW kernels."""
    return "bf16x3" if all(int(c) % 16 == 0 for c in number_of_channels) else "f32"


# ---------------------------------------------------------------------- per-op autograd (building blocks used stand-alone)
class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return ops.conv3d(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k = int(w.shape[2])
        dx = ops.conv3d_bwd_data(dy, w, x.shape[2:]) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.has_bias:
            if ctx.has_bias:
                dw, db = ops.conv3d_bwd_weight(x, dy, k, with_bias=True)
            else:
                dw = ops.conv3d_bwd_weight(x, dy, k)
        return dx, dw, db


class _GroupNormActFn(torch.autograd.Function):
    """y = residual + lrelu(GroupNorm(x), slope)  (model.py:103-115 fused; slope 1 = no activation)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, groups, slope):
        y, mean, rstd = ops.group_norm(x, gamma, beta, groups, 1e-5, slope, residual)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.groups, ctx.slope, ctx.has_res = groups, slope, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = ops.group_norm_bwd(x, gamma, beta, mean, rstd, dy, ctx.groups, ctx.slope)
        return dx, dgamma, dbeta, (dy if ctx.has_res else None), None, None


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2x_bwd(dy)


class _LeakyReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        y = ops.leaky_relu(x, slope)
        ctx.save_for_backward(y)          # the reference's in-place LeakyReLU keeps the OUTPUT (model.py:352)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.leaky_relu_bwd(y, dy, ctx.slope), None


def conv3d_op(x, conv_module):
    """Apply an nn.Conv3d parameter holder through the HIP kernels."""
    return _ConvFn.apply(x, conv_module.weight, conv_module.bias)


# ---------------------------------------------------------------------- reference-named modules
class Trilinear(nn.Module):
    """model.py:7-14: F.interpolate(scale_factor=scale, mode='trilinear'); only scale 2 exists in the path."""

    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, x):
        if self.scale != 2:
            raise NotImplementedError("only the x2 trilinear up-sampling of model.py:399 is implemented")
        return _UpsampleFn.apply(x)


class conv(nn.Module):
    """model.py:66-79: 3x3x3, stride 1, pad 1, no bias; parameter key `conv1.weight`."""

    def __init__(self, in_channels, out_channels, stride=1, groups=1):
        super().__init__()
        if stride != 1 or groups != 1:
            raise NotImplementedError("the shipped configuration only uses stride=1, groups=1 (model.py:89-91)")
        self.conv1 = nn.Conv3d(in_channels, out_channels, kernel_size=(3, 3, 3), stride=stride, padding=1, bias=False, groups=groups)

    def forward(self, x):
        return conv3d_op(x, self.conv1)


class Residual(nn.Module):
    """model.py:81-117.  Stand-alone forward = 2 x (conv -> fused GroupNorm+LeakyReLU) + skip add."""

    def __init__(self, in_channels, out_channels, stride, downsample=None, conv_groups=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.downsample = downsample
        self.conv1 = conv(in_channels=in_channels, out_channels=out_channels, stride=stride)
        self.conv2 = conv(in_channels=out_channels, out_channels=out_channels, stride=1)
        self.relu1 = nn.LeakyReLU(LEAKY_SLOPE, inplace=True)
        self.relu2 = nn.LeakyReLU(LEAKY_SLOPE, inplace=True)
        self.norm1 = nn.GroupNorm(num_groups=8, num_channels=out_channels)
        self.norm2 = nn.GroupNorm(num_groups=8, num_channels=out_channels)

    def forward(self, x):
        if self.downsample is not None:
            x = conv3d_op(x, self.downsample[0])
        out = self.conv1(x)
        out = _GroupNormActFn.apply(out, self.norm1.weight, self.norm1.bias, None, 8, LEAKY_SLOPE)
        out = self.conv2(out)
        return _GroupNormActFn.apply(out, self.norm2.weight, self.norm2.bias, x, 8, LEAKY_SLOPE)


class _UNetFn(torch.autograd.Function):
    """Whole-network autograd node: forward/backward are single calls into the HIP executor."""

    @staticmethod
    def forward(ctx, net, training, x, *params):
        eng = net._get_engine()
        flat = net._flat_params()
        probs = eng.forward(flat, x, training=training)
        ctx.net, ctx.generation, ctx.training = net, eng.generation, training
        ctx.x_needs_grad = x.requires_grad
        # loss._PairFn / _CriterionFn hand their gradient over as a DESCRIPTION (target, sums, weights) when the probabilities they were given
        # are this node's output: the executor then forms d(loss)/d(probs) inside its first backward pass (ru_unet_backward_criterion)
        ctx.accepts_criterion = bool(training) and not x.requires_grad
        ctx.pending_criterion = None
        if training:
            ctx.save_for_backward(probs)          # the executor reads this buffer again in backward: autograd's version check then
        return probs                              # catches an in-place edit of the returned probabilities

    @staticmethod
    def backward(ctx, dprobs):
        net = ctx.net
        eng = net._get_engine()
        if not ctx.training:
            raise RuntimeError("UNet forward ran in inference mode; no activations were kept for backward")
        if eng.generation != ctx.generation:
            raise RuntimeError("UNet.forward was called again before backward(): the executor keeps ONE forward state")
        _ = ctx.saved_tensors                     # raises if the probabilities were modified in place since the forward
        flat = net._flat_params()
        pc, ctx.pending_criterion = ctx.pending_criterion, None
        if pc is not None and dprobs.data_ptr() == pc["dummy"].data_ptr() and not any(dprobs.stride()):
            # the incoming "gradient" is the criterion's zero-stride placeholder and nothing else was added to it: form the real one on the way
            grads = eng.backward_criterion(flat, pc["target"], pc["sums"], pc["count"], pc["w_dice"], pc["w_bce"], pc["bg_weight"], pc["priority"])
            grads.mul_(pc["gout"].to(grads.dtype))                 # d(final loss)/d(criterion value): 21 MB, once (1 for `loss.backward()`)
            dx = None
        else:
            if pc is not None:                    # somebody else also used the probabilities: the placeholder was summed with a real gradient
                from . import ops
                dprobs = dprobs + ops.criterion_grad(ctx.saved_tensors[0], pc["target"], pc["sums"], pc["count"], pc["w_dice"], pc["w_bce"], pc["bg_weight"],
                                                     pc["priority"]).mul_(pc["gout"].to(dprobs.dtype))
            res = eng.backward(flat, dprobs, want_dx=ctx.x_needs_grad)
            grads, dx = res if ctx.x_needs_grad else (res, None)
        net.__dict__["_last_flat_grads"] = grads      # parallel.all_reduce_gradients reduces this bucket in place when p.grad alias it
        views = eng.layout.views(grads)
        out = []
        for name in net._param_order():
            _shape, _off, dead = eng.layout.entries[name]
            out.append(None if dead else views[name])      # dead parameters keep grad=None like the reference (model.py:420)
        return (None, None, dx) + tuple(out)


class UNet(nn.Module):
    """model.py:308-433 with block=Residual.  `forward(x)`: x is a list, x[0] = [N,4,D,H,W]; returns [probs]."""

    def __init__(self, depth, encoder_layers, decoder_layers, number_of_channels, number_of_outputs, block=Residual):
        super().__init__()
        if block is not Residual:
            raise NotImplementedError("only block=Residual is on the accelerated path (main.py:56-59)")
        self.encoder_layers = encoder_layers
        self.decoder_layers = decoder_layers
        self.number_of_channels = number_of_channels
        self.number_of_outputs = number_of_outputs
        self.depth = depth
        self.block = block
        ch = number_of_channels
        # registration order fixes the state_dict() key order (model.py:320-357)
        self.encoder_convs = nn.ModuleList()
        self.upsampling = nn.ModuleList()
        self.decoder_convs = nn.ModuleList()
        self.decoder_convs1x1 = nn.ModuleList()
        self.attention_convs = nn.ModuleList()
        self.upsampling_distance = nn.ModuleList()
        self.conv_input = nn.Conv3d(4, ch[0], kernel_size=(3, 3, 3), stride=1, padding=(1, 1, 1), bias=False)
        self.norm_input = nn.GroupNorm(num_groups=8, num_channels=ch[0])
        self.conv_first = nn.Sequential(*[block(in_channels=ch[0], out_channels=ch[0], stride=1) for _ in range(encoder_layers[0])])
        self.conv_output = nn.Conv3d(ch[0], number_of_outputs, kernel_size=3, stride=1, padding=1, bias=True, groups=1)
        self.softmax = nn.Softmax(dim=1)
        self.sigmoid = nn.Sigmoid()
        self.relu = nn.LeakyReLU(LEAKY_SLOPE, inplace=True)
        for i in range(depth):                                   # model.py:379-395 (the last stage is built but never run)
            self.decoder_convs.append(nn.Sequential(*[block(in_channels=ch[i], out_channels=ch[i], stride=1)
                                                      for _ in range(decoder_layers[i])]))
            self.decoder_convs1x1.append(nn.Conv3d(2 * ch[i], ch[i], kernel_size=1, padding=0, bias=False))
        for i in range(depth - 1):                               # model.py:359-377
            down = nn.Sequential(nn.Conv3d(ch[i], ch[i + 1], kernel_size=2, stride=2, bias=False))
            layers = [block(in_channels=ch[i + 1], out_channels=ch[i + 1], stride=1, downsample=down)]
            layers += [block(in_channels=ch[i + 1], out_channels=ch[i + 1], stride=1) for _ in range(1, encoder_layers[i + 1])]
            self.encoder_convs.append(nn.Sequential(*layers))
        for i in range(depth - 1):                               # model.py:397-404
            self.upsampling.append(nn.Sequential(Trilinear(scale=2), nn.Conv3d(ch[i + 1], ch[i], kernel_size=1, stride=1, bias=False)))

    # ---- executor plumbing (all lazy: un-pickled instances never ran __init__)
    def _get_engine(self):
        eng = self.__dict__.get("_engine_obj")
        if eng is None:
            eng = UNetEngine(self.depth, list(self.encoder_layers), list(self.decoder_layers), list(self.number_of_channels),
                             self.number_of_outputs, precision=self.__dict__.get("precision") or default_precision(self.number_of_channels))
            names = [n for n, _ in self.named_parameters()]
            if names != list(eng.layout.entries.keys()):
                raise RuntimeError("parameter names/order differ from the executor's layout")
            for n, p in self.named_parameters():
                if tuple(p.shape) != eng.layout.entries[n][0]:
                    raise RuntimeError("parameter %s has shape %s, executor expects %s" % (n, tuple(p.shape), eng.layout.entries[n][0]))
            if self.__dict__.get("grad_precision"):
                eng.set_grad_precision(self.__dict__["grad_precision"])
            self.__dict__["_engine_obj"] = eng
            self.__dict__["_param_names"] = names
        return eng

    def set_precision(self, precision):
        """Arithmetic of the 3x3x3 convolutions: "bf16x3" (split-bf16 operands, 3 MFMA products, |dp| ~ 5e-5; the default when every
        channel count is a multiple of 16, i.e. for the shipped configuration) or "f32" (exact-f32 MFMA, ~3x slower; opt-in, and
        the default for other channel counts)."""
        if precision not in L.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(L.PRECISIONS))
        self.__dict__["precision"] = precision
        eng = self.__dict__.get("_engine_obj")
        if eng is not None:
            eng.set_precision(precision)
        return self

    def set_grad_precision(self, grad_precision):
        """Arithmetic of the 3x3x3 data / weight gradients under the "bf16x3" forward: "bf16x3" (default: gradients within ~1e-5 of
        float32) or "bf16" (gradient convolutions on bf16-rounded operands, one MFMA product -- the literal reading of BASELINE
        configs[2]'s "bf16 forward+backward"; ~15 % faster training steps, parameter gradients with bf16 rounding noise).  The forward
        pass and its probabilities are the same in both."""
        if grad_precision not in L.GRAD_PRECISIONS:
            raise ValueError("grad_precision must be one of %s" % sorted(L.GRAD_PRECISIONS))
        self.__dict__["grad_precision"] = grad_precision
        eng = self.__dict__.get("_engine_obj")
        if eng is not None:
            eng.set_grad_precision(grad_precision)
        return self

    def freeze_params(self, frozen=True):
        """Inference with constant weights (test.py): the executor builds its packed weights once and reuses them until the flag is
        cleared or the weights move.  Call again after load_state_dict / an optimizer step -- every call drops the cached packs."""
        self._get_engine().freeze_params(frozen)
        return self

    def _unfreeze(self):
        eng = self.__dict__.get("_engine_obj")
        if eng is not None and getattr(eng, "_frozen", False):
            eng.freeze_params(False)

    def train(self, mode=True):                                   # weights are about to change: never reuse packed copies
        if mode:
            self._unfreeze()
        return super().train(mode)

    def _drop_param_caches(self):
        """anything that can REPLACE nn.Parameter objects (load_state_dict(assign=True), an overwrite-on-conversion _apply, re-registration):
        the cached parameter list, its end pointers and the alias check are rebuilt from named_parameters() on the next forward"""
        for k in ("_flat_checked", "_flat_ends", "_params_cached", "_param_slots"):
            self.__dict__.pop(k, None)

    def load_state_dict(self, *args, **kwargs):                   # writes through the aliased flat buffer in place (assign=True: new Parameters)
        self._unfreeze()
        self._drop_param_caches()
        try:
            return super().load_state_dict(*args, **kwargs)
        finally:
            self._drop_param_caches()

    def _param_order(self):
        self._get_engine()
        return self.__dict__["_param_names"]

    def _apply(self, fn, *args, **kwargs):                       # .cuda() / .to() / .float(): the parameters move (or are replaced), the alias check must run again
        self._drop_param_caches()
        try:
            return super()._apply(fn, *args, **kwargs)
        finally:
            self._drop_param_caches()

    def _param_list(self):
        """the nn.Parameter objects in state-dict order (they outlive .cuda() / load_state_dict: only their .data moves)"""
        lst = self.__dict__.get("_params_cached")
        if lst is not None:
            # every nn.Parameter of this network lives in a SUBMODULE (Conv3d, GroupNorm), so a re-registration there -- `net.conv_output.weight = nn.Parameter(...)` --
            # never passes through this module's own register_parameter / __setattr__: the cached objects are checked against the submodules' parameter
            # dictionaries by identity on every use (93 dictionary look-ups, a few microseconds; named_parameters() itself walks the module tree)
            for (d, k), p in zip(self.__dict__["_param_slots"], lst):
                if d.get(k) is not p:
                    self._drop_param_caches()
                    lst = None
                    break
        if lst is None:
            lst, slots = [], []
            for mod in self.modules():
                for k, p in mod._parameters.items():
                    if p is not None:
                        slots.append((mod._parameters, k))
                        lst.append(p)
            assert len(lst) == len(list(self.parameters())) and all(a is b for a, b in zip(lst, self.parameters())), "parameter walk out of state-dict order"
            self.__dict__["_params_cached"] = lst
            self.__dict__["_param_slots"] = slots
            self.__dict__.pop("_flat_checked", None)             # a new list: the alias check walks all of it once
        return lst

    def _flat_params(self):
        """One flat float32 device buffer aliased by every nn.Parameter (re-built if .to()/.cuda() broke the aliasing)."""
        eng = self._get_engine()
        flat = self.__dict__.get("_flat_buf")
        lst = self._param_list()                                  # (drops `_flat_checked` when a Parameter object was replaced since the last forward)
        if flat is not None and self.__dict__.get("_flat_checked"):
            # steady state (a forward per step or per tile): the full walk over 93 pointers ran once after the last move; between moves only
            # the first and last parameter are looked at (an optimizer updates in place; `p.data = ...` on a middle tensor is not a supported move)
            first, last = self.__dict__["_flat_ends"]
            if lst[0].data_ptr() == first and lst[-1].data_ptr() == last:
                return flat
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        ok = flat is not None and flat.device == dev
        if ok:
            base = flat.data_ptr()
            for name, (shape, off, _dead) in eng.layout.entries.items():
                p = params[name]
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32 or not p.is_contiguous():
                    ok = False
                    break
        if not ok:
            if dev.type != "cuda":
                raise RuntimeError("brats2019_amd.model.UNet runs on a ROCm GPU only: move the model with .cuda() first "
                                   "(there is no CPU fallback for the HIP path)")
            flat = torch.empty(eng.layout.total, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for name, view in eng.layout.views(flat).items():
                    view.copy_(params[name].data)
                    params[name].data = view
            self.__dict__["_flat_buf"] = flat
        lst = self._param_list()
        self.__dict__["_flat_ends"] = (lst[0].data_ptr(), lst[-1].data_ptr())
        self.__dict__["_flat_checked"] = True
        return flat

    def _replicate_for_data_parallel(self):
        # nn.DataParallel over SEVERAL devices replicates the module per step (main.py:61 with --gpus > 1).  The executor's state (flat
        # parameter buffer, workspace, HIP handle) belongs to one device: data parallelism here is one process per GPU (INTEGRATION.md,
        # "Data parallel").  A single-device wrap -- nn.DataParallel(net, device_ids=[0]), the reference's default --gpus 1, and what its
        # checkpoints pickle -- never replicates and works unchanged.
        raise RuntimeError("brats2019_amd.model.UNet: nn.DataParallel over several devices is replaced by one process per GPU "
                           "(python -m torch.distributed.run --nproc-per-node N ...; INTEGRATION.md 'Data parallel'); "
                           "a single-device wrap (device_ids=[0]) works as in the reference")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_engine_obj", "_flat_buf", "_param_names", "_last_flat_grads", "_flat_checked", "_flat_ends", "_params_cached", "_param_slots"):     # never pickle the ctypes handle / the alias buffer
            state.pop(k, None)
        return state

    def forward(self, x):
        L.require_gpu()
        inp = x[0]                                                # model.py:410
        if not inp.is_cuda:
            raise RuntimeError("brats2019_amd.model.UNet: input must be a ROCm device tensor (HIP-only path)")
        params = self._param_list()
        training = torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in params))
        probs = _UNetFn.apply(self, training, inp, *params)
        return [probs]                                            # model.py:433
