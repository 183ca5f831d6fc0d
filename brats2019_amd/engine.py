"""Host handle on the whole-network executor of libresunet_hip.so (ru_unet_*): one C call enqueues the
entire UNet.forward (model.py:407-433) or its backward on torch's current HIP stream.

Parameters and gradients are single flat float32 device buffers in reference state_dict() order; the
library reports the layout (names, shapes, offsets, which tensors are dead).  The flat gradient buffer is
what the data-parallel step all-reduces over RCCL (parallel.py)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch

from . import _lib as L

DEFAULT_CFG = dict(depth=4, encoder_layers=[1, 2, 2, 4], decoder_layers=[1, 1, 1, 1],
                   number_of_channels=[16, 32, 64, 128], number_of_outputs=3)     # main.py:56-59


class ParamLayout:
    """Names / shapes / offsets of the flat parameter buffer (pure metadata; needs the library, not a GPU)."""

    def __init__(self, depth, encoder_layers, decoder_layers, number_of_channels, number_of_outputs):
        lib = L.load()
        self.cfg = dict(depth=int(depth), encoder_layers=[int(v) for v in encoder_layers],
                        decoder_layers=[int(v) for v in decoder_layers],
                        number_of_channels=[int(v) for v in number_of_channels], number_of_outputs=int(number_of_outputs))
        arr = lambda v: (C.c_int * len(v))(*v)
        self.handle = lib.ru_unet_create(self.cfg["depth"], arr(self.cfg["encoder_layers"]), arr(self.cfg["decoder_layers"]),
                                         arr(self.cfg["number_of_channels"]), self.cfg["number_of_outputs"])
        if not self.handle:
            raise RuntimeError("ru_unet_create failed: " + L.last_error())
        self.handle = C.c_void_p(self.handle)
        self.entries = OrderedDict()
        for i in range(lib.ru_unet_param_count(self.handle)):
            name = lib.ru_unet_param_name(self.handle, i).decode()
            shape = tuple(lib.ru_unet_param_dim(self.handle, i, d) for d in range(lib.ru_unet_param_ndim(self.handle, i)))
            self.entries[name] = (shape, int(lib.ru_unet_param_offset(self.handle, i)), bool(lib.ru_unet_param_is_dead(self.handle, i)))
        self.total = int(lib.ru_unet_param_total(self.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.load().ru_unet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def views(self, flat):
        """OrderedDict name -> view of `flat` with the parameter's shape."""
        out = OrderedDict()
        for name, (shape, off, _dead) in self.entries.items():
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off:off + n].view(shape)
        return out


class UNetEngine:
    """forward()/backward() on device buffers.  One engine == one in-flight forward state."""

    def __init__(self, depth=4, encoder_layers=(1, 2, 2, 4), decoder_layers=(1, 1, 1, 1),
                 number_of_channels=(16, 32, 64, 128), number_of_outputs=3, precision="f32"):
        self.layout = ParamLayout(depth, encoder_layers, decoder_layers, number_of_channels, number_of_outputs)
        self.h = self.layout.handle
        self.grad_precision = "bf16x3"
        self.set_precision(precision)
        self.n_out = int(number_of_outputs)
        self._ws = None
        self._ws_key = None
        self.generation = 0

    def set_precision(self, precision):
        """"f32": exact-f32 MFMA convolutions; "bf16x3": split-bf16 3-product convolutions (|dp| ~ 5e-5)."""
        L.check(L.load().ru_unet_set_precision(self.h, L.PRECISIONS[precision]), "ru_unet_set_precision")
        self.precision = precision

    def set_grad_precision(self, grad_precision):
        """Arithmetic of the 3x3x3 data / weight gradients under a bf16x3 forward (ru_unet_set_grad_precision): "bf16x3" (default, three
        split-bf16 products) or "bf16" (operands rounded to bf16, one product: BASELINE configs[2]'s "bf16 forward+backward" for the
        backward; the forward and the probabilities do not change)."""
        if grad_precision not in L.GRAD_PRECISIONS:
            raise ValueError("grad_precision must be one of %s" % sorted(L.GRAD_PRECISIONS))
        L.check(L.load().ru_unet_set_grad_precision(self.h, L.GRAD_PRECISIONS[grad_precision]), "ru_unet_set_grad_precision")
        self.grad_precision = grad_precision

    def set_fusion(self, gn_bwd_stats=True, gn_bwd_apply=True, side_stream=True, batch_wreduce=True, tail_finalize=False, pw_dgrad=True):
        """Fusions of the voxel-major engine (ru_unet_set_fusion; tests switch them off to hold the fused kernels to the separate
        passes).  batch_wreduce: one launch sums the partials of every weight gradient of a backward pass; tail_finalize (opt-in: measured
        slower than the finalize launches it replaces): GroupNorm statistics / backward coefficients are finalized
        by the last workgroup of the producing kernel; pw_dgrad: the decoder's concat 1x1 weight-gradient kernel also forms its data gradient."""
        mask = ((L.FUSE_GN_BWD_STATS if gn_bwd_stats else 0) | (L.FUSE_GN_BWD_APPLY if gn_bwd_apply else 0) | (L.FUSE_SIDE_STREAM if side_stream else 0) |
                (L.FUSE_BATCH_WREDUCE if batch_wreduce else 0) | (L.FUSE_TAIL_FINALIZE if tail_finalize else 0) | (L.FUSE_PW_DGRAD if pw_dgrad else 0))
        L.check(L.load().ru_unet_set_fusion(self.h, mask), "ru_unet_set_fusion")
        self._ws_key = None                    # the backward's workspace layout depends on it

    FAMILIES = ("conv3_l0", "conv3_deep", "wgrad3_l0", "wgrad3_deep", "groupnorm", "pointwise_1x1_s2_up", "other")

    def probe(self, enable=True):
        """In-situ timing (ru_unet_probe): True / 1 = HIP event pairs around the 16->16 3x3x3 convolutions of the forward (the dominant
        kernel); 2 = around EVERY launch, booked per kernel family (probe_read_families)."""
        L.check(L.load().ru_unet_probe(self.h, int(enable)), "ru_unet_probe")

    # single kernel instantiations the executor tags (engine.hip INST_*): rows BEHIND the families in ru_unet_probe_read_families
    INSTANCES = ("conv16_fwd", "conv16_dgrad", "conv_deep_fwd", "conv_deep_dgrad", "wgrad16_fused_apply", "wgrad16_plain", "wgrad_deep")

    def probe_read_families(self, instances=False):
        """-> {family: (total_ms, launches)} since the last read (ru_unet_probe_read_families); waits for the recorded events.
        instances=True: (families, {instance: (total_ms, launches)}) -- the tagged launches are booked to their family AND to their instance."""
        import ctypes
        names = self.FAMILIES + (self.INSTANCES if instances else ())
        n = len(names)
        ms, cnt = (ctypes.c_double * n)(), (ctypes.c_int * n)()
        L.check(L.load().ru_unet_probe_read_families(self.h, ms, cnt, n), "ru_unet_probe_read_families")
        got = {f: (ms[i], cnt[i]) for i, f in enumerate(names)}
        if not instances:
            return got
        return {f: got[f] for f in self.FAMILIES}, {f: got[f] for f in self.INSTANCES}

    def probe_read(self):
        """-> (total_ms, launches) since the last read; waits for the recorded events (ru_unet_probe_read)."""
        import ctypes
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        L.check(L.load().ru_unet_probe_read(self.h, ctypes.byref(ms), ctypes.byref(n)), "ru_unet_probe_read")
        return ms.value, n.value

    def freeze_params(self, frozen=True):
        """Inference with constant weights: the packed weights in the workspace are built once and reused (ru_unet_freeze_params).
        Leave off while training -- the optimizer rewrites the flat parameter buffer in place every step."""
        frozen = bool(frozen)
        if frozen and getattr(self, "_frozen", False):
            return                                # already frozen: keep the cached packs
        L.check(L.load().ru_unet_freeze_params(self.h, int(frozen)), "ru_unet_freeze_params")
        self._frozen = frozen

    def _workspace(self, n, d, h, w, training, device):
        key = (n, d, h, w, bool(training), str(device), self.precision)      # the two precisions keep different scratch tensors
        if self._ws is None or self._ws_key != key:
            nbytes = L.load().ru_unet_workspace_bytes(self.h, n, d, h, w, int(training))
            if nbytes == 0:
                raise RuntimeError("ru_unet_workspace_bytes: " + L.last_error())
            self._ws = None                      # release the old arena before allocating the new one
            self._ws = L.workspace(nbytes, device)
            self._ws_key = key
        return self._ws

    def forward(self, flat_params, x, training=False):
        """x: [N,4,D,H,W] float32 device tensor -> sigmoid probabilities [N,n_out,D,H,W]."""
        L.require_gpu()
        if x.dim() != 5 or int(x.shape[1]) != 4:
            raise ValueError("expected input [N,4,D,H,W] (model.py:336), got %s" % (tuple(x.shape),))
        if flat_params.numel() != self.layout.total or flat_params.dtype != torch.float32:
            raise ValueError("flat parameter buffer has the wrong size/dtype")
        x = x.contiguous().float()
        n, _, d, h, w = [int(v) for v in x.shape]
        ws = self._workspace(n, d, h, w, training, x.device)
        probs = torch.empty((n, self.n_out, d, h, w), dtype=torch.float32, device=x.device)
        L.check(L.load().ru_unet_forward(self.h, L.f32(flat_params), L.f32(x), L.f32(probs), n, d, h, w, int(training),
                                         L.ptr(ws), ws.numel(), L.stream()), "ru_unet_forward")
        self.generation += 1
        self._x = x                               # keep the input alive until backward (wgrad of conv_input reads it)
        self._probs = probs if training else None  # ... and the probabilities (the sigmoid backward reads the caller's buffer)
        return probs

    def backward(self, flat_params, dprobs, flat_grads=None, want_dx=False):
        """dprobs = d(loss)/d(probs) -> flat gradient buffer (overwritten; dead parameters get zeros)."""
        if flat_grads is None:
            flat_grads = torch.empty_like(flat_params)
        dprobs = dprobs.contiguous().float()
        dx = torch.empty_like(self._x) if want_dx else None
        L.check(L.load().ru_unet_backward(self.h, L.f32(flat_params), L.f32(dprobs), L.f32(flat_grads), L.ptr(dx, True), L.stream()),
                "ru_unet_backward")
        return (flat_grads, dx) if want_dx else flat_grads

    def backward_criterion(self, flat_params, target, sums, count, w_dice, w_bce, bg_weight, priority, flat_grads=None):
        """loss.backward() of the Dice/BCE criterion through the network in one call (ru_unet_backward_criterion): the criterion's gradient
        (second phase of ops.criterion_grad) is formed inside the head's sigmoid-backward pass -- d(loss)/d(probs) is never written."""
        if flat_grads is None:
            flat_grads = torch.empty_like(flat_params)
        if self._probs is None or tuple(target.shape) != tuple(self._probs.shape):
            raise ValueError("backward_criterion: target must have the shape of the probabilities of the preceding training forward")
        target = target.contiguous().float()
        L.check(L.load().ru_unet_backward_criterion(self.h, L.f32(flat_params), L.f32(target), L.ptr(sums), float(count), float(w_dice), float(w_bce),
                                                    float(bg_weight), float(priority), L.f32(flat_grads), None, L.stream()), "ru_unet_backward_criterion")
        return flat_grads

    def gn_stats(self):
        """[(mean[N*8], rstd[N*8])] of every GroupNorm of the last forward, in execution order."""
        lib = L.load()
        cnt = lib.ru_unet_gn_stats(self.h, -1, None, None, L.stream())
        n = int(self._x.shape[0])
        out = []
        for i in range(cnt):
            m = torch.empty(n * 8, dtype=torch.float32, device=self._x.device)
            r = torch.empty_like(m)
            L.check(lib.ru_unet_gn_stats(self.h, i, L.f32(m), L.f32(r), L.stream()), "ru_unet_gn_stats")
            out.append((m, r))
        return out
