"""Drop-in for `torch.optim.Adam` in the reference's training call (main.py:133-137: `optimizer=optim.Adam`, lr 2e-5, weight_decay
1e-6, amsgrad=True; instantiated by `Trainer.train`, train.py:82-83), running the update as the HIP kernel `ru_adam_step`.

`Adam(params, lr, betas, eps, weight_decay, amsgrad)` is a `torch.optim.Optimizer`: param groups, `zero_grad()`, LR schedulers
(`StepLR` rewrites `group["lr"]`) and `state_dict()` / `load_state_dict()` behave as torch's -- the state keeps torch's own layout
(`state[p] = {"step", "exp_avg", "exp_avg_sq"[, "max_exp_avg_sq"]}`), so a checkpoint written by either optimizer resumes under the
other (train.py:92-94, 320-324).  What differs is where the state lives and how many launches a step takes: the moment tensors of a
group are views of THREE flat buffers laid out like the parameters' own flat buffer (model.UNet aliases every nn.Parameter to one
float32 buffer; the executor writes every gradient into one bucket), so `step()` is one kernel launch per contiguous RUN of
parameters that have a gradient -- three for the shipped network (the never-executed deepest decoder stage keeps grad=None,
model.py:420, and is skipped exactly as torch.optim.Adam skips it) -- instead of ~15 foreach kernels over 86 tensors.
Parameters or gradients that are not laid out that way (any other model) take one launch per tensor.  HIP-only: CPU tensors raise.
"""
from __future__ import annotations

import torch

from . import _lib as L


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unsupported):
        for k, v in unsupported.items():
            # accepted-and-ignored when they have torch's defaults; anything that would change the arithmetic is refused
            if k in ("foreach", "fused", "capturable", "differentiable", "maximize", "decoupled_weight_decay") and not v:
                continue
            raise NotImplementedError("brats2019_amd.optim.Adam: option %s=%r is not implemented by the HIP kernel" % (k, v))
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: %r" % (lr,))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: %r" % (eps,))
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("Invalid beta parameters: %r" % (betas,))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: %r" % (weight_decay,))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._flat = {}            # group index -> flat-state plan

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("amsgrad", False)
        self._flat = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # deep-copies the moments onto the parameters' device: the aliasing is re-made lazily
        self._flat = {}

    # ------------------------------------------------------------------ flat state
    _KEYS = ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")

    def _plan(self, gi, group):
        """Lay the group's moments out like its parameters: if every parameter is a contiguous float32 slice of one allocation span,
        the moments become views of flat buffers covering the same span at the same offsets (existing state is copied in)."""
        plan = self._flat.get(gi)
        params = [p for p in group["params"]]
        sig = (bool(group["amsgrad"]),) + tuple((p.data_ptr(), p.dtype, p.device) for p in params)     # (a flipped amsgrad flag needs the third buffer)
        if plan is not None and plan["sig"] == sig:
            return plan
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("brats2019_amd.optim.Adam: parameters must live on a ROCm GPU (HIP-only path; no CPU fallback)")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("brats2019_amd.optim.Adam: contiguous float32 parameters only")
        keys = self._KEYS if group["amsgrad"] else self._KEYS[:2]
        lo = min(p.data_ptr() for p in params)
        hi = max(p.data_ptr() + 4 * p.numel() for p in params)
        span = (hi - lo) // 4
        dev = params[0].device
        total = sum(p.numel() for p in params)
        flat_ok = span <= 2 * total + 1024 and all(p.device == dev for p in params)       # one allocation span, not scattered tensors
        bufs = {k: torch.zeros(span if flat_ok else 0, dtype=torch.float32, device=dev) for k in keys}
        order = sorted(range(len(params)), key=lambda i: params[i].data_ptr())
        plan = dict(sig=sig, lo=lo, span=span, flat=flat_ok, bufs=bufs, order=order, keys=keys)
        for p in params:                     # state that exists already (load_state_dict, an earlier plan) moves into the flat buffers
            if p in self.state and self.state[p]:
                self._init_state(plan, p)
        self._flat[gi] = plan
        return plan

    def _init_state(self, plan, p):
        """torch.optim.Adam's lazy state initialisation (zeros, step 0) -- with the moments as views of the group's flat buffers."""
        st = self.state[p]
        off = (p.data_ptr() - plan["lo"]) // 4
        for k in plan["keys"]:
            old = st.get(k)
            if plan["flat"]:
                view = plan["bufs"][k][off:off + p.numel()].view_as(p)
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old.to(device=p.device, dtype=torch.float32))
                st[k] = view
            elif old is None or not (old.is_cuda and old.dtype == torch.float32 and old.is_contiguous()):
                st[k] = torch.zeros_like(p, memory_format=torch.contiguous_format) if old is None else old.to(device=p.device, dtype=torch.float32).contiguous()
        if "step" not in st:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)              # torch.optim.Adam's default: a CPU scalar tensor
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        stream = L.stream()
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            plan = self._plan(gi, group)
            ams = bool(group["amsgrad"])
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr = float(lr.item()) if isinstance(lr, torch.Tensor) else float(lr)
            # runs: consecutive (in memory) parameters with a gradient whose gradients and moments are consecutive at the same stride
            runs = []      # [w_ptr, g_ptr, m_ptr, v_ptr, vmax_ptr, numel, step]
            for i in plan["order"]:
                p = params[i]
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(device=p.device, dtype=torch.float32).contiguous()
                    p.grad = g
                st = self.state[p]
                if "step" not in st or any(k not in st for k in plan["keys"]):      # lazy init; also a state loaded without `max_exp_avg_sq` under amsgrad
                    st = self._init_state(plan, p)
                stp = st["step"]
                k = int(stp.item() if isinstance(stp, torch.Tensor) else stp) + 1
                if isinstance(stp, torch.Tensor):
                    stp += 1
                else:
                    st["step"] = k
                n = p.numel()
                ptrs = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        st["max_exp_avg_sq"].data_ptr() if ams else 0)
                if runs:
                    r = runs[-1]
                    if r[6] == k and all(r[j] + 4 * r[5] == ptrs[j] for j in range(4 + int(ams))):
                        r[5] += n
                        continue
                runs.append([ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], n, k])
            for w, g, m, v, vm, n, k in runs:
                L.check(lib.ru_adam_step(w, g, m, v, vm if ams else None, n, lr, float(b1), float(b2), float(group["eps"]),
                                         float(group["weight_decay"]), k, stream), "ru_adam_step")
            self.__dict__["last_launches"] = len(runs)
        return loss
