"""Lean SGD(nesterov) + LinearWarmupPolyLR for the training step.

Same update rule as `torch.optim.SGD(momentum, nesterov, weight_decay)` as configured by the reference
(nndet/ptmodule/retinaunet/base.py:300-336) and the same per-iteration schedule as `LinearWarmupPolyLR`
(nndet/training/learning_rate.py:26-57,127-184), but issued as a handful of `torch._foreach_*` launches over static
tensor lists: torch.optim's per-step Python bookkeeping cost ~5 ms of GPU idle time per step after backward
(profiles/round1_v2_kernel_stats.txt gap analysis), which is 10 % of the whole step at MI355X speed.
"""
import os
from typing import List

import torch


def _bump_versions(tensors) -> None:
    """`torch._fused_sgd_` writes the parameters WITHOUT incrementing their autograd version counters (measured on torch 2.10, CPU and
    ROCm; the `_foreach_*` ops do increment them). Everything that caches something derived from a parameter keys it on `_version` --
    here the packed / cast convolution weights (arch/conv.py: _packed) -- so the counters are advanced by hand after the launch."""
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        try:
            setter(tuple(tensors), tuple(t._version + 1 for t in tensors))
            return
        except TypeError:                                      # a torch build whose private setter takes (Tensor, int)
            try:
                for t in tensors:
                    setter(t, t._version + 1)
                return
            except TypeError:
                pass
    torch._foreach_add_(list(tensors), 0.0)                    # an in-place no-op that does bump


class SGDNesterov:
    def __init__(self, param_groups: List[dict], lr: float, momentum: float = 0.9, nesterov: bool = True):
        self.param_groups = []
        for g in param_groups:
            params = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append({"params": params, "weight_decay": float(g.get("weight_decay", 0.0)), "lr": lr})
        self.momentum, self.nesterov = momentum, nesterov
        self.defaults = {"lr": lr, "momentum": momentum, "nesterov": nesterov}
        self._buf = {}

    fused = os.environ.get("NNDET_FUSED_SGD", "1") != "0"   # one multi-tensor launch per group (torch._fused_sgd_, the kernel behind torch.optim.SGD(fused=True))

    # Loss scaling without a host synchronisation (fp16 activations, the reference's precision=16): torch.amp.GradScaler.step() hands
    # an optimizer with this flag the scale and the found-inf flag as DEVICE tensors (attributes `grad_scale` / `found_inf`) instead of
    # reading found_inf on the host; the fused kernel divides the gradients by the scale and skips the update when an inf / nan was
    # found. (The stock path -- torch.optim.SGD without fused=True, what the reference configures -- costs one `.item()` per step.)
    _step_supports_amp_scaling = True

    @torch.no_grad()
    def step(self):
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        for g in self.param_groups:
            ps = [p for p in g["params"] if p.grad is not None]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            wd, lr, mu = g["weight_decay"], g["lr"], self.momentum
            if (self.fused and mu != 0.0 and hasattr(torch, "_fused_sgd_")
                    and all(p.is_cuda and p.dtype == torch.float32 and gr.dtype == torch.float32 for p, gr in zip(ps, grads))):
                # weight decay, momentum, Nesterov and the update in ONE pass over (p, g, buf) instead of 4-5 foreach passes
                new = {id(p) for p in ps if p not in self._buf}
                for first in (True, False):
                    sel = [(p, gr) for p, gr in zip(ps, grads) if (id(p) in new) == first]
                    if not sel:
                        continue
                    if first:
                        for p, gr in sel:
                            self._buf[p] = torch.empty_like(gr)               # filled by the kernel: buf = g (+ wd * p)
                    torch._fused_sgd_([p for p, _ in sel], [gr.contiguous() for _, gr in sel], [self._buf[p] for p, _ in sel],
                                      weight_decay=wd, momentum=mu, lr=lr, dampening=0.0, nesterov=self.nesterov, maximize=False,
                                      is_first_step=first, grad_scale=grad_scale, found_inf=found_inf)
                _bump_versions(ps)
                continue
            if found_inf is not None and bool(found_inf.item()):      # foreach fallback (CPU / NNDET_FUSED_SGD=0): host decision
                return
            if grad_scale is not None:
                grads = torch._foreach_div(grads, float(grad_scale.item()))
            if wd != 0.0:
                grads = torch._foreach_add(grads, ps, alpha=wd)          # g + wd * p (new tensors, p.grad untouched)
            if mu != 0.0:
                new_ids = {id(p) for p in ps if p not in self._buf}
                for p, gr in zip(ps, grads):                              # first step: buf = g (torch.optim.SGD semantics)
                    if id(p) in new_ids:
                        self._buf[p] = gr.clone()
                old = [(p, gr) for p, gr in zip(ps, grads) if id(p) not in new_ids]
                if old:
                    bufs_old = [self._buf[p] for p, _ in old]
                    torch._foreach_mul_(bufs_old, mu)
                    torch._foreach_add_(bufs_old, [gr for _, gr in old])
                bufs = [self._buf[p] for p in ps]
                if self.nesterov:
                    grads = torch._foreach_add(grads, bufs, alpha=mu)
                else:
                    grads = bufs
            torch._foreach_add_(ps, grads, alpha=-lr)

    # ---- checkpointing, laid out like torch.optim.SGD.state_dict(): {"state": {param index: {"momentum_buffer": t}},
    # "param_groups": [{"lr", "momentum", "nesterov", "weight_decay", "params": [indices]}]} (indices run over the groups in order),
    # so a checkpoint written by torch.optim.SGD with the same parameter groups loads here and vice versa.
    def state_dict(self) -> dict:
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                if p in self._buf:
                    state[idx] = {"momentum_buffer": self._buf[p]}
                ids.append(idx); idx += 1
            groups.append({"lr": g["lr"], "momentum": self.momentum, "dampening": 0, "nesterov": self.nesterov,
                           "weight_decay": g["weight_decay"], "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd: dict) -> None:
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has different parameter groups")
        flat = [p for g in self.param_groups for p in g["params"]]
        self._buf = {}
        for k, st in sd["state"].items():
            buf = st.get("momentum_buffer")
            if buf is not None:
                p = flat[int(k)]
                self._buf[p] = buf.detach().to(device=p.device, dtype=p.dtype).clone()
        for g, src in zip(self.param_groups, groups):
            g["lr"], g["weight_decay"] = float(src["lr"]), float(src["weight_decay"])
        if groups:
            self.momentum, self.nesterov = float(groups[0]["momentum"]), bool(groups[0]["nesterov"])

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()


class LinearWarmupPolyLR:
    """lr(k) with k = number of scheduler steps + 1, exactly the reference's `_step_count` arithmetic."""

    def __init__(self, optimizer: SGDNesterov, warm_iterations: int, warm_lr: float, poly_gamma: float, num_iterations: int):
        self.opt, self.warm, self.warm_lr, self.gamma, self.total = optimizer, warm_iterations, warm_lr, poly_gamma, num_iterations
        self.base_lr = optimizer.defaults["lr"]
        self._step_count = 0
        self.step()

    def get_lr(self) -> float:
        k = self._step_count
        if k - 1 < self.warm:
            return self.warm_lr + (self.base_lr - self.warm_lr) * (float(k) / float(self.warm))
        return self.base_lr * (1 - (k - self.warm) / float(self.total - self.warm)) ** self.gamma

    def step(self):
        self._step_count += 1
        lr = self.get_lr()
        for g in self.opt.param_groups:
            g["lr"] = lr

    def state_dict(self) -> dict:
        """`_step_count` / `last_epoch` as torch's _LRScheduler keeps them (last_epoch = _step_count - 1)."""
        return {"_step_count": self._step_count, "last_epoch": self._step_count - 1, "base_lrs": [self.base_lr]}

    def load_state_dict(self, sd: dict) -> None:
        self._step_count = int(sd["_step_count"])
        lr = self.get_lr()
        for g in self.opt.param_groups:
            g["lr"] = lr
