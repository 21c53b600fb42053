"""Lean SGD(nesterov) + LinearWarmupPolyLR for the training step.

Same update rule as `torch.optim.SGD(momentum, nesterov, weight_decay)` as configured by the reference
(nndet/ptmodule/retinaunet/base.py:300-336) and the same per-iteration schedule as `LinearWarmupPolyLR`
(nndet/training/learning_rate.py:26-57,127-184), but issued as a handful of `torch._foreach_*` launches over static
tensor lists: torch.optim's per-step Python bookkeeping cost ~5 ms of GPU idle time per step after backward
(profiles/round1_v2_kernel_stats.txt gap analysis), which is 10 % of the whole step at MI355X speed.
"""
from typing import List

import torch


class SGDNesterov:
    def __init__(self, param_groups: List[dict], lr: float, momentum: float = 0.9, nesterov: bool = True):
        self.param_groups = []
        for g in param_groups:
            params = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append({"params": params, "weight_decay": float(g.get("weight_decay", 0.0)), "lr": lr})
        self.momentum, self.nesterov = momentum, nesterov
        self.defaults = {"lr": lr, "momentum": momentum, "nesterov": nesterov}
        self._buf = {}

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            ps = [p for p in g["params"] if p.grad is not None]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            wd, lr, mu = g["weight_decay"], g["lr"], self.momentum
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
