#!/usr/bin/env python
"""Where does the run-to-run variation of dgamma come from? Wraps _NormFn.backward: right after the kernels it clones dgamma / dbeta
(same stream) and evaluates them with torch from the SAME inputs; after the step p.grad is compared with both."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nndetection_amd.plans import get_plan
from tests.gpu_util import det_randperm, synth_inputs
from tests.test_parity_full_gpu import _hip_model, _cuda_targets
from nndetection_amd.arch import conv as CV
from nndetection_amd.layout import phys

plan = get_plan("luna160")
plan["batch_size"] = 1
x, tg = synth_inputs(plan)
net = _hip_model(plan)
torch.randperm = det_randperm
xg, tgg = x.cuda().to(torch.bfloat16), _cuda_targets(tg)
names = {id(m): n for n, m in net.named_modules()}
rec = {}
orig = CV._NormFn.backward


def wrapped(ctx, grad_out, _g=None):
    out = orig(ctx, grad_out, _g)
    mod = ctx.mod
    y_p, mean_rstd, g32, b32 = ctx.saved_tensors
    N, spatial, cout, cout_p = ctx.dims
    if mod.norm_groups == cout and spatial * N < 2.6e6 * 4:
        g_p, _ = phys(grad_out, dtype=y_p.dtype, cp=cout_p)
        mu, rs = mean_rstd[:, :cout, 0].double().view(N, 1, cout), mean_rstd[:, :cout, 1].double().view(N, 1, cout)
        yd = y_p.reshape(N, -1, cout_p)[..., :cout].double()
        xh = (yd - mu) * rs
        z = torch.addcmul(b32.double() - (mu * rs * g32.double()), yd, rs * g32.double())
        gm = g_p.reshape(N, -1, cout_p)[..., :cout].double() * (z > 0)
        rec[names[id(mod)]] = (out[1].clone(), out[2].clone(), (gm * xh).sum((0, 1)).float(), gm.sum((0, 1)).float(), out[1])
    return out


CV._NormFn.backward = staticmethod(wrapped)
for rep in range(3):
    rec.clear()
    net.zero_grad(set_to_none=True)
    losses, _ = net.train_step(xg, tgg, evaluation=False)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    for n, (dg_k, db_k, dg_t, db_t, dg_live) in rec.items():
        p = dict(net.named_parameters())
        pg, pb = p[n + ".norm.weight"].grad, p[n + ".norm.bias"].grad
        s = float(dg_t.abs().max())
        print(f"rep {rep} {n:32s} dgamma: kernel-vs-torch {float((dg_k - dg_t).abs().max()) / s:.2e}  final p.grad-vs-kernel-clone "
              f"{float((pg - dg_k).abs().max()) / s:.2e} | dbeta: kernel-vs-torch {float((db_k - db_t).abs().max()) / float(db_t.abs().max()):.2e} "
              f"final-vs-clone {float((pb - db_k).abs().max()) / float(db_t.abs().max()):.2e}  same storage {pg.data_ptr() == dg_live.data_ptr()}")
