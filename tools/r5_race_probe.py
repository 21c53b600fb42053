#!/usr/bin/env python
"""Probe for the deviation of tests/test_parity_full_gpu.py::test_luna160_absorbed_top_down_step: the step with NNDET_SEG_UP on vs off,
top deviations of the parameter gradients, repeated. Usage: tools/r5_race_probe.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nndetection_amd.plans import get_plan
from tests.gpu_util import det_randperm, synth_inputs
from tests.test_parity_full_gpu import _hip_model, _cuda_targets
from nndetection_amd.arch import segmenter as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lp = torch.bfloat16
plan = get_plan("luna160")
plan["batch_size"] = 1
x, tg = synth_inputs(plan)
net = _hip_model(plan)
torch.randperm = det_randperm
xg, tgg = x.cuda().to(lp), _cuda_targets(tg)
runs = []
for rep in range(reps):
    for up in (True, False):
        S.SEG_UP = up
        net.zero_grad(set_to_none=True)
        losses, _ = net.train_step(xg, tgg, evaluation=False)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        runs.append((up, {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}))


def dev(a, b):
    out = []
    for n, g0 in b.items():
        out.append(((float((a[n] - g0).abs().max()) - 1e-7) / (float(g0.abs().max()) + 1e-12), n))
    return sorted(out, reverse=True)[:4]


ups = [g for u, g in runs if u]
offs = [g for u, g in runs if not u]
print("up vs off (same rep):", [[f"{d:.3f} {n.replace('encoder.stages.', 'e').replace('.convs.0.', '.c')}" for d, n in dev(a, b)] for a, b in zip(ups, offs)])
print("off vs off (rep 0 vs k):", [[f"{d:.4f} {n.replace('encoder.stages.', 'e').replace('.convs.0.', '.c')}" for d, n in dev(offs[k], offs[0])] for k in range(1, reps)])
print("up vs up (rep 0 vs k):", [[f"{d:.4f} {n.replace('encoder.stages.', 'e').replace('.convs.0.', '.c')}" for d, n in dev(ups[k], ups[0])] for k in range(1, reps)])
