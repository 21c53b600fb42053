#!/usr/bin/env python
"""Training trajectory on ONE fixed batch in fp32 / bf16 / fp16(+GradScaler) from identical weights:
tools/trajectory.py [plan=luna160] [steps=200] [batch=2] [lr=0.01] [warm-up steps=20]  -> one line per 10 steps with the total loss of each run and the
relative deviation of the low-precision runs from fp32. Same loop as tests/test_parity_full_gpu.py::_trajectory."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model, get_params_no_wd_on_norm
from nndetection_amd.optim import SGDNesterov
from tests.gpu_util import det_randperm, synth_inputs


def run(plan, dtype_name, steps, lr, warm=20):
    torch.manual_seed(0)
    net = build_model(plan).cuda()
    x, tg = synth_inputs(plan, seed=4)
    dt = {"f32": torch.float32, "f32b": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype_name]
    if dtype_name == "f32b":                       # a second fp32 run from a 1e-6 relative perturbation of the input: the natural spread
        x = x * (1.0 + 1e-6)
    xg = x.cuda().to(dt)
    tgg = {"target_boxes": [b.cuda() for b in tg["target_boxes"]], "target_classes": [c.cuda() for c in tg["target_classes"]],
           "target_seg": tg["target_seg"].cuda()}
    opt = SGDNesterov(get_params_no_wd_on_norm(net, 3e-5), lr, momentum=0.9, nesterov=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14) if dt == torch.float16 else None
    curve = []
    for it in range(steps):
        for g in opt.param_groups:
            g["lr"] = lr * min(1.0, (it + 1) / float(warm))
        losses, _ = net.train_step(xg, tgg, evaluation=False)
        loss = sum(losses.values())
        curve.append(loss.detach())
        (scaler.scale(loss) if scaler is not None else loss).backward()
        if scaler is not None:
            scaler.step(opt); scaler.update()
        else:
            opt.step()
        opt.zero_grad(set_to_none=True)
    return torch.stack(curve).float().cpu().numpy(), (scaler.get_scale() if scaler else None)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "luna160"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    plan = get_plan(name)
    plan["batch_size"] = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lr = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
    warm = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    torch.randperm = det_randperm
    curves, scale = {}, None
    for dn in ("f32", "f32b", "bf16", "f16"):
        curves[dn], sc = run(plan, dn, steps, lr, warm)
        scale = sc or scale
    sm = lambda c: np.convolve(c, np.ones(10) / 10, mode="valid")
    ref = sm(curves["f32"])
    print(f"# {name} batch {plan['batch_size']}, {steps} steps on one fixed batch, SGD nesterov 0.9, wd 3e-5, lr {lr} after {warm} warm-up steps; "
          f"fp16 with GradScaler (final scale {scale})")
    print("# 10-step means of the total loss; gap = |run - f32| / initial loss; f32b = fp32 from an input perturbed by 1e-6 (natural spread)")
    print("# step   loss_f32  loss_f32b  loss_bf16   loss_f16    gap_f32b  gap_bf16   gap_f16")
    l0 = float(curves["f32"][0])
    for i in range(0, len(ref), 10):
        v = [sm(curves[dn])[i] for dn in ("f32b", "bf16", "f16")]
        print(f"{i:5d}  {ref[i]:9.5f}  " + "  ".join(f"{a:9.5f}" for a in v) + "   " + "  ".join(f"{abs(a - ref[i]) / l0:8.4f}" for a in v))
    for dn in ("f32b", "bf16", "f16"):
        gap = np.abs(sm(curves[dn]) - ref) / l0
        half = lambda c: int(np.argmax(sm(c) < 0.5 * l0))
        print(f"# {dn}: largest gap {gap.max():.4f} of the initial loss at step {int(gap.argmax())}, mean {gap.mean():.4f}; first 10-step mean below half the "
              f"initial loss at step {half(curves[dn])} (fp32 {half(curves['f32'])}); end loss {curves[dn][-10:].mean():.5f} (fp32 {curves['f32'][-10:].mean():.5f})")


if __name__ == "__main__":
    main()
