#!/usr/bin/env python
"""Training trajectory on ONE fixed batch in fp32 / bf16 / fp16(+GradScaler) from identical weights:
tools/trajectory.py [plan=luna160] [steps=200] [batch=2]  -> one line per 10 steps with the total loss of each run and the
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


def run(plan, dtype_name, steps, lr):
    torch.manual_seed(0)
    net = build_model(plan).cuda()
    x, tg = synth_inputs(plan, seed=4)
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype_name]
    xg = x.cuda().to(dt)
    tgg = {"target_boxes": [b.cuda() for b in tg["target_boxes"]], "target_classes": [c.cuda() for c in tg["target_classes"]],
           "target_seg": tg["target_seg"].cuda()}
    opt = SGDNesterov(get_params_no_wd_on_norm(net, 3e-5), lr, momentum=0.9, nesterov=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14) if dt == torch.float16 else None
    curve = []
    for it in range(steps):
        for g in opt.param_groups:
            g["lr"] = lr * min(1.0, (it + 1) / 20.0)
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
    torch.randperm = det_randperm
    curves, scale = {}, None
    for dn in ("f32", "bf16", "f16"):
        curves[dn], sc = run(plan, dn, steps, 0.01)
        scale = sc or scale
    sm = lambda c: np.convolve(c, np.ones(10) / 10, mode="valid")
    ref = sm(curves["f32"])
    print(f"# {name} batch {plan['batch_size']}, {steps} steps on one fixed batch, SGD nesterov 0.9, wd 3e-5, lr 0.01 after 20 warm-up steps; "
          f"fp16 with GradScaler (final scale {scale})")
    print("# step   loss_f32   loss_bf16   loss_f16   (10-step means)   dev_bf16  dev_f16")
    for i in range(0, len(ref), 10):
        b, h = sm(curves["bf16"])[i], sm(curves["f16"])[i]
        print(f"{i:5d}  {ref[i]:9.5f}  {b:9.5f}  {h:9.5f}   {abs(b - ref[i]) / max(ref[i], 0.05):8.4f} {abs(h - ref[i]) / max(ref[i], 0.05):8.4f}")
    for dn in ("bf16", "f16"):
        dev = np.abs(sm(curves[dn]) - ref) / np.maximum(ref, 0.05)
        print(f"# {dn}: max deviation {dev.max():.4f} at step {int(dev.argmax())}, mean {dev.mean():.4f}; end loss {curves[dn][-10:].mean():.5f} (fp32 {curves['f32'][-10:].mean():.5f})")


if __name__ == "__main__":
    main()
