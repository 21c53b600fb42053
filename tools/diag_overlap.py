"""Diagnostic: three optimizer steps on toy64 with all stream-overlap features on / off; run-to-run differences of the parameters."""
import sys
import torch
from nndetection_amd import _lib as L
from nndetection_amd.arch.heads import DetectionHeadHNMNative
from nndetection_amd.core.retina import BaseRetinaNet
from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model, configure_optimizer
from tests.gpu_util import synth_inputs, det_randperm

torch.randperm = det_randperm
plan = get_plan("toy64")
x, tg = synth_inputs(plan)
dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
tgd = {"target_boxes": [t.cuda() for t in tg["target_boxes"]], "target_classes": [t.cuda() for t in tg["target_classes"]], "target_seg": tg["target_seg"].cuda()}


def run(mode):
    DetectionHeadHNMNative.multi_stream = mode[0]; BaseRetinaNet.overlap_aux = mode[1]; L.wgrad_streams.enabled = mode[2]
    torch.manual_seed(0)
    net = build_model(plan).cuda()
    opt, sched = configure_optimizer(net)
    for g in opt.param_groups:
        g["lr"] = 1e-2
    hist, grads = [], None
    for it in range(3):
        losses, _ = net.train_step(x.cuda().to(dtype), tgd, evaluation=False)
        sum(losses.values()).backward()
        if it == 0:
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        opt.step(); opt.zero_grad(set_to_none=True)
        hist.append([round(float(v.detach()), 6) for v in losses.values()])
    torch.cuda.synchronize()
    return hist, {n: p.detach().clone() for n, p in net.named_parameters()}, grads


def cmp(a, b, what):
    worst = sorted(((float((a[n] - b[n]).abs().max()) / (float(b[n].abs().max()) + 1e-9), n) for n in b), reverse=True)[:4]
    print("   ", what, [(round(w, 6), n) for w, n in worst])


cfgs = {"off": (False, False, False), "on": (True, True, True), "head": (True, False, False), "aux": (False, True, False), "wgrad": (False, False, True)}
base = run(cfgs["off"])
for name, c in [("off", cfgs["off"]), ("on", cfgs["on"]), ("head", cfgs["head"]), ("aux", cfgs["aux"]), ("wgrad", cfgs["wgrad"])]:
    r = run(c)
    print(name, "losses equal:", r[0] == base[0], r[0][2])
    cmp(r[2], base[2], "step-1 grads vs off:")
    cmp(r[1], base[1], "params after 3 steps vs off:")
