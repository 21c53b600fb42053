"""Diagnostic: one luna160 train step (batch, dtype from argv) -> losses and which parameter gradients are non-finite."""
import sys
import torch
from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model
from tests.gpu_util import synth_inputs

batch = int(sys.argv[1]); dtype = torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32
plan = get_plan("luna160"); plan["batch_size"] = batch
x, tg = synth_inputs(plan)
torch.manual_seed(0)
net = build_model(plan).cuda()
tgd = {"target_boxes": [t.cuda() for t in tg["target_boxes"]], "target_classes": [t.cuda() for t in tg["target_classes"]],
       "target_seg": tg["target_seg"].cuda()}
for it in range(2):
    net.zero_grad(set_to_none=True)
    losses, _ = net.train_step(x.cuda().to(dtype), tgd, evaluation=False)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f"iter {it} losses", {k: float(v) for k, v in losses.items()}, "non-finite grads:", len(bad), bad[:6], bad[-3:] if len(bad) > 6 else "")
