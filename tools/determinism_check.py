#!/usr/bin/env python
"""Run-to-run determinism of one training step (pl.Trainer(deterministic=...), scripts/train.py:277): the same
luna160 batch-4 step from the same state R times; every loss and every parameter gradient compared BITWISE with the first run.
Usage: tools/determinism_check.py [--runs 4] [--dtype bf16] [--batch 4]   (env NNDET_DETERMINISTIC=1 selects the ordered reductions)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synth_batch, _TORCH_DT                       # noqa: E402
from nndetection_amd.plans import get_plan                     # noqa: E402
from nndetection_amd.ptmodule import build_model               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--plan", default="luna160")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    plan = get_plan(a.plan)
    torch.manual_seed(0)
    net = build_model(plan).to(dev)
    x, tg = synth_batch(plan, a.batch, _TORCH_DT[a.dtype], dev, seed=1000)
    ref, diffs = None, {}
    for r in range(a.runs):
        net.zero_grad(set_to_none=True)
        torch.manual_seed(1234)                                # the sampler's hash seed
        losses, _ = net.train_step(x, tg, evaluation=False, batch_num=1)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        cur = {"loss." + k: v.detach().float().clone() for k, v in losses.items()}
        cur.update({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
        if ref is None:
            ref = cur
            continue
        for n, v in cur.items():
            if not torch.equal(v, ref[n]):
                d = float((v.double() - ref[n].double()).abs().max() / max(1e-300, float(ref[n].double().abs().max())))
                diffs[n] = max(diffs.get(n, 0.0), d)
    print("deterministic switch: NNDET_DETERMINISTIC=%s; %d runs, %d tensors compared, %d differ from the first run"
          % (os.environ.get("NNDET_DETERMINISTIC", "0"), a.runs, len(ref), len(diffs)))
    for n, d in sorted(diffs.items(), key=lambda kv: -kv[1]):
        print("   %-60s max rel diff %.2e" % (n, d))
    return 0 if not diffs else 1


if __name__ == "__main__":
    sys.exit(main())
