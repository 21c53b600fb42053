#!/usr/bin/env python
"""GPU-side phase times of the training step WITHOUT a profiler (events on the main stream and on the weight-gradient stream):
forward + losses | backward (main chain, joined with the weight-gradient stream) | when the weight-gradient stream finished |
optimizer. tools/phase_times.py [steps=40] [dtype=bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd import _lib as L
from nndetection_amd.plans import get_plan

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dn = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
plan = get_plan("luna160")
r = bench.Route(plan, 4, dn, dev, 0, False)
for _ in range(10):
    r.step()
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
rec = []
marks = {}
_cl = r.net.head.compute_loss
def compute_loss(*a, **k):                      # (events around the detection loss: sampler + loss kernels on the main stream)
    marks["l0"] = ev(); marks["l0"].record()
    out = _cl(*a, **k)
    marks["l1"] = ev(); marks["l1"].record()
    return out
r.net.head.compute_loss = compute_loss
_call = L.call
def call(name, *a):                             # (event in front of the FIRST norm backward of a pass: the head trunks' backward starts)
    if name == "nndet_norm_backward_items" and "nb" not in marks:
        marks["nb"] = ev(); marks["nb"].record()
    return _call(name, *a)
L.call = call
for _ in range(steps):
    e = [ev() for _ in range(5)]
    e[0].record()
    marks.pop("nb", None)
    losses, _ = r.net.train_step(r.x, r.tg, evaluation=False, batch_num=0)
    loss = sum(losses.values())
    e[1].record()
    loss.backward()
    e[2].record()
    ws = L.wgrad_streams.streams.get(0)
    if ws is not None:
        e[4].record(ws)
    r.opt.step(); r.sched.step(); r.opt.zero_grad(set_to_none=True)
    e[3].record()
    rec.append(e + [marks["l0"], marks["l1"], marks.get("nb", e[2])])
torch.cuda.synchronize()
import numpy as np
rows = []
for e in rec[5:]:
    rows.append([e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[1].elapsed_time(e[4]) if ws is not None else float("nan"), e[2].elapsed_time(e[3]),
                 e[0].elapsed_time(e[3]), e[0].elapsed_time(e[5]), e[5].elapsed_time(e[6]), e[6].elapsed_time(e[1]), e[1].elapsed_time(e[7])])
m = np.mean(rows, 0)
print(f"forward+losses {m[0]:.3f} ms | backward (joined) {m[1]:.3f} ms | weight-gradient stream done {m[2]:.3f} ms after the backward started "
      f"(slack of the main chain behind it: {m[1] - m[2]:.3f} ms) | optimizer {m[3]:.3f} ms | step {m[4]:.3f} ms")
print(f"forward split: network + target assignment {m[5]:.3f} ms | detection loss (sampler, sparse regressor conv, loss kernel) {m[6]:.3f} ms | "
      f"join with the segmentation branch + loss sum {m[7]:.3f} ms")
print(f"backward: {m[8]:.3f} ms from its start until the first norm backward of the head trunks is reached on its stream "
      f"(loss backward, sparse head-output backward; the segmentation branch's backward is issued in between)")
