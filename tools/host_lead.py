#!/usr/bin/env python
"""How far does the host run AHEAD of the GPU at the phase boundaries of the training step? After one synchronisation (host clock and
GPU timeline aligned), N steps are issued without synchronising; at every boundary the host time is noted and an event recorded on the
main stream. lead = (time the GPU reached the event) - (time the host issued it): a lead near zero means the GPU executes that part
as fast as the host can issue it, i.e. the phase behind the boundary is host-bound. Usage: tools/host_lead.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from nndetection_amd.plans import get_plan

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
plan = get_plan("luna160")
r = bench.Route(plan, plan["batch_size"], "bf16", torch.device("cuda", 0), 0, False)
for _ in range(10):
    r.step()
torch.cuda.synchronize()
names = ["step start", "forward + losses issued", "backward issued", "optimizer issued"]
marks = []


def mark():
    ev = torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    ev.record()
    return t, ev


t0, e0 = mark()
for _ in range(steps):
    row = [mark()]
    losses, _ = r.net.train_step(r.x, r.tg, evaluation=False, batch_num=0)
    loss = sum(losses.values())
    row.append(mark())
    loss.backward()
    row.append(mark())
    r.opt.step(); r.sched.step(); r.opt.zero_grad(set_to_none=True)
    row.append(mark())
    marks.append(row)
torch.cuda.synchronize()
print(f"{steps} steps issued without synchronisation; host: {(marks[-1][-1][0] - t0) / steps * 1e3:.3f} ms/step issue, "
      f"GPU: {e0.elapsed_time(marks[-1][-1][1]) / steps:.3f} ms/step")
for sel, label in ((range(2, 6), "steps 2-5"), (range(steps - 8, steps), "last 8 steps")):
    print(f"-- {label}: lead of the host over the GPU (ms) at each boundary, host time spent issuing the phase before it (ms)")
    for k, nm in enumerate(names):
        lead = [e0.elapsed_time(marks[i][k][1]) - (marks[i][k][0] - t0) * 1e3 for i in sel]
        dur = [(marks[i][k][0] - marks[i][k - 1][0]) * 1e3 for i in sel] if k else [(marks[i][0][0] - marks[i - 1][-1][0]) * 1e3 for i in sel]
        print(f"   {nm:28s} lead {min(lead):7.3f} .. {max(lead):7.3f} (median {sorted(lead)[len(lead) // 2]:7.3f})   host {sorted(dur)[len(dur) // 2]:7.3f}")
