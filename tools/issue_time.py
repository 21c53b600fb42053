#!/usr/bin/env python
"""Host issue time vs GPU time of the training step: the host loop is timed WITHOUT synchronising (time until step() returns) and
the whole run with one synchronisation at the end. issue < total: the GPU is the bottleneck and the host runs ahead;
issue ~ total: the step is host-bound (Python + torch dispatch + ctypes). Usage: tools/issue_time.py [steps] [--plugin]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from nndetection_amd.plans import get_plan

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
plan = get_plan("luna160")
r = bench.Route(plan, plan["batch_size"], "bf16", torch.device("cuda", 0), 0, "--plugin" in sys.argv)
for _ in range(10):
    r.step()
torch.cuda.synchronize()
iss = []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    r.step()
    iss.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
iss.sort()
print(f"steps {steps}: host loop returned after {t_issue / steps * 1e3:.3f} ms/step, GPU done after {t_all / steps * 1e3:.3f} ms/step; "
      f"per-step issue time min {iss[0] * 1e3:.3f} median {iss[len(iss) // 2] * 1e3:.3f} max {iss[-1] * 1e3:.3f} ms")
# the same with the GPU kept out of the way: host cost alone (kernels still launched, but we synchronise BEFORE each step so the
# queue is empty and nothing ever blocks on a full queue) -- an upper bound of the pure issue cost
iss2 = []
for _ in range(10):
    torch.cuda.synchronize()
    a = time.perf_counter()
    r.step()
    iss2.append(time.perf_counter() - a)
torch.cuda.synchronize()
iss2.sort()
print(f"issue time with an empty queue: min {iss2[0] * 1e3:.3f} median {iss2[5] * 1e3:.3f} ms")
