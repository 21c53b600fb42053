#!/usr/bin/env python
"""How the step time depends on which hardware queue a side stream lands on.  HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware
queues when they are created / first used; D dummy streams created (and touched) BEFORE the framework creates its own shift that mapping.
  tools/stream_map_probe.py D [prio ...]      D dummies of normal priority, then optional extra dummies of the given priorities
Prints ms per step (60 steps after 15 warm-up, batch 4 bf16 luna160)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
D = int(sys.argv[1]) if len(sys.argv) > 1 else 0
extra = [int(a) for a in sys.argv[2:]]
dev = torch.device("cuda:0")
torch.cuda.init()
keep = []
for pr in [0] * D + extra:
    s = torch.cuda.Stream(device=dev, priority=pr)
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    keep.append(s)
torch.cuda.synchronize()
import bench
from nndetection_amd.plans import get_plan
r = bench.Route(get_plan("luna160"), 4, "bf16", dev, 0, False)
for _ in range(15):
    r.step()
torch.cuda.synchronize()
ts = []
for b in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        r.step()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print(f"dummies {D} + {extra}: " + " / ".join(f"{t:.3f}" for t in ts) + " ms per step")
