#!/usr/bin/env python
"""Does the mapping of our side streams onto the runtime's hardware queues matter? HIP hands new streams the existing hardware queues
round-robin, so k dummy streams created BEFORE the model's side streams shift which of them share a queue. Usage: stream_pad_probe.py k"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.cuda.init()
pads = [torch.cuda.Stream() for _ in range(k)]
import bench
from nndetection_amd.plans import get_plan

plan = get_plan("luna160")
r = bench.Route(plan, plan["batch_size"], "bf16", torch.device("cuda", 0), 0, False)
dt, _ = r.timed(15, 60)
print(f"pad streams {k}: {dt / 60 * 1e3:.3f} ms per step")
