#!/usr/bin/env python
"""20 inference steps (luna160, batch 4, bf16) for `rocprofv3 --kernel-trace --stats`: where does the inference step spend its time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nndetection_amd.plans import get_plan
plan = get_plan("luna160")
r = bench.Route(plan, 4, "bf16", torch.device("cuda:0"), 0, False)
net = r.net.eval()
with torch.no_grad():
    for _ in range(20):
        net.inference_step(r.x)
torch.cuda.synchronize()
