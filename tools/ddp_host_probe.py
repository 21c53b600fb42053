#!/usr/bin/env python
"""What the in-place DDP path costs at world size 1 (RCCL initialised, every bucket launched from the hooks, the all-reduce itself a no-op):
GPU time per step and host time to enqueue a step, with and without the reducer, in one process.  torchrun-free: sets up its own rendezvous."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist
import bench
from nndetection_amd.plans import get_plan
from nndetection_amd.ddp import GradAllReducer

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
plan = get_plan("luna160")
res = {}
for name, fac in (("plain", None), ("ddp", lambda net: GradAllReducer(net, force_overlap=True, overlap=True, profile=False))):
    r = bench.Route(plan, 4, "bf16", dev, 0, False, fac)
    for _ in range(12):
        r.step()
    torch.cuda.synchronize()
    g, h = [], []
    for b in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); c0 = time.perf_counter()
        for _ in range(20):
            r.step()
        h.append((time.perf_counter() - c0) / 20 * 1e3)
        e1.record(); torch.cuda.synchronize()
        g.append(e0.elapsed_time(e1) / 20)
    res[name] = (np.mean(g), np.mean(h))
    print(f"{name}: GPU {np.mean(g):.3f} ms / step (min {np.min(g):.3f}), host enqueue {np.mean(h):.3f} ms / step", flush=True)
    if r.ddp is not None:
        r.ddp.close()
    del r
    torch.cuda.empty_cache()
dist.destroy_process_group()
