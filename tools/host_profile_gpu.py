#!/usr/bin/env python
"""cProfile of the HOST side of real training steps on the GPU (no synchronisation inside the profiled region: the GPU queue absorbs the
launches): where the ~9-12 ms of host time per step go.  tools/host_profile_gpu.py [steps=30]"""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd.plans import get_plan
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
r = bench.Route(get_plan("luna160"), 4, "bf16", dev, 0, False)
for _ in range(10):
    r.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    r.step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    out = s.getvalue()
    print(f"==== by {key} ({steps} steps; divide by {steps} for per-step)")
    print("\n".join(l[:170] for l in out.splitlines()[4:45]))
