#!/usr/bin/env python
"""Can ONE whole training step (forward on 7 streams, losses, backward, fused SGD) be captured in a hipGraph and replayed?
(round 4 feasibility probe: what would replay save over eager launches -- the host already runs ahead of the GPU, so any gain is
GPU-side dispatch latency between the ~390 dependent launches of a step.)

    tools/graph_probe.py [replays=60]
"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd.plans import get_plan

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
plan = get_plan("luna160")
r = bench.Route(plan, 4, "bf16", dev, 0, False)
# deterministic sampler keys: the seed is a host value that a graph would bake in anyway
r.net.head.fg_bg_sampler.deterministic = True


def eager(n):
    for _ in range(8):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"eager: {eager(replays):.3f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(side):
        for _ in range(3):
            r.step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        loss = r.step()
    torch.cuda.synchronize()
    print("captured one step", flush=True)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / replays * 1e3:.3f} ms/step; loss of the last replay {float(loss):.5f}", flush=True)
except Exception as e:                                                        # noqa: BLE001
    print("capture / replay failed:", type(e).__name__, str(e)[:600], flush=True)
    traceback.print_exc(limit=12)
