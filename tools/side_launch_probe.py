#!/usr/bin/env python
"""What do SMALL launches on a side stream cost the training step's main chain? (round 4: composing the segmentation branch's
parameter-only tensors -- ~35 tiny torch launches -- on a side stream UNDER the encoder made the step 0.7-0.9 ms SLOWER than doing it in
place later.) At the start of every forward pass N tiny kernels of a given kind are queued on an otherwise idle side stream:

    tools/side_launch_probe.py [steps=40]   ->  ms per step for N in {0, 32, 64, 128} x kind in {fill, add, mm}
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd.plans import get_plan

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
plan = get_plan("luna160")
r = bench.Route(plan, 4, "bf16", dev, 0, False)
side = torch.cuda.Stream(device=dev)
cfg = {"n": 0, "kind": "fill", "where": "start"}
a = torch.randn(27, 32, device=dev)
b = torch.randn(32, 32, device=dev)
buf = torch.zeros(64, device=dev)
orig_forward = r.net.forward


def tiny():
    k = cfg["kind"]
    for _ in range(cfg["n"]):
        if k == "fill":
            buf.zero_()
        elif k == "add":
            buf.add_(1.0)
        else:
            torch.mm(a, b)


def forward(inp):
    if cfg["n"]:
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            tiny()
    return orig_forward(inp)


r.net.forward = forward


def run():
    for _ in range(8):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    for kind in ("fill", "add", "mm"):
        for n in (0, 32, 64, 128):
            cfg["n"], cfg["kind"] = n, kind
            print(f"rep {rep} kind {kind:4s} n {n:4d}: {run():.3f} ms/step", flush=True)
