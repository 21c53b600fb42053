#!/usr/bin/env python
"""Isolated timing of the fused segmentation-branch kernels (csrc/segbranch.hip) at the luna160 shape:
tools/segbranch_microbench.py [N=4] [D=160] [H=160] [W=96]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndetection_amd import _lib as L

N, D, H, W = [int(a) for a in sys.argv[1:5]] + [4, 160, 160, 96][len(sys.argv) - 1:]
for dt in (torch.bfloat16, torch.float16):
    x = torch.randn((N, D, H, W, 32), device="cuda").to(dt)
    wq = (torch.randn((27, 32), device="cuda") * 0.05).to(dt).contiguous()
    c0 = torch.zeros((1,), device="cuda")
    tgt = (torch.rand((N, D, H, W), device="cuda") > 0.8).to(torch.uint8)
    z = torch.empty((N, D, H, W), device="cuda")
    R = int(L.load().nndet_segbranch_replicas())
    sums = torch.zeros((R, 4), dtype=torch.float64, device="cuda")
    d1 = torch.empty((N, D, H, W), device="cuda", dtype=dt)
    dsum = torch.zeros((R,), dtype=torch.float64, device="cuda")
    co = torch.tensor([1e-7, 1e-4, -1e-4, 2e-4], device="cuda")
    f = lambda: L.call("nndet_segbranch_forward", L._DT[dt], L.ptr(x), N, D, H, W, 32, L.ptr(wq), L.ptr(c0), L.ptr(tgt), L.ptr(z), L.ptr(sums), L.stream())
    b = lambda: L.call("nndet_segbranch_backward", L._DT[dt], L.ptr(z), L.ptr(tgt), z.numel(), L.ptr(co), L.ptr(d1), L.ptr(dsum), L.stream())
    for name, fn, byts in (("fwd", f, x.numel() * 2 + z.numel() * 5), ("bwd", b, z.numel() * 7)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"segbranch {name} {str(dt)[6:]:9s} {N}x{D}x{H}x{W}: {ms:7.3f} ms  {byts / ms / 1e9:7.2f} TB/s algorithmic", flush=True)
