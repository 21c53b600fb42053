#!/usr/bin/env python
"""Stress of k_ig3s<.., PRE> (nndet_conv3d_forward_norm_input): random volumes / batch sizes / grids, every launch compared bit for bit with
nndet_affine_apply + nndet_conv3d_forward on the same inputs (NaN canaries in both outputs). The kernel orders its LDS traffic with counted
vmcnt waits and one barrier per tile; a timing-dependent slip would show here as a mismatch.  tools/ig3s_pre_stress.py [seconds=60]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nndetection_amd import _lib as L
from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    dtype = torch.bfloat16 if rng.integers(2) else torch.float16
    B = int(rng.integers(1, 5))
    shape = tuple(int(v) for v in (rng.integers(2, 40), rng.integers(2, 48), rng.integers(2, 70)))
    relu = int(rng.integers(2))
    m = ConvInstanceRelu(3, 32, 64, 3, stride=2, padding=1, add_norm=False, add_act=False, bias=bool(rng.integers(2))).to(dev)
    y0 = (torch.randn(B, *shape, 32, device=dev) * 1.7 + 0.2).to(dtype)
    ss = torch.stack((torch.rand(B, 32, device=dev) + 0.5, torch.randn(B, 32, device=dev) * 0.4), -1).contiguous()
    d = _desc(y0, 32, 64, m.k, m.s, m.p, False)
    w0 = _packed(m, 0, m.conv.weight, d, dtype)
    bias = m.conv.bias.detach().float().contiguous() if m.conv.bias is not None else None
    st = L.stream()
    a_ref = torch.empty_like(y0)
    L.call("nndet_affine_apply", L.dtype_code(y0), L.ptr(y0), L.ptr(ss), B, shape[0] * shape[1] * shape[2], 32, relu, L.ptr(a_ref), st)
    out_ref = torch.full((B, d.out_d, d.out_h, d.out_w, 64), float("nan"), device=dev, dtype=dtype)
    os.environ["NNDET_IG3S_WGS"] = "256"
    L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(a_ref), L.ptr(w0), L.ptr(bias), None, L.ptr(out_ref), None, st)
    dp = _desc(y0, 32, 64, m.k, m.s, m.p, False)
    dp.in_affine, dp.in_relu = ss.data_ptr(), relu
    for _ in range(4):
        wgs = int(rng.choice([1, 2, 3, 5, 8, 13, 32, 100, 256]))
        os.environ["NNDET_IG3S_WGS"] = str(wgs)
        a = torch.full_like(y0, float("nan")); out = torch.full_like(out_ref, float("nan"))
        stats = torch.zeros(L.STATS_REPLICAS, B, 64, 2, dtype=torch.float64, device=dev) if rng.integers(2) else None
        L.call("nndet_conv3d_forward_norm_input", ctypes.byref(dp), L.ptr(y0), L.ptr(a), L.ptr(w0), L.ptr(bias), L.ptr(out), L.ptr(stats), st)
        torch.cuda.synchronize()
        ok = torch.equal(a.view(torch.int16), a_ref.view(torch.int16)) and torch.equal(out.view(torch.int16), out_ref.view(torch.int16))
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", dtype, B, shape, relu, wgs, stats is not None, flush=True)
print(f"{n} launches in {time.time() - t0:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
