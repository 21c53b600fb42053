#!/usr/bin/env python
"""The fused stem block's backward (k_stem_bwd3) alone and NEXT TO the full-resolution 32 -> 32 weight gradient on a second stream
(what the end of the backward pass looks like), for the two point-tile depths (NNDET_STEM_BWD_TD = 2 / 4).
tools/stem_microbench.py [iters=20]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndetection_amd import _lib as L
from nndetection_amd.arch.conv import ConvInstanceRelu, _desc
from nndetection_amd.layout import cpad

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt, dev = torch.bfloat16, torch.device("cuda:0")
B, sp = 4, (160, 160, 96)
stem = ConvInstanceRelu(3, 1, 32, 3, stride=1, padding=1).to(dev)
x = torch.randn(B, *sp, 1, device=dev).to(dt)
d = _desc(x, 1, 32, stem.k, stem.s, stem.p, False)
w32 = stem.conv.weight.detach().float().contiguous()
g32, b32 = stem.norm.weight.detach().float().contiguous(), stem.norm.bias.detach().float().contiguous()
out = torch.empty(B, *sp, 32, dtype=dt, device=dev)
stats = torch.zeros(L.STATS_REPLICAS, B, 32, 2, dtype=torch.float64, device=dev)
mr = torch.empty(B, 32, 2, dtype=torch.float32, device=dev)
L.call("nndet_stem_block_forward", ctypes.byref(d), L.ptr(x), L.ptr(w32), L.ptr(g32), L.ptr(b32), 1e-5, 1, L.ptr(out), L.ptr(stats), L.ptr(mr), L.stream())
g = torch.randn_like(out)
dw, dga, dbe = torch.zeros(32 * 27, device=dev), torch.zeros(32, device=dev), torch.zeros(32, device=dev)
wsb = L.load().nndet_stem_block_backward_workspace_bytes(ctypes.byref(d))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
# the companion: weight gradient of the full-resolution 32 -> 32 convolution
m = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False).to(dev)
x2 = torch.randn(B, *sp, 32, device=dev).to(dt)
d2 = _desc(x2, 32, 32, m.k, m.s, m.p, False)
dy2 = torch.randn_like(x2)
dw2 = torch.zeros_like(m.conv.weight)
wsb2 = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(d2))
ws2 = torch.empty(max(wsb2, 16), dtype=torch.uint8, device=dev)


def stem_bwd(raw):
    L.call("nndet_stem_block_backward", ctypes.byref(d), L.ptr(x), L.ptr(g), L.ptr(w32), L.ptr(mr), L.ptr(g32), L.ptr(b32), 1, L.ptr(dw), L.ptr(dga),
           L.ptr(dbe), L.ptr(ws), wsb, raw)


def wgrad(raw):
    L.call("nndet_conv3d_backward_weight", ctypes.byref(d2), L.ptr(x2), L.ptr(dy2), L.ptr(dw2), None, L.ptr(ws2), wsb2, raw)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    wgrad(side.cuda_stream)
    stem_bwd(L.stream())
    main.wait_stream(side)


print(f"wgrad 32->32 full resolution alone: {timed(lambda: wgrad(L.stream())):.3f} ms")
for td in ("4", "2", "4", "2"):
    os.environ["NNDET_STEM_BWD_TD"] = td
    print(f"TD={td}: stem backward alone {timed(lambda: stem_bwd(L.stream())):.3f} ms | next to the weight gradient (both done) {timed(both):.3f} ms", flush=True)
