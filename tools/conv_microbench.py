#!/usr/bin/env python
"""Micro-benchmark of single convolution problems of the RetinaUNet (forward / data gradient / weight gradient),
HIP-event timed. Usage: tools/conv_microbench.py [name ...]   (default: the layer shapes that dominate config 2)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nndetection_amd import _lib as L
from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
from nndetection_amd.layout import cpad

# name: (cin, cout, k, s, p, transposed, spatial, batch)
SHAPES = {
    "e0_32x32_full": (32, 32, 3, 1, 1, False, (160, 160, 96), 4),
    "e1_64x64": (64, 64, 3, 1, 1, False, (80, 80, 48), 4),
    "p2_128x128": (128, 128, 3, 1, 1, False, (40, 40, 24), 4),
    "e1_32to64_s2": (32, 64, 3, 2, 1, False, (160, 160, 96), 4),
    "up_p1_64to32": (64, 32, 2, 2, 0, True, (80, 80, 48), 4),
    "lat_p0_1x1": (32, 32, 1, 1, 0, False, (160, 160, 96), 4),
    "head_reg_out": (128, 162, 3, 1, 1, False, (40, 40, 24), 4),
    "e3_256x256": (256, 256, 3, 1, 1, False, (20, 20, 12), 4),
    "e2_64to128_s2": (64, 128, 3, 2, 1, False, (80, 80, 48), 4),
    "e3_128to256_s2": (128, 256, 3, 2, 1, False, (40, 40, 24), 4),
    "up_p2_128to64": (128, 64, 2, 2, 0, True, (40, 40, 24), 4),
    "lat_p1_1x1": (64, 64, 1, 1, 0, False, (80, 80, 48), 4),
    "p3_128x128": (128, 128, 3, 1, 1, False, (20, 20, 12), 4),
    "p4_128x128": (128, 128, 3, 1, 1, False, (10, 10, 6), 4),
    "e4_320x320": (320, 320, 3, 1, 1, False, (10, 10, 6), 4),
    "e5_320x320": (320, 320, 3, 1, 1, False, (5, 5, 6), 4),
    "e4_320x320_b1": (320, 320, 3, 1, 1, False, (10, 10, 6), 1),            # scaling probes of the small-level kernel
    "e4_64to320": (64, 320, 3, 1, 1, False, (10, 10, 6), 4),
    "e4_320to64": (320, 64, 3, 1, 1, False, (10, 10, 6), 4),
    "e4_320x320_444": (320, 320, 3, 1, 1, False, (4, 4, 4), 4),
    "e4_256to320_s2": (256, 320, 3, 2, 1, False, (20, 20, 12), 4),
}


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(SHAPES)
    iters = int(os.environ.get("MICRO_ITERS", "5"))
    order = os.environ.get("MICRO_ORDER", "fwd,dgrad,wgrad").split(",")     # e.g. dgrad,fwd,dgrad: clocks depend on what ran just before
    dt = torch.bfloat16
    for name in names:
        cin, cout, k, s, p, tr, sp, B = SHAPES[name]
        m = ConvInstanceRelu(3, cin, cout, k, stride=s, padding=p, transposed=tr, add_norm=False, add_act=False).cuda()
        x = torch.randn(B, *sp, cpad(cin), device="cuda").to(dt)
        d = _desc(x, cin, cout, m.k, m.s, m.p, tr)
        w0 = _packed(m, 0, m.conv.weight, d, dt); w1 = _packed(m, 1, m.conv.weight, d, dt)
        y = torch.empty(B, d.out_d, d.out_h, d.out_w, d.cout_p, dtype=dt, device="cuda")
        dy = torch.randn_like(y); dx = torch.empty_like(x)
        dw = torch.zeros_like(m.conv.weight)
        st = L.stream()
        f = lambda: L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, None, L.ptr(y), None, st)
        g = lambda: L.call("nndet_conv3d_backward_data", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx), st)
        # split-K variants (what arch/conv.py calls for the small / deep layers): workspace for the partial sums
        sk0, sk1 = (int(L.load().nndet_conv3d_splitk_workspace_bytes(ctypes.byref(d), kk)) for kk in (0, 1))
        if sk0:
            wsk0 = torch.empty((sk0,), dtype=torch.uint8, device="cuda")
            f = lambda: L.call("nndet_conv3d_forward_ws", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, None, L.ptr(y), None, L.ptr(wsk0), sk0, st)
        if sk1:
            wsk1 = torch.empty((sk1,), dtype=torch.uint8, device="cuda")
            g = lambda: L.call("nndet_conv3d_backward_data_ws", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx), L.ptr(wsk1), sk1, st)
        wsb = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
        ws = L.workspace(wsb, x.device)
        h = lambda: L.call("nndet_conv3d_backward_weight", ctypes.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), None, L.ptr(ws), wsb, st)
        nvox_out = B * d.out_d * d.out_h * d.out_w
        taps = m.k[0] * m.k[1] * m.k[2]
        flops = 2.0 * (B * sp[0] * sp[1] * sp[2] if tr else nvox_out) * taps * cin * cout
        byts = 2.0 * (x.numel() + y.numel())
        res = []
        r = torch.randn_like(y)
        fr = lambda: L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, L.ptr(r), L.ptr(y), None, st)     # + fused residual (decoder top-down add)
        ga = lambda: L.call("nndet_conv3d_backward_data_acc", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx), None, st)            # dx += (fused gradient accumulation)
        # forward that reads a pre-norm tensor + coefficient table and writes the normalised input on the way (round 6, k_ig3s<.., PRE>),
        # next to the two launches it replaces (fwd_two); both with the statistics epilogue as in the step
        ss = torch.stack((torch.rand(B, d.cin_p, device="cuda") + 0.5, torch.randn(B, d.cin_p, device="cuda") * 0.3), -1).contiguous()
        xn = torch.empty_like(x)
        stats = torch.zeros(L.STATS_REPLICAS, B, d.cout_p, 2, dtype=torch.float64, device="cuda")
        dpre = _desc(x, cin, cout, m.k, m.s, m.p, tr)
        dpre.in_affine, dpre.in_relu = ss.data_ptr(), 1
        fp = lambda: L.call("nndet_conv3d_forward_norm_input", ctypes.byref(dpre), L.ptr(x), L.ptr(xn), L.ptr(w0), None, L.ptr(y), L.ptr(stats), st)
        def f2():
            L.call("nndet_affine_apply", L.dtype_code(x), L.ptr(x), L.ptr(ss), B, sp[0] * sp[1] * sp[2], d.cin_p, 1, L.ptr(xn), st)
            L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(xn), L.ptr(w0), None, None, L.ptr(y), L.ptr(stats), st)
        fs = lambda: L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, None, L.ptr(y), L.ptr(stats), st)     # + statistics epilogue
        fns = {"fwd": f, "dgrad": g, "wgrad": h, "fwd_res": fr, "dgrad_acc": ga, "fwd_pre": fp, "fwd_two": f2, "fwd_stats": fs}
        for nm in order:
            ms = timeit(fns[nm], iters)
            res.append(f"{nm} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF/s {byts / ms / 1e6:7.0f} GB/s")
        print(f"{name:16s} {flops / 1e9:7.1f} GF {byts / 1e6:7.0f} MB | " + " | ".join(res), flush=True)


def items_bench():
    """The head-trunk convolution over all pyramid levels as one ragged batch (NndetItems): 128 -> 128, luna160 pyramid, batch 4."""
    from nndetection_amd.arch import pyramid as P
    from nndetection_amd.arch.conv import ConvGroupRelu
    dt = torch.bfloat16
    lev = [(40, 40, 24), (20, 20, 12), (10, 10, 6), (5, 5, 6)]
    iters = int(os.environ.get("MICRO_ITERS", "20"))
    for cin, cout in ((128, 128), (128, 162), (128, 27)):
        m = ConvGroupRelu(3, cin, cout, 3, stride=1, padding=1, add_norm=False, add_act=False).cuda()
        meta = P.pyramid_meta([(4, *s) for s in lev])
        x = torch.randn(meta.rows, cin, device="cuda").to(dt)
        d = P._items_desc(x, m, meta)
        w0 = _packed(m, 0, m.conv.weight, d, dt); w1 = _packed(m, 1, m.conv.weight, d, dt)
        y = torch.empty(meta.rows, d.cout_p, dtype=dt, device="cuda")
        dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(m.conv.weight)
        st = L.stream()
        it = ctypes.byref(meta.items)
        f = lambda: L.call("nndet_conv3d_forward_items", ctypes.byref(d), it, L.ptr(x), L.ptr(w0), None, L.ptr(y), None, st)
        g = lambda: L.call("nndet_conv3d_backward_data_items", ctypes.byref(d), it, L.ptr(dy), L.ptr(w1), L.ptr(dx), st)
        wsb = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
        ws = L.workspace(wsb, x.device)
        h = lambda: L.call("nndet_conv3d_backward_weight_items", ctypes.byref(d), it, L.ptr(x), L.ptr(dy), L.ptr(dw), None, L.ptr(ws), wsb, st)
        flops = 2.0 * meta.rows * 27 * cin * cout
        res = [f"{nm} {timeit(fn, iters):8.3f} ms" for nm, fn in (("fwd", f), ("dgrad", g), ("wgrad", h))]
        ms = [timeit(fn, iters) for fn in (f, g, h)]
        print(f"head_items {cin}->{cout} {flops / 1e9:7.1f} GF | " + " | ".join(f"{nm} {t:7.3f} ms {flops / t / 1e9:7.1f} TF/s" for nm, t in zip(("fwd", "dgrad", "wgrad"), ms)), flush=True)


if __name__ == "__main__":
    items_bench() if "--items" in sys.argv else main()
