#!/usr/bin/env python
"""The full-resolution strided data gradient (32 <- 64 channels, stride 2) through the C ABI in its three forms -- plain
(nndet_conv3d_backward_data), accumulating (.._acc) and accumulating + norm-backward sums (.._acc_normred) -- with the persistent kernel
k_dgsp (NNDET_DGSP=1, default) and with k_dgs (NNDET_DGSP=0): dx compared bitwise, the sums to fp32 round-off, HIP-event timed.
Usage: tools/dgs_microbench.py [batch=4] [D H W = 160 160 96] [iters=20]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nndetection_amd import _lib as L
from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
from nndetection_amd.layout import cpad


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sp = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (160, 160, 96)
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    dt = torch.float16 if os.environ.get("MICRO_DT") == "f16" else torch.bfloat16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = ConvInstanceRelu(3, 32, 64, 3, stride=2, padding=1, add_norm=False, add_act=False).to(dev)
    x = torch.randn(B, *sp, 32, device=dev).to(dt)                       # (shape only: the conv input)
    d = _desc(x, 32, 64, m.k, m.s, m.p, False)
    w1 = _packed(m, 1, m.conv.weight, d, dt)
    dy = torch.randn(B, d.out_d, d.out_h, d.out_w, 64, device=dev).to(dt)
    res0 = torch.randn(B, *sp, 32, device=dev).to(dt)
    ny = (torch.randn(B, *sp, 32, device=dev) * 1.5 + 0.3).to(dt)
    mr = torch.stack((ny.float().mean((1, 2, 3)), 1.0 / (ny.float().var((1, 2, 3), unbiased=False) + 1e-5).sqrt()), -1).contiguous()
    gam, bet = (torch.rand(32, device=dev) + 0.5), torch.randn(32, device=dev) * 0.3
    st = L.stream()
    out = {}
    for mode in ("0", "1"):
        os.environ["NNDET_DGSP"] = mode
        dx_plain = torch.empty_like(x)
        L.call("nndet_conv3d_backward_data", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_plain), st)
        dx_acc = res0.clone()
        L.call("nndet_conv3d_backward_data_acc", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_acc), None, st)
        dx_nb = res0.clone()
        red = torch.zeros(L.STATS_REPLICAS * B * 32 * 2 + B, dtype=torch.float64, device=dev)
        L.call("nndet_conv3d_backward_data_acc_normred", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_nb), L.ptr(ny), L.ptr(mr), L.ptr(gam), L.ptr(bet), 1, 32,
               L.ptr(red), st)
        torch.cuda.synchronize()
        sums = red[:L.STATS_REPLICAS * B * 32 * 2].view(L.STATS_REPLICAS, B, 32, 2).sum(0)
        t = {}
        for name, fn in (("plain", lambda: L.call("nndet_conv3d_backward_data", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_plain), st)),
                         ("acc", lambda: L.call("nndet_conv3d_backward_data_acc", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_acc), None, st)),
                         ("acc+normred", lambda: L.call("nndet_conv3d_backward_data_acc_normred", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_nb), L.ptr(ny),
                                                        L.ptr(mr), L.ptr(gam), L.ptr(bet), 1, 32, L.ptr(red), st))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            t[name] = e0.elapsed_time(e1) / iters
        # re-run once for the comparison (the timed loops accumulated into dx_acc / dx_nb)
        dx_acc = res0.clone(); dx_nb = res0.clone(); red.zero_()
        L.call("nndet_conv3d_backward_data_acc", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_acc), None, st)
        L.call("nndet_conv3d_backward_data_acc_normred", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_nb), L.ptr(ny), L.ptr(mr), L.ptr(gam), L.ptr(bet), 1, 32,
               L.ptr(red), st)
        torch.cuda.synchronize()
        sums = red[:L.STATS_REPLICAS * B * 32 * 2].view(L.STATS_REPLICAS, B, 32, 2).sum(0).clone()
        out[mode] = (dx_plain, dx_acc, dx_nb, sums, t)
        gb = {"plain": 2.0 * (dy.numel() + x.numel()), "acc": 2.0 * (dy.numel() + 2 * x.numel()), "acc+normred": 2.0 * (dy.numel() + 3 * x.numel())}
        print("NNDET_DGSP=%s  " % mode + "  ".join("%s %.3f ms (%.2f TB/s)" % (k, v, gb[k] / v / 1e9) for k, v in t.items()), flush=True)
    a, b = out["0"], out["1"]
    print("nan counts (dgsp: plain, acc, nb, sums):", [int(torch.isnan(b[i].float()).sum()) for i in range(4)], "(dgs):", [int(torch.isnan(a[i].float()).sum()) for i in range(4)])
    print("dx identical (plain / acc / acc+normred):", [bool(torch.equal(a[i], b[i])) for i in range(3)],
          " max |diff|:", [float((a[i].float() - b[i].float()).abs().max()) for i in range(3)])
    rel = ((a[3] - b[3]).abs() / a[3].abs().clamp_min(1e-30)).max().item()
    print("norm-backward sums: max rel diff %.2e  (S1[0,:3] %s vs %s)" % (rel, a[3][0, :3, 0].tolist(), b[3][0, :3, 0].tolist()))
    # float64 reference of the sums from the stored dx
    xh = (ny.double() - mr[:, :, 0].double().view(B, 1, 1, 1, 32)) * mr[:, :, 1].double().view(B, 1, 1, 1, 32)
    sc = (mr[:, :, 1] * gam).view(B, 1, 1, 1, 32); sh = (bet - mr[:, :, 0] * (mr[:, :, 1] * gam)).view(B, 1, 1, 1, 32)
    mask = torch.addcmul(sh, ny.float(), sc) > 0
    gm = b[2].double() * mask
    s1, s2 = gm.sum((1, 2, 3)), (gm * xh).sum((1, 2, 3))
    print("k_dgsp sums vs float64 from its own dx: S1 rel %.2e  S2 rel %.2e" % (float((b[3][..., 0] - s1).abs().max() / s1.abs().max()),
                                                                             float((b[3][..., 1] - s2).abs().max() / s2.abs().max())))


if __name__ == "__main__":
    main()
