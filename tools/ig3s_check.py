#!/usr/bin/env python
"""k_ig3s vs k_igemm on the same input: run twice (NNDET_IG3S=1 / 0, the switch is read once per process) with a file in between.
    NNDET_IG3S=0 tools/ig3s_check.py save /tmp/ig3s_ref.pt ; NNDET_IG3S=1 tools/ig3s_check.py cmp /tmp/ig3s_ref.pt"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndetection_amd import _lib as L
from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
mode, path = sys.argv[1], sys.argv[2]
res = {}
for name, sp, B in (("odd", (21, 19, 35), 2), ("e1", (160, 160, 96), 4), ("small", (10, 9, 12), 2)):
    torch.manual_seed(1)
    m = ConvInstanceRelu(3, 32, 64, 3, stride=2, padding=1, add_norm=False, add_act=False).cuda()
    x = torch.randn(B, *sp, 32, device="cuda").to(torch.bfloat16)
    d = _desc(x, 32, 64, m.k, m.s, m.p, False)
    w0 = _packed(m, 0, m.conv.weight, d, torch.bfloat16)
    outs = []
    for rep in range(3):
        y = torch.full((B, d.out_d, d.out_h, d.out_w, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros((32, B, 64, 2), dtype=torch.float64, device="cuda")
        L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, None, L.ptr(y), L.ptr(stats), L.stream())
        torch.cuda.synchronize()
        outs.append((y.clone(), stats.sum(0).clone()))
    det = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    print(name, "deterministic output:", det, "nan:", int(torch.isnan(outs[0][0].float()).sum()), "stats run-to-run max rel diff:",
          float(((outs[0][1] - outs[1][1]).abs() / (outs[0][1].abs() + 1e-9)).max()))
    res[name] = (outs[0][0].cpu(), outs[0][1].cpu())
if mode == "save":
    torch.save(res, path)
else:
    ref = torch.load(path)
    for k in res:
        a, b = res[k][0].float(), ref[k][0].float()
        dmax = float((a - b).abs().max()); nz = int(((a - b).abs() > 0).sum())
        big = int(((a - b).abs() > 0.02 * b.abs().max()).sum())
        sa, sb = res[k][1], ref[k][1]
        print(k, "max |diff|", dmax, "of max", float(b.abs().max()), "differing elements", nz, "of", a.numel(), "| > 2% of max:", big,
              "| stats max rel diff", float(((sa - sb).abs() / (sb.abs() + 1e-6)).max()))
