#!/usr/bin/env python
"""Rounding quality of k_ig3s / k_ig3s2 vs k_igemm: outputs (bf16) against the correctly rounded float64 convolution of the same bf16
operands; also the InstanceNorm statistics (sum, sum of squares) against float64 sums of the kernel's own rounded outputs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nndetection_amd import _lib as L
from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
for cin, cout in ((32, 64), (64, 128)):
    torch.manual_seed(1)
    sp, B = (33, 31, 35), 2
    m = ConvInstanceRelu(3, cin, cout, 3, stride=2, padding=1, add_norm=False, add_act=False).cuda()
    x = (torch.randn(B, *sp, cin, device="cuda") * 1.5).to(torch.bfloat16)
    d = _desc(x, cin, cout, m.k, m.s, m.p, False)
    w0 = _packed(m, 0, m.conv.weight, d, torch.bfloat16)
    wq = m.conv.weight.detach().to(torch.bfloat16).double().cpu()
    ref = F.conv3d(x.double().cpu().permute(0, 4, 1, 2, 3), wq, None, stride=2, padding=1).permute(0, 2, 3, 4, 1)
    ref_r = ref.to(torch.bfloat16).double()
    for ig in ("0", "1"):
        os.environ["NNDET_IG3S"] = ig
        y = torch.empty((B, d.out_d, d.out_h, d.out_w, cout), dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros((32, B, cout, 2), dtype=torch.float64, device="cuda")
        L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, None, L.ptr(y), L.ptr(stats), L.stream())
        torch.cuda.synchronize()
        yd = y.double().cpu()
        mis = int((yd != ref_r).sum())
        err = (yd - ref)
        st = stats.sum(0).cpu()
        s_ref = torch.stack((yd.sum((1, 2, 3)), (yd * yd).sum((1, 2, 3))), -1)
        print(f"{cin}->{cout} IG3S={ig}: misrounded {mis} of {yd.numel()} ({100.0 * mis / yd.numel():.3f} %), mean signed error {float(err.mean()):+.3e}, "
              f"rms error {float(err.pow(2).mean().sqrt()):.3e}, stats max rel err {float(((st - s_ref).abs() / s_ref.abs().clamp_min(1e-9)).max()):.2e}")
