"""Full-resolution block (conv 32->32 + IN + ReLU) feeding a 1x1x1 lateral and the 32->64 stride-2 convolution: one backward pass per
iteration, for rocprofv3 --kernel-trace (k_dgs / k_norm_bwd_reduce / k_norm_bwd_apply durations with NNDET_NORM_RED_FUSE=0/1 and the
k_dgs variants of NNDET_DGS_NB_LATE). usage: python tools/dgs_normred_microbench.py [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nndetection_amd.arch import conv as CV
from nndetection_amd.arch.conv import ConvInstanceRelu

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dt = torch.bfloat16
torch.manual_seed(0)
b0 = ConvInstanceRelu(3, 32, 32, 3, padding=1, add_norm=True, add_act=True).cuda()
c1 = ConvInstanceRelu(3, 32, 64, 3, stride=2, padding=1, add_norm=True, add_act=True).cuda()
lat = ConvInstanceRelu(3, 32, 32, 1, add_norm=False, add_act=False).cuda()
x0 = torch.randn(2, 32, 160, 160, 96, device="cuda").to(dt)
g1 = torch.randn(2, 64, 80, 80, 48, device="cuda").to(dt)
g2 = torch.randn(2, 32, 160, 160, 96, device="cuda").to(dt)
for it in range(iters):
    for m in (b0, c1, lat):
        m.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    a = b0(x)
    a._nndet_gacc = {"buf": None, "stream": torch.cuda.current_stream()}
    y1, y2 = c1(a), lat(a)
    torch.autograd.backward([y2, y1], [g2, g1])
    torch.cuda.synchronize()
print("fused norm backward passes:", CV.norm_red_fused[0], "dgamma[:4]", b0.norm.weight.grad[:4].tolist())
