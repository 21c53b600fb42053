"""The fused stem block conv(1 -> 32) + IN + ReLU at full resolution (batch 4, 160 x 160 x 96): forward + backward per iteration, for
rocprofv3 --kernel-trace (k_stem_bwd3 alone). usage: python tools/stem_bwd_microbench.py [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nndetection_amd.arch.conv import ConvInstanceRelu

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(0)
m = ConvInstanceRelu(3, 1, 32, 3, padding=1, add_norm=True, add_act=True).cuda()
x = torch.randn(4, 1, 160, 160, 96, device="cuda").to(torch.bfloat16)
g = torch.randn(4, 32, 160, 160, 96, device="cuda").to(torch.bfloat16)
for it in range(iters):
    m.zero_grad(set_to_none=True)
    y = m(x)
    y.backward(g)
    torch.cuda.synchronize()
print("dw[0,0,0,0,:3]", m.conv.weight.grad[0, 0, 0, 0, :3].tolist(), "dgamma[:3]", m.norm.weight.grad[:3].tolist())
