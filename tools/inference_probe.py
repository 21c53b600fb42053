import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from nndetection_amd.plans import get_plan
plan = get_plan("luna160")
dev = torch.device("cuda:0")
r = bench.Route(plan, 4, "bf16", dev, 0, False)
net = r.net.eval()
x = r.x
def t(fn, n=10):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("forward only (batch 4): %.3f ms" % t(lambda: net(x)))
print("inference_step (batch 4): %.3f ms" % t(lambda: net.inference_step(x)))
for dt in (torch.float16,):
    xx = x.to(dt)
    print("inference_step f16: %.3f ms" % t(lambda: net.inference_step(xx)))
x1 = x[:1]
print("inference_step (batch 1): %.3f ms" % t(lambda: net.inference_step(x1)))
print("forward only (batch 1): %.3f ms" % t(lambda: net(x1)))
