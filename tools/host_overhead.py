#!/usr/bin/env python
"""Measure the pure HOST cost of one training step (Python + torch dispatch + ctypes), with every kernel launch
stubbed out, on CPU tensors. Usage: tools/host_overhead.py [plan] [--profile]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nndetection_amd import _lib as L

L.ptr = lambda t: 0 if t is not None else None
L.stream = lambda: 0
L.call = lambda name, *a: None


class _FakeLib:
    def __getattr__(self, n):
        return lambda *a: 1 << 20


L.load = lambda: _FakeLib()
from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model, configure_optimizer
import nndetection_amd.core.boxes.matcher as M
import nndetection_amd.core.boxes.anchors as AN

plan = get_plan(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "tiny")
net = build_model(plan)
opt, sched = configure_optimizer(net)
B = 2
x = torch.randn(B, 1, *plan["patch_size"]).to(torch.bfloat16)
tg = {"target_boxes": [torch.tensor([[4., 4, 12, 12, 4, 12]])] * B, "target_classes": [torch.zeros(1)] * B,
      "target_seg": torch.zeros(B, *plan["patch_size"])}


def step():
    losses, _ = net.train_step(x, tg, evaluation=False)
    loss = sum(losses.values())
    loss.backward()
    opt.step(); sched.step(); opt.zero_grad(set_to_none=True)


# make the (garbage) matches valid indices
orig = M.ATSSMatcher.__call__
def fake_match(self, boxes, anchors, num_anchors_per_level=None, num_anchors_per_loc=None):
    m = torch.full((anchors.shape[0],), -1, dtype=torch.int64); m[:50] = 0
    return anchors.new_ones(1), m
M.ATSSMatcher.__call__ = fake_match
def fake_match_batch(self, boxes, anchors, npl, napl):
    gt = torch.cat([b.reshape(-1, 6).float() for b in boxes], 0)
    offs = [0]
    for b in boxes:
        offs.append(offs[-1] + b.reshape(-1, 6).shape[0])
    m = torch.full((len(boxes), anchors.shape[0]), -1, dtype=torch.int64); m[:, :50] = 0
    return gt, m, offs
M.ATSSMatcher.match_batch = fake_match_batch
import nndetection_amd.core.boxes.sampler as S
S.HardNegativeSamplerBatched.sample_indices = lambda self, labels, scores, bs, **k: (torch.arange(8), torch.arange(8, 40))
for _ in range(2):
    step()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
print(f"host time per step (kernels stubbed, CPU tensors, plan {plan['patch_size']}): {(time.perf_counter() - t0) / n * 1e3:.1f} ms")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
if "--conv" in sys.argv:
    from nndetection_amd.arch.conv import ConvInstanceRelu
    m = ConvInstanceRelu(3, 64, 64, 3, padding=1)
    xx = torch.randn(2, 64, 8, 8, 8).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    for _ in range(20):
        y = m(xx); y.sum().backward()
    import time as _t
    t0 = _t.perf_counter()
    for _ in range(200):
        y = m(xx)
    t1 = _t.perf_counter()
    ys = [m(xx) for _ in range(200)]
    g = torch.ones_like(ys[0])
    t2 = _t.perf_counter()
    for y in ys:
        y.backward(g)
    t3 = _t.perf_counter()
    print(f"conv block forward {(t1 - t0) / 200 * 1e6:.0f} us, backward {(t3 - t2) / 200 * 1e6:.0f} us per call (host only)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50):
        y = m(xx); y.backward(g)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
