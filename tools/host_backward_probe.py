#!/usr/bin/env python
"""Host time spent inside the Python backward functions of a training step (the autograd engine calls them from its own thread, where
cProfile does not look): wall time per class, accumulated over N steps.  tools/host_backward_probe.py [steps=30]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd.plans import get_plan
steps = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 30
dev = torch.device("cuda:0")
r = bench.Route(get_plan("luna160"), 4, "bf16", dev, 0, False)
for _ in range(10):
    r.step()
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
import cProfile, pstats, io
PROF = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--profile=")]     # --profile=_ConvFn: cProfile inside that class's backward
prof = cProfile.Profile() if PROF else None
import gc
seen = set()
def wrap(cls):
    if cls in seen or "backward" not in cls.__dict__:
        return
    seen.add(cls)
    orig = cls.__dict__["backward"].__func__ if isinstance(cls.__dict__["backward"], staticmethod) else cls.__dict__["backward"]
    def timed(*a, **k):
        p = prof is not None and cls.__name__ in PROF
        t0 = time.perf_counter()
        if p:
            prof.enable()
        try:
            return orig(*a, **k)
        finally:
            if p:
                prof.disable()
            e = acc[cls.__name__]; e[0] += time.perf_counter() - t0; e[1] += 1
    cls.backward = staticmethod(timed)
for cls in list(torch.autograd.Function.__subclasses__()):
    if cls.__module__.startswith("nndetection_amd"):
        wrap(cls)
t_bwd = [0.0]
_b = torch.Tensor.backward
def tb(self, *a, **k):
    t0 = time.perf_counter(); out = _b(self, *a, **k); t_bwd[0] += time.perf_counter() - t0; return out
torch.Tensor.backward = tb
c0 = time.perf_counter()
for _ in range(steps):
    r.step()
host = (time.perf_counter() - c0) / steps * 1e3
torch.cuda.synchronize()
print(f"host enqueue {host:.3f} ms / step, of which loss.backward() {t_bwd[0] / steps * 1e3:.3f} ms")
tot = 0.0
for n, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:32s} {t / steps * 1e3:7.3f} ms / step  {c / steps:6.1f} calls  {t / max(c, 1) * 1e6:7.1f} us / call")
    tot += t
print(f"  sum {tot / steps * 1e3:.3f} ms / step")
if prof is not None:
    st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats("tottime").print_stats(30)
    print("\n".join(l[:160] for l in st.getvalue().splitlines()[4:44]))
