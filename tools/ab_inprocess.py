#!/usr/bin/env python
"""A/B of a host-side switch INSIDE one process: blocks of training steps alternate between the two settings (same allocations, same
clock / thermal state, drift cancels), HIP-event timed per block. Process-level A/B runs of bench.py differ by up to +-1 % between
identical runs on this pool; this tool resolves ~0.1 %.
  tools/ab_inprocess.py module.attr [blocks=12] [steps_per_block=20]        e.g. nndetection_amd.arch.conv.NORM_INPUT_FUSE
  tools/ab_inprocess.py env:NAME ... | env:NAME=A,B ...                     (an environment variable the library reads per call: 1 / 0, or A / B)
  --new-wgrad-stream: re-create the weight-gradient stream at every switch (NNDET_WGRAD_PRIO=0,1: normal +0.74 ms, =-1,1: high +2.25 ms vs the
                      default low priority); --main-high-when-on: run the "on" blocks on a high-priority stream (+0.05 ms)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from nndetection_amd.plans import get_plan

what = sys.argv[1]
_pos = [a for a in sys.argv[2:] if not a.startswith("--")]
blocks = int(_pos[0]) if len(_pos) > 0 else 12
spb = int(_pos[1]) if len(_pos) > 1 else 20
if what.startswith("env:"):                       # env:NAME (1 / 0) or env:NAME=A,B (A = "on", B = "off")
    name, _, vals = what[4:].partition("=")
    va, vb = vals.split(",") if vals else ("1", "0")
    def setv(on): os.environ[name] = va if on else vb
else:
    parts = what.split(".")
    for k in range(len(parts) - 1, 0, -1):          # module path, then attributes (a class attribute: pkg.mod.Class.attr)
        try:
            M = importlib.import_module(".".join(parts[:k]))
            break
        except ImportError:
            continue
    for a in parts[k:-1]:
        M = getattr(M, a)
    attr = parts[-1]
    assert hasattr(M, attr), what
    def setv(on): setattr(M, attr, bool(on))
from nndetection_amd import _lib as L
_setv = setv
def setv(on):                                     # (switches read when the weight-gradient stream is created: NNDET_WGRAD_PRIO)
    _setv(on)
    if "--new-wgrad-stream" in sys.argv:
        torch.cuda.synchronize()
        L.wgrad_streams.streams.clear()
    if "--new-side-streams" in sys.argv:          # (NNDET_PRIO_TAIL / _HEAD / _AUX / _AUX0 / _AUX1: read when the pools are filled)
        torch.cuda.synchronize()
        from nndetection_amd.arch.decoder import UFPNModular
        from nndetection_amd.arch.heads import DetectionHeadHNMNative
        from nndetection_amd.core.retina import BaseRetinaNet
        UFPNModular._tail_streams.clear(); DetectionHeadHNMNative._streams.clear(); BaseRetinaNet._aux_streams.clear()
dev = torch.device("cuda:0")
_hi = torch.cuda.Stream(device=dev, priority=-1) if "--main-high-when-on" in sys.argv else None      # experiment: the step on a high-priority stream
r = bench.Route(get_plan("luna160"), 4, "bf16", dev, 0, False)
for on in (True, False):
    setv(on)
    for _ in range(8):
        r.step()
torch.cuda.synchronize()
t = {True: [], False: []}
cpu = {True: [], False: []}
for b in range(2 * blocks):
    on = (b % 2 == 0) ^ ((b // 2) % 2 == 1)          # ABBA order
    setv(on)
    import contextlib
    cm = torch.cuda.stream(_hi) if (_hi is not None and on) else contextlib.nullcontext()
    with cm:
        for _ in range(3):
            r.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        c0 = time.perf_counter()
        for _ in range(spb):
            r.step()
        cpu[on].append((time.perf_counter() - c0) / spb * 1e3)        # host time to ENQUEUE a step (no synchronisation inside)
        e1.record(); torch.cuda.synchronize()
    t[on].append(e0.elapsed_time(e1) / spb)
for on in (True, False):
    a = np.array(t[on])
    print(f"{what}={int(on)}: mean {a.mean():.4f} ms  median {np.median(a):.4f}  min {a.min():.4f}  max {a.max():.4f}  ({len(a)} blocks of {spb} steps); host enqueue {np.mean(cpu[on]):.3f} ms / step")
d = np.array(t[True]) - np.array(t[False])
print(f"on - off: mean {d.mean():+.4f} ms  median {np.median(d):+.4f}  (paired by position; standard error {d.std(ddof=1) / np.sqrt(len(d)):.4f})")
