#!/usr/bin/env python
"""Run-to-run spread of the toy64 bf16 trajectory (the test's measure: mean / max gap of the 10-step mean to ONE fp32 run, in units of the
initial loss) for two settings of an environment switch read per call:  tools/trajectory_spread.py NNDET_IG3S 0 1 [runs=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.trajectory import run
from nndetection_amd.plans import get_plan
from tests.gpu_util import det_randperm
torch.randperm = det_randperm
var, vals, runs = sys.argv[1], sys.argv[2:4], int(sys.argv[4]) if len(sys.argv) > 4 else 6
sm = lambda c: np.convolve(c, np.ones(10) / 10, mode="valid")
for v in vals:
    os.environ[var] = v
    plan = get_plan("toy64")
    ref, _ = run(plan, "f32", 200, 1e-3)
    l0 = float(ref[0]); half = lambda c: int(np.argmax(sm(c) < 0.5 * l0))
    out = []
    for dn in ["f32b"] + ["bf16"] * runs:
        c, _ = run(plan, dn, 200, 1e-3)
        gap = np.abs(sm(c) - sm(ref)) / l0
        out.append("%s mean %.4f max %.4f half@%d" % (dn, gap.mean(), gap.max(), half(c)))
    print("%s=%s (fp32 half@%d): " % (var, v, half(ref)) + " | ".join(out), flush=True)
