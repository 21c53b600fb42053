#!/usr/bin/env python
"""Which calls of one training step block the host on the GPU? Runs a few steps of the bench configuration under
torch.cuda.set_sync_debug_mode("warn") and prints every synchronizing call with its Python location, then times the step with
the host free-running (wall per step) against the host-only time (time until the last launch of a step is queued)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model, configure_optimizer

dev = torch.device("cuda:0")
plan = get_plan(sys.argv[1] if len(sys.argv) > 1 else "luna160")
batch = plan["batch_size"]
torch.manual_seed(0)
net = build_model(plan).to(dev)
opt, sched = configure_optimizer(net)
x, tg = bench.synth_batch(plan, batch, torch.bfloat16, dev, seed=1000)


def step():
    losses, _ = net.train_step(x, tg, evaluation=False, batch_num=0)
    loss = sum(losses.values())
    loss.backward()
    opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"synchronizing calls in one step: {len(w)}")
for m in w:
    print(f"  {m.filename.split('repo/')[-1]}:{m.lineno}: {str(m.message)[:100]}")
n = 20
host = 0.0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    h0 = time.perf_counter()
    step()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall per step {wall / n * 1e3:.2f} ms, host time inside step() {host / n * 1e3:.2f} ms (includes the time the host is blocked in synchronizing calls)")
if "--profile" in sys.argv:
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        step()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
