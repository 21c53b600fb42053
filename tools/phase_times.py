#!/usr/bin/env python
"""Un-profiled GPU time of the phases of a training step (CUDA events on the main stream): encoder + decoder forward, head forward,
targets + losses, backward until every head-input gradient exists (= loss + head backward), the rest of backward, optimizer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model, configure_optimizer

dev = torch.device("cuda:0")
plan = get_plan(sys.argv[1] if len(sys.argv) > 1 else "luna160")
net = build_model(plan).to(dev)
opt, sched = configure_optimizer(net)
x, tg = bench.synth_batch(plan, plan["batch_size"], torch.bfloat16, dev, seed=1000)
ev = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev[name] = e


head_fwd = net.head.forward
pending = {"n": 0}


def timed_head(fmaps):
    mark("head_fwd_start")
    pending["n"] = len(fmaps)
    for p in fmaps:
        def hook(g):
            pending["n"] -= 1
            if pending["n"] == 0:
                mark("head_bwd_done")
            return g
        p.register_hook(hook)
    out = head_fwd(fmaps)
    mark("head_fwd_end")
    return out


net.head.forward = timed_head


def step():
    mark("start")
    losses, _ = net.train_step(x, tg, evaluation=False, batch_num=0)
    loss = sum(losses.values())
    mark("loss_done")
    loss.backward()
    mark("bwd_done")
    opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
    mark("end")


for _ in range(8):
    step()
acc = {}
N = 20
for _ in range(N):
    step()
    torch.cuda.synchronize()
    seq = ["start", "head_fwd_start", "head_fwd_end", "loss_done", "head_bwd_done", "bwd_done", "end"]
    for a, b in zip(seq[:-1], seq[1:]):
        acc[(a, b)] = acc.get((a, b), 0.0) + ev[a].elapsed_time(ev[b])
names = {("start", "head_fwd_start"): "encoder + decoder forward", ("head_fwd_start", "head_fwd_end"): "head forward",
         ("head_fwd_end", "loss_done"): "anchors + ATSS + sampler + losses (+ seg head)", ("loss_done", "head_bwd_done"): "loss + head backward",
         ("head_bwd_done", "bwd_done"): "decoder + encoder backward", ("bwd_done", "end"): "optimizer"}
tot = 0.0
for k, v in acc.items():
    print(f"{names[k]:50s} {v / N:7.3f} ms")
    tot += v / N
print(f"{'sum':50s} {tot:7.3f} ms   (events on the main stream; the synchronize per step lets the host fall behind, so this is GPU time, not the free-running wall)")
