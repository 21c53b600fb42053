"""Diagnostic: host time of the phases of one training step with the RCCL gradient path forced at world size 1."""
import os, sys, time
import torch
import torch.distributed as dist
from nndetection_amd.plans import get_plan
from nndetection_amd.ptmodule import build_model, configure_optimizer
from nndetection_amd.ddp import GradAllReducer
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if os.environ.get("DIAG_DEVICE_ID") == "1" else {}))
dev = torch.device("cuda:0")
plan = get_plan("luna160")
net = build_model(plan).to(dev)
opt, sched = configure_optimizer(net)
mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
ddp = None if mode == "none" else GradAllReducer(net, force_overlap=True, overlap=(mode == "overlap"))
x, tg = synth_batch(plan, 4, torch.bfloat16, dev, seed=1000)
acc = {}
def lap(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for it in range(40):
    if it == 10:
        torch.cuda.synchronize(); acc = {}; w0 = time.perf_counter()
    t = time.perf_counter()
    losses, _ = net.train_step(x, tg, evaluation=False); loss = sum(losses.values()); t = lap("forward+loss", t)
    loss.backward(); t = lap("backward", t)
    if ddp is not None:
        ddp.finish(); t = lap("finish", t)
    opt.step(); sched.step(); opt.zero_grad(set_to_none=True); t = lap("optimizer", t)
torch.cuda.synchronize()
wall = (time.perf_counter() - w0) / 30
print(mode, "wall %.2f ms/step; host:" % (wall * 1e3), {k: round(v / 30 * 1e3, 2) for k, v in acc.items()})
