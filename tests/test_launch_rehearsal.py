"""CPU rehearsal of the N-rank launch (VERDICT r5 item 8 "multi-GPU without hardware"): the launcher line the driver
uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
...`) with `--cpu-dry-run` (gloo, CPU tensors, a stand-in network that takes its gradient memory from the product's gradient pool), and
`tools/scale_sweep.sh` end to end at 8 processes. What it proves without a GPU: rendezvous and environment handling, per-rank seeding,
the reducer's cross-rank layout check, the in-place bucket path with a positive-free rank (every bucket still launched from the hooks),
bit-identical parameters on all ranks after the steps although the ranks started from different parameters.
It measures nothing and the product has no CPU path: the stand-in is declared as such in the JSON line (`value` null, `dry_run` true)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    e = dict(os.environ, OMP_NUM_THREADS="1", **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_bench_dry_run_at_8_ranks_through_the_drivers_launcher_line():
    port = 21000 + os.getpid() % 4000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--cpu-dry-run"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(NNDET_DDP_FIRST_MB="0.05", NNDET_DDP_BUCKET_MB="0.15"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1
    assert d["params_identical_on_all_ranks"] is True, "ranks diverged (broadcast, all-reduce or optimizer wiring)"
    assert d["positive_free_rank"] == 7 and d["all_buckets_launched_from_hooks_on_all_ranks"] is True
    assert d["ddp"]["buckets"] >= 3 and d["ddp"]["in_place"] is True and d["ddp"]["copied_last"] == 0
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "dp8"


def test_scale_sweep_script_end_to_end_at_8_gloo_processes():
    port = 25000 + os.getpid() % 4000
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_sweep.sh"), "8", "2", "1"], cwd=ROOT,
                       env=_env(DRY="1", FIRST="0.05", BUCKET="0.1 0.4", BF16="0", PORT=str(port)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.splitlines() if l and not l.startswith("#") and not l.startswith("first_MB")]
    assert len(rows) == 2, r.stdout
    for row, bucket in zip(rows, ("0.1", "0.4")):
        assert row[0] == "0.05" and row[1] == bucket and row[2] == "0" and row[3] == "dry-run:ok" and row[-1] == "identical", row
    assert int(rows[0][5]) > int(rows[1][5]) >= 2, "smaller buckets -> more buckets (the knob reaches the reducer of every rank)"


def _mismatch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from nndetection_amd.ddp import GradAllReducer
    torch.manual_seed(0)
    model = nn.Sequential(*[nn.Linear(32, 32) for _ in range(6)])
    res = {}
    for tag, kw in (("bucket_size", dict(bucket_mb=2e-3 if rank == 0 else 8e-3)),
                    ("never_used_set", dict(bucket_mb=4e-3, static_unused=list(model[5].parameters()) if rank == 1 else [])),
                    ("same", dict(bucket_mb=4e-3))):
        try:
            ddp = GradAllReducer(model, first_bucket_mb=1e-3, **kw)
            res[tag] = "ok:" + ddp.layout_digest[:8]
            ddp.close()
        except RuntimeError as e:
            res[tag] = "raised:" + str(e)[:80]
    if rank == 1:                                  # a parameter frozen on one rank only
        model[2].weight.requires_grad_(False)
    try:
        GradAllReducer(model, first_bucket_mb=1e-3, bucket_mb=4e-3).close()
        res["frozen_on_one_rank"] = "ok"
    except RuntimeError as e:
        res["frozen_on_one_rank"] = "raised:" + str(e)[:80]
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_refuses_a_bucket_layout_that_differs_between_ranks():
    """A mismatch would otherwise be a hang inside the first RCCL collective of an 8-GPU lease: different bucket sizes, a different
    never-used set, a parameter frozen on one rank -- every rank raises at construction; identical layouts agree on one digest."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27000 + os.getpid() % 2000
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for rank in range(2):
        for tag in ("bucket_size", "never_used_set", "frozen_on_one_rank"):
            assert res[rank][tag].startswith("raised:GradAllReducer: the gradient bucket layout differs"), (rank, tag, res[rank][tag])
    assert res[0]["same"].startswith("ok:") and res[0]["same"] == res[1]["same"]
