#!/usr/bin/env python
"""Benchmark of the MI355X-native RetinaUNet hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one full training step of RetinaUNetV001 on one batch of synthetic 160x160x96 patches
(BASELINE.json configs[1]: Task016_Luna-like plan, batch 4 per GPU, bf16 activations): forward, ATSS target
assignment, hard-negative sampling, losses, backward, gradient all-reduce (N > 1), SGD(nesterov) step, LR step.
Inputs are resident in HBM before the timed region. Prints ONE JSON line on rank 0:
  metric/value = patches/s (whole job), roofline = dominant conv kernel vs the HBM roofline (HIP-event timed here),
  cpu_baseline = the CPU oracle ("port" of the reference, plain PyTorch fp32) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def synth_batch(plan, batch, dtype, device, seed):
    P = plan["patch_size"]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 1, *P, generator=g).to(device=device, dtype=dtype)
    rng = np.random.default_rng(seed + 1)
    boxes, classes = [], []
    seg = torch.zeros(batch, *P)
    for b in range(batch):
        c = rng.uniform(0, 1, (3, 3)) * np.asarray(P)
        s = rng.uniform(4, 24, (3, 3))
        lo, hi = np.clip(c - s / 2, 0, None), np.minimum(c + s / 2, np.asarray(P))
        bb = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1).astype(np.float32)
        boxes.append(torch.from_numpy(bb).to(device)); classes.append(torch.zeros(3, device=device))
        for q in bb:
            seg[b, int(q[0]):int(q[2]) + 1, int(q[1]):int(q[3]) + 1, int(q[4]):int(q[5]) + 1] = 1
    return x, {"target_boxes": boxes, "target_classes": classes, "target_seg": seg.to(device)}


def conv_roofline(plan, batch, dtype, device, iters=50):
    """Dominant kernel: the 3x3x3 implicit-GEMM conv at full resolution (encoder.stages.0.convs.0.1 and
    decoder.out.P0 have this shape: 32 -> 32 channels, 135.9 GFLOP per patch each).
    Algorithmic traffic per launch = read the input once + write the output once."""
    from nndetection_amd.arch.conv import ConvInstanceRelu
    P = plan["patch_size"]
    c = plan["arch"]["start_channels"]
    m = ConvInstanceRelu(3, c, c, 3, stride=1, padding=1, add_norm=False, add_act=False).to(device)
    x = torch.randn(batch, c, *P, device=device).to(dtype).contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        for _ in range(20):              # the first ~20 launches after other work run at a lower clock (measured 0.56 vs 0.44 ms)
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            m(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters     # average launch duration, back to back on torch's current stream (= the stream the kernel runs on)
    nvox = batch * P[0] * P[1] * P[2]
    esz = torch.tensor([], dtype=dtype).element_size()
    alg_bytes = 2 * nvox * c * esz + 27 * c * c * esz
    flops = 2.0 * nvox * 27 * c * c
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    tfs = flops / (ms * 1e-3) / 1e12
    # HBM traffic per launch from the PMC passes of the same kernel / shape (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate
    # passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): measured by `tools/gpu_round.sh pmc`, committed
    # as profiles/round1_pmc_traffic.json together with the raw summary. Not re-measured inside this run (PMC needs rocprofv3).
    traffic = None
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("round2_pmc_traffic.json", "round1_pmc_traffic.json"):
        tj = os.path.join(pdir, name)
        if os.path.isfile(tj) and batch == 4 and tuple(P) == (160, 160, 96) and dtype == torch.bfloat16:
            with open(tj) as f:
                tr = json.load(f)
                key = "k_ig3r_e0" if os.environ.get("NNDET_IG3R", "1") != "0" and "k_ig3r_e0" in tr else "k_ig3_cfgA_e0"
                traffic = int(tr[key]["hbm_bytes"])
            break
    # 432 FLOP per algorithmic byte is above the MFMA/HBM ridge (2500 TF/s / 8 TB/s = 312): the kernel is priced against the
    # dense bf16 MFMA peak; the HBM view (algorithmic bytes / time) is reported next to it
    kname = "k_ig3r<bf16> (persistent, weights in registers)" if os.environ.get("NNDET_IG3R", "1") != "0" else "k_ig3<bf16,WR=1,MT=2,NT=8>"
    return {"bound": "mfma", "kernel": kname + " conv3d 3x3x3 32->32 @%dx%dx%d, batch %d (forward)" % (*P, batch),
            "achieved": round(tfs, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tfs / 2500.0, 4),
            "traffic": traffic, "algorithmic_flops_per_launch": int(flops), "algorithmic_bytes_per_launch": int(alg_bytes),
            "ms_per_launch": round(float(ms), 4), "algorithmic_GBs": round(gbs, 1), "hbm_frac_of_8TBs": round(gbs / 8000.0, 4),
            "measured_mfma_ceiling_TFs": 1900.0}


def nms_rate(device, n=10000, iters=5):
    from nndetection_amd.core.boxes import nms
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 160, (n, 3)); s = rng.uniform(2, 26, (n, 3))
    b = np.stack([c[:, 0] - s[:, 0] / 2, c[:, 1] - s[:, 1] / 2, c[:, 0] + s[:, 0] / 2, c[:, 1] + s[:, 1] / 2,
                  c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1).astype(np.float32)
    sc = ((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)
    bt, st = torch.from_numpy(b).to(device), torch.from_numpy(sc).to(device)
    nms(bt, st, 0.6); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        k = nms(bt, st, 0.6)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"n": n, "thr": 0.6, "kept": int(k.numel()), "boxes_per_s": round(n / dt, 1), "ms": round(dt * 1e3, 3)}


def _det_randperm(n, *a, **k):
    return torch.arange(n - 1, -1, -1, device=k.get("device", None))


_det_randperm.nndet_reversed_arange = True     # the device sampler then selects what this permutation selects (parity_check)


def cpu_baseline(plan, device=None):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference, oracle/retina_torch.py; kind "port") on this host:
    ONE patch forward + ATSS + losses + backward (a bounded sample of the same workload). The thread count is chosen by a
    short sweep over {physical cores, 64, 32, 16} on a proxy (the full-resolution 32->32 3x3x3 convolution, forward + backward:
    the layer type that dominates the CPU time), after one warm-up call; 256 SMT threads oversubscribe oneDNN badly (round 1
    measured 0.008 patches/s that way). The same patch / weights then go through the HIP fp32 kernels and the four losses are
    compared (`parity_check`): the headline configuration checked against the oracle inside the benchmark run."""
    from oracle.retina_torch import OracleRetinaUNet
    from nndetection_amd.plans import MODEL_CFG_V001
    logical = os.cpu_count() or 1
    cands = sorted({max(1, logical // 2), 64, 32, 16} & set(range(1, logical + 1)) | {min(16, logical)}, reverse=True)
    P = plan["patch_size"]
    conv = torch.nn.Conv3d(32, 32, 3, padding=1)
    xs = torch.randn(1, 32, *P)
    sweep = {}
    for i, th in enumerate([cands[-1]] + cands):             # first entry = warm-up (not recorded)
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        conv(xs).sum().backward()
        if i:
            sweep[th] = round(time.perf_counter() - t0, 3)
    best = min(sweep, key=sweep.get)
    del conv, xs
    torch.set_num_threads(best)
    torch.manual_seed(0)
    net = OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001)
    x, tg = synth_batch(plan, 1, torch.float32, "cpu", 0)
    orig = torch.randperm
    torch.randperm = _det_randperm                          # the sampler's permutation, fixed for the parity comparison
    try:
        t0 = time.perf_counter()
        losses, _ = net.train_step(x, tg, evaluation=False)
        sum(losses.values()).backward()
        dt = time.perf_counter() - t0
        out = {"value": round(1.0 / dt, 4), "unit": "patches/s", "cores": best, "threads": best, "logical_cpus": logical,
               "kind": "port", "thread_sweep_proxy_s": sweep,
               "sample": "1 patch %dx%dx%d fp32: forward + ATSS + losses + backward of the CPU oracle (%.1f s, %d threads)" % (*P, dt, best)}
        parity = None
        if device is not None:
            from nndetection_amd.ptmodule import build_model
            hip = build_model(plan)
            hip.load_state_dict(net.state_dict())
            hip.to(device)
            tgd = {"target_boxes": [b.to(device) for b in tg["target_boxes"]], "target_classes": [c.to(device) for c in tg["target_classes"]],
                   "target_seg": tg["target_seg"].to(device)}
            lg, _ = hip.train_step(x.to(device), tgd, evaluation=False)
            lc = {k: float(v) for k, v in losses.items()}
            lgv = {k: float(v) for k, v in lg.items()}
            diff = max(abs(lc[k] - lgv[k]) for k in lc)
            parity = {"what": "losses of the HIP fp32 kernels vs the CPU oracle on the cpu_baseline patch (same weights, same sampler permutation)",
                      "losses_cpu_oracle": {k: round(v, 6) for k, v in lc.items()}, "losses_hip_fp32": {k: round(v, 6) for k, v in lgv.items()},
                      "max_abs_diff": diff, "tolerance": 1e-4, "ok": bool(diff <= 1e-4 and set(lc) == set(lgv))}
            del hip
    finally:
        torch.randperm = orig
    return out, parity


def inference_rate(net, x, iters=5):
    """`inference_step` (forward + fused post-processing: top-k on the logits, decode of the survivors, batched NMS) per image."""
    with torch.no_grad():
        net.inference_step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = net.inference_step(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"ms_per_image": round(dt * 1e3 / x.shape[0], 3), "batch": int(x.shape[0]),
            "detections": [int(b.shape[0]) for b in out["pred_boxes"]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--plan", default="luna160")
    ap.add_argument("--batch", type=int, default=None, help="patches per GPU (default: the plan's batch size, 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / NMS / CPU legs (only the timed steps)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("NNDET_BENCH_FORCE_DIST") == "1"     # exercise the RCCL / bucket path at world size 1 (testing)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # (no device_id=: eager communicator initialisation bound to the device measured +0.85 ms per training step on MI355X /
        # ROCm 7.2 / torch 2.10 even at world size 1 with no collective in flight -- tools/diag_ddp.py; the device is selected below,
        # the communicator is created by the first collective)
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from nndetection_amd.plans import get_plan
    from nndetection_amd.ptmodule import build_model, configure_optimizer
    from nndetection_amd.ddp import GradAllReducer

    plan = get_plan(args.plan)
    batch = args.batch or plan["batch_size"]
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    net = build_model(plan).to(device)
    opt, sched = configure_optimizer(net)
    ddp = GradAllReducer(net, force_overlap=force_dist, overlap=os.environ.get("NNDET_DDP_OVERLAP", "1") != "0") if (world > 1 or force_dist) else None
    x, tg = synth_batch(plan, batch, dtype, device, seed=1000 + rank)
    torch.manual_seed(1234 + rank)

    def step():
        losses, _ = net.train_step(x, tg, evaluation=False, batch_num=0)
        loss = sum(losses.values())
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        sched.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(args.warmup):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(last.detach().float().item())
    peak_gb = torch.cuda.max_memory_allocated(device) / 2 ** 30

    if rank == 0:
        out = {
            "metric": "patches/sec (fwd+bwd) %dx%dx%d RetinaUNet" % tuple(plan["patch_size"]), "value": round(batch * world * args.steps / dt, 3),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: Task016_Luna-like plan (SURVEY 8)" if args.plan == "luna160" else
                                    "plan '%s' (not the headline configuration)" % args.plan) +
                                   ", RetinaUNetV001 train step (fwd + ATSS + losses + bwd + SGD), %dx%dx%d patches" % tuple(plan["patch_size"]),
                       "plan": args.plan, "batch_per_gpu": batch, "global_batch": batch * world,
                       "parallelism": "dp%d" % world, "params": sum(p.numel() for p in net.parameters())},
            "final_loss": round(loss_val, 5), "peak_hbm_gib": round(peak_gb, 2),
        }
        if not args.no_extras:
            out["inference"] = inference_rate(net, x)
            del x, tg
            torch.cuda.empty_cache()
            out["roofline"] = conv_roofline(plan, batch, dtype, device)      # rank 0's GPU; the other ranks are done
            out["nms"] = nms_rate(device)
            if not args.no_cpu_baseline and world == 1:                      # the CPU leg only at N = 1 (rank 0)
                del net, opt, sched
                torch.cuda.empty_cache()
                out["cpu_baseline"], out["parity_check"] = cpu_baseline(plan, device)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
