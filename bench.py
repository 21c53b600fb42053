#!/usr/bin/env python
"""Benchmark of the MI355X-native RetinaUNet hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one full training step of RetinaUNetV001 on one batch of synthetic 160x160x96 patches
(BASELINE.json configs[1]: Task016_Luna-like plan, batch 4 per GPU, bf16 activations): forward, ATSS target
assignment, hard-negative sampling, losses, backward, gradient all-reduce (N > 1), SGD(nesterov) step, LR step.
Inputs are resident in HBM before the timed region. Prints ONE JSON line on rank 0:
  metric/value = patches/s (whole job),
  roofline = the kernel family with the most TIME per step and the one that ends the step: the 3x3x3 weight gradient k_wgrad3d, its largest
             launch (32 -> 32 at full resolution) vs the MFMA roofline, HIP-event timed INSIDE the training step on the weight-gradient
             stream (and alone), HBM traffic from PMC passes run by this process when rocprofv3 is installed,
  roofline_ig3r / chip_probe = the kernel with the most FLOPs per launch (k_ig3r forward, the `roofline` of rounds 1-5), alone: a chip probe,
  timed_blocks = three back-to-back timed blocks (value = the first), box_microbench = BASELINE.json configs[4] at its sizes,
  config3_lidc192 = configs[3] (192x192x128, fp16, batch 4): ms per step + peak HBM,
  roofline_dominant = the kernel with the most TIME per step (the ragged head trunk launch), step_roofline = algorithmic FLOPs and
             bytes of the whole step / ms_per_step,
  routes = the same step through the registered plugin's `training_step` (batch dicts, device-side targets), in fp32 and in fp16
           (+ GradScaler), each timed on a short run next to the headline,
  ddp (N > 1) = exposed (non-overlapped) gradient all-reduce time per step and per-bucket launch offsets,
  cpu_baseline = the CPU oracle ("port" of the reference, plain PyTorch fp32) timed on this host's cores.

    --via-plugin   the headline itself goes through RetinaUNetAMDSteps.training_step (what nnDetection + Lightning would call)
    --dtype f16    fp16 activations + torch.amp.GradScaler (the reference's precision=16); bf16 is BASELINE.json's configs[1]
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


def synth_plugin_batch(plan, batch, device, seed):
    """The same patch / boxes as `synth_batch`, as the raw nnDetection batch dict a data loader hands `training_step`
    (nndet/ptmodule/retinaunet/base.py:135-154): fp32 image, instance-id volume, {instance id: class} per image."""
    x, tg = synth_batch(plan, batch, torch.float32, "cpu", seed)
    P = plan["patch_size"]
    inst = torch.zeros(batch, 1, *P)
    maps = []
    for b in range(batch):
        m = {}
        for i, q in enumerate(tg["target_boxes"][b].numpy(), start=1):
            inst[b, 0, int(q[0]):int(q[2]) + 1, int(q[1]):int(q[3]) + 1, int(q[4]):int(q[5]) + 1] = i
            m[i] = 0
        m[len(m) + 5] = 0                       # an instance of the case that is not inside this patch
        maps.append(m)
    return {"data": x.to(device), "target": inst.to(device), "instance_mapping": maps}


_TORCH_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
NBATCH = int(os.environ.get("NNDET_BENCH_NBATCH", "3"))     # distinct synthetic batches a route cycles through


class Route:
    """One way of running the training step: direct (BaseRetinaNet.train_step on prepared targets) or via the plugin
    (RetinaUNetAMDSteps.training_step on a batch dict, inside torch.autocast as Lightning's native AMP does), in a given activation
    dtype; fp16 uses torch.amp.GradScaler (scale -> backward -> unscale + inf check -> step, sync-free with the fused optimizer)."""

    def __init__(self, plan, batch, dtype_name, device, rank, via_plugin, ddp_factory=None):
        import contextlib
        from nndetection_amd.plans import MODEL_CFG_V001, TRAINER_CFG_V001
        from nndetection_amd.ptmodule import build_model, configure_optimizer, StandaloneRetinaUNetV001AMD
        import copy
        self.dtype_name, self.via_plugin = dtype_name, via_plugin
        dt = _TORCH_DT[dtype_name]
        torch.manual_seed(0)
        if via_plugin:
            pl_plan = {"architecture": plan["arch"], "anchors": plan["anchors"], "patch_size": plan["patch_size"], "batch_size": batch}
            cfg = dict(TRAINER_CFG_V001, precision=32 if dtype_name == "f32" else 16)
            self.mod = StandaloneRetinaUNetV001AMD(copy.deepcopy(MODEL_CFG_V001), cfg, pl_plan).to(device)
            self.net = self.mod.model
            self.opt, self.sched = self.mod.configure_optimizers()
            # NBATCH distinct synthetic batches, resident in HBM, visited round-robin (not one identical batch every step)
            self.batches = [synth_plugin_batch(plan, batch, device, seed=1000 + rank + 17 * i) for i in range(NBATCH)]
            self.batch = self.batches[0]
            self.ctx = (lambda: torch.autocast("cuda", dtype=dt)) if dtype_name != "f32" else contextlib.nullcontext
        else:
            self.net = build_model(plan).to(device)
            self.opt, self.sched = configure_optimizer(self.net)
            self.data = [synth_batch(plan, batch, dt, device, seed=1000 + rank + 17 * i) for i in range(NBATCH)]
            self.x, self.tg = self.data[0]
        self.it = 0
        self.scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14) if dtype_name == "f16" else None
        self.ddp = ddp_factory(self.net) if ddp_factory is not None else None
        torch.manual_seed(1234 + rank)

    def step(self):
        self.it += 1
        if self.via_plugin:
            with self.ctx():
                out = self.mod.training_step(self.batches[self.it % NBATCH], self.it)
            loss = out["loss"]
        else:
            x, tg = self.data[self.it % NBATCH]
            losses, _ = self.net.train_step(x, tg, evaluation=False, batch_num=self.it)
            loss = sum(losses.values())
        if self.ddp is not None:
            self.ddp.begin_step()
        (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        if self.ddp is not None:
            self.ddp.finish()
        if self.scaler is not None:
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            self.opt.step()
        self.sched.step()
        self.opt.zero_grad(set_to_none=True)
        return loss

    def timed(self, warmup, steps, world=1):
        sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)      # (--cpu-dry-run: nothing to wait for)
        for _ in range(warmup):
            last = self.step()
        sync()
        if self.ddp is not None and self.ddp.profile:
            self.ddp.profile_reset()                    # the exposed-communication figures cover the timed steps only
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = self.step()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        return time.perf_counter() - t0, last


class _DryLinearFn(torch.autograd.Function):
    """Dry run only: a linear layer whose backward takes its parameter-gradient memory from `_lib.grad_pool.take_for` exactly as the
    convolution nodes do (arch/conv.py), so that the in-place reducer path is the one the launcher rehearsal exercises."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, g):
        from nndetection_amd import _lib as L
        x, w, b = ctx.saved_tensors
        gw, gb = L.grad_pool.take_for([(w, w.numel()), (b, b.numel())], g.device)
        dw = gw.view(w.shape)
        dw += g.t() @ x
        gb += g.sum(0)
        return g @ w, dw, gb


class _DryNet(torch.nn.Module):
    """Stand-in with the three kinds of parameters the reducer has to cope with: a trunk every rank uses, a "regressor"
    that a rank without positive anchors does not use, a `decoder.out.P1`-like layer nobody uses. NOT the product and NOT a
    measurement: `--cpu-dry-run` rehearses the launcher / rendezvous / seeding / reducer / optimizer wiring on CPU tensors over gloo."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.ModuleList([torch.nn.Linear(64, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 64)])
        self.regressor = torch.nn.Linear(64, 64)
        self.out_p1 = torch.nn.Linear(64, 64)

    def never_used_parameters(self):
        return list(self.out_p1.parameters())

    def forward(self, x, positives: bool):
        from nndetection_amd import _lib as L
        L.grad_pool.begin(sum(p.numel() + 64 for p in self.parameters()), x.device, owner=self)
        h = x
        for m in self.trunk:
            h = torch.relu(_DryLinearFn.apply(h, m.weight, m.bias))
        if positives:
            return (h * h).mean() + _DryLinearFn.apply(h, self.regressor.weight, self.regressor.bias).pow(2).mean()
        L.notify_no_grad(list(self.regressor.parameters()))          # what arch/heads.py does on a batch without positive anchors
        return (h * h).mean()


class DryRoute:
    """`Route` for `--cpu-dry-run`: same construction order (seed 0 model -> optimizer -> reducer -> per-rank seed), same step
    (forward, begin_step, backward, finish, optimizer step, scheduler step, zero_grad), same `timed` bracket -- on the stand-in."""

    def __init__(self, rank, world, ddp_factory):
        from nndetection_amd.ptmodule import configure_optimizer
        torch.manual_seed(rank)                           # DIFFERENT initial parameters per rank: the reducer must broadcast rank 0's
        self.net = _DryNet()
        self.opt, self.sched = configure_optimizer(self.net)
        g = torch.Generator().manual_seed(1000 + rank)
        self.data = [torch.randn(16, 64, generator=g) for _ in range(NBATCH)]
        self.positive_free = world > 1 and rank == world - 1
        self.ddp = ddp_factory(self.net) if ddp_factory is not None else None
        self.scaler, self.it = None, 0
        torch.manual_seed(1234 + rank)

    def step(self):
        self.it += 1
        loss = self.net(self.data[self.it % NBATCH], positives=not self.positive_free)
        if self.ddp is not None:
            self.ddp.begin_step()
        loss.backward()
        if self.ddp is not None:
            self.ddp.finish()
        self.opt.step()
        self.sched.step()
        self.opt.zero_grad(set_to_none=True)
        return loss

    timed = None                                          # bound below: Route.timed works on both (its synchronisations are no-ops on the CPU)


DryRoute.timed = Route.timed


def side_route(plan, batch, dtype_name, device, via_plugin, warmup=8, steps=24):
    r = Route(plan, batch, dtype_name, device, 0, via_plugin)
    dt, last = r.timed(warmup, steps)
    out = {"patches_per_s": round(batch * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
           "dtype": dtype_name, "via_plugin": bool(via_plugin), "final_loss": round(float(last.detach().float().item()), 5)}
    if r.scaler is not None:
        out["grad_scale"] = float(r.scaler.get_scale())
    del r
    torch.cuda.empty_cache()
    return out


def step_algorithmic_work(net, plan, batch, esz):
    """Algorithmic FLOPs / HBM bytes of one training step (every convolution reads its input once and writes its output
    once; forward + data gradient + weight gradient = 3 x the forward figures; norm / ReLU / loss passes count as zero bytes).
    Walks the network's own conv blocks with the feature-map sizes of the plan; `decoder.out.P<l>` blocks nobody reads are not
    computed by this implementation and not counted."""
    import math
    from nndetection_amd.arch.conv import BaseConvNormAct
    from nndetection_amd.layout import cpad
    P = tuple(plan["patch_size"])
    strides = net.encoder.get_strides()
    size = [tuple(int(math.ceil(P[a] / st[a])) for a in range(3)) for st in strides]      # feature-map size per stage / level
    flops = byts = 0.0

    def add(m, sp_in, n_dir=3):
        nonlocal flops, byts
        k, s = m.k, m.s
        if m.transposed:
            sp_out = tuple(sp_in[a] * s[a] for a in range(3)); gemm_vox = math.prod(sp_in)
        else:
            sp_out = tuple((sp_in[a] + 2 * m.p[a] - k[a]) // s[a] + 1 for a in range(3)); gemm_vox = math.prod(sp_out)
        flops += n_dir * 2.0 * gemm_vox * math.prod(k) * m.in_channels * m.out_channels
        cin_p = 1 if m.in_channels == 1 else cpad(m.in_channels)
        byts += n_dir * esz * (math.prod(sp_in) * cin_p + math.prod(sp_out) * cpad(m.out_channels))
        return sp_out

    for i, stage in enumerate(net.encoder.stages):
        sp = P if i == 0 else size[i - 1]
        for j, m in enumerate([q for q in stage.modules() if isinstance(q, BaseConvNormAct)]):
            sp = add(m, sp, 2 if (i == 0 and j == 0) else 3)             # the stem has no data gradient
    used = getattr(net.decoder, "used_levels", None)
    convs = lambda mod: [q for q in mod.modules() if isinstance(q, BaseConvNormAct)]
    for name, mod in net.decoder.lateral.items():
        for m in convs(mod):
            add(m, size[int(name[1:])])
    for name, mod in net.decoder.up.items():
        for m in convs(mod):
            add(m, size[int(name[1:])])
    for name, mod in net.decoder.out.items():
        l = int(name[1:])
        if used is None or l in used:
            for m in convs(mod):
                add(m, size[l])
    for l in net.decoder_levels:
        for head in (net.head.classifier, net.head.regressor):
            for m in [q for q in head.modules() if isinstance(q, BaseConvNormAct)]:
                add(m, size[l])
    if net.segmenter is not None:
        add(net.segmenter.conv_out, size[0])
    return flops * batch, byts * batch


def executed_flops_skipped(net, plan, batch, dtype_name):
    """FLOPs of the reference's graph that a TRAINING step of this implementation does not execute as dense convolutions:
    the regressor's output convolution (evaluated at the sampled positives only: forward + both gradients), the dense gradients of the
    classifier's output convolution (sparse at the sampled anchors), and -- 16-bit route -- decoder.out.P0 + lateral.P0 + up.P1, which
    the segmentation branch absorbs into one composed convolution. None when a switch that changes this is set."""
    import math
    from nndetection_amd.arch import heads as H, segmenter as S
    if not (H.SPARSE_OUT and H.SPARSE_REG):
        return None
    P = tuple(plan["patch_size"])
    strides = net.encoder.get_strides()
    size = [tuple(int(math.ceil(P[a] / st[a])) for a in range(3)) for st in strides]
    conv_fl = lambda m, sp: 2.0 * math.prod(sp) * math.prod(m.k) * m.in_channels * m.out_channels
    fl, what = 0.0, []
    for l in net.decoder_levels:
        fl += 3 * conv_fl(net.head.regressor.conv_out, size[l]) + 2 * conv_fl(net.head.classifier.conv_out, size[l])
    what.append("head.regressor.conv_out (x3), head.classifier.conv_out gradients (x2)")
    if dtype_name in ("bf16", "f16") and S.SEG_BRANCH and S.SEG_LATERAL and S.SEG_UP and os.environ.get("NNDET_SEG_FUSED", "1") != "0":
        fl += 3 * conv_fl(net.decoder.out["P0"][0], size[0]) + 3 * conv_fl(net.decoder.lateral["P0"][0], size[0])
        fl += 3 * conv_fl(net.decoder.up["P1"], size[1])
        what.append("decoder.out.P0, decoder.lateral.P0, decoder.up.P1 (x3 each)")
    return {"flops": fl * batch, "what": "; ".join(what)}


def head_trunk_roofline(plan, batch, dtype, device, iters=30):
    """The launch with the most TIME per training step: the shared 128 -> 128 head-trunk convolution over ALL pyramid levels of all
    images as one ragged batch (k_ig3<..., ITEMS>, arch/pyramid.py): 4 such forward launches + 4 data gradients per step."""
    import ctypes
    import math
    from nndetection_amd import _lib as L
    from nndetection_amd.arch import pyramid as PY
    from nndetection_amd.arch.conv import ConvGroupRelu, _packed
    from nndetection_amd.ptmodule import build_model
    Pz = tuple(plan["patch_size"])
    arch = plan["arch"]
    st = [[1, 1, 1]]
    for s_ in arch["strides"]:
        st.append([a * b for a, b in zip(st[-1], s_)])
    lev = [tuple(int(math.ceil(Pz[a] / st[l][a])) for a in range(3)) for l in arch["decoder_levels"]]
    c = arch["head_channels"]
    m = ConvGroupRelu(3, c, c, 3, stride=1, padding=1, add_norm=False, add_act=False).to(device)
    meta = PY.pyramid_meta([(batch, *s_) for s_ in lev])
    x = torch.randn(meta.rows, c, device=device).to(dtype)
    d = PY._items_desc(x, m, meta)
    w0 = _packed(m, 0, m.conv.weight, d, dtype)
    y = torch.empty(meta.rows, d.cout_p, dtype=dtype, device=device)
    it = ctypes.byref(meta.items)
    stq = L.stream()
    f = lambda: L.call("nndet_conv3d_forward_items", ctypes.byref(d), it, L.ptr(x), L.ptr(w0), None, L.ptr(y), None, stq)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * meta.rows * 27 * c * c
    esz = x.element_size()
    byts = meta.rows * 2 * c * esz + 27 * c * c * esz
    tfs = flops / (ms * 1e-3) / 1e12
    # round 5: the FIRST layer of the two trunks is one launch 128 -> 2 x 128 (arch/pyramid.py: _FusedItemsBlockFn): the same kernel with
    # 256 output rows, timed the same way
    m2 = ConvGroupRelu(3, c, 2 * c, 3, stride=1, padding=1, add_norm=False, add_act=False).to(device)
    d2 = PY._items_desc(x, m2, meta)
    w2 = _packed(m2, 0, m2.conv.weight, d2, dtype)
    y2 = torch.empty(meta.rows, d2.cout_p, dtype=dtype, device=device)
    f2 = lambda: L.call("nndet_conv3d_forward_items", ctypes.byref(d2), it, L.ptr(x), L.ptr(w2), None, L.ptr(y2), None, stq)
    for _ in range(10):
        f2()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f2()
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / iters
    byts2 = meta.rows * 3 * c * esz + 27 * c * 2 * c * esz
    tfs2 = 2.0 * flops / (ms2 * 1e-3) / 1e12
    from nndetection_amd.arch.heads import FUSE_CIN
    return {"bound": "mfma", "kernel": "k_ig3<%s, NT=16, ITEMS> conv3d 3x3x3 %d->%d over the pyramid levels %s x batch %d as one ragged batch (forward)"
            % (str(dtype).replace("torch.", ""), c, c, "/".join("x".join(map(str, s_)) for s_ in lev), batch),
            "achieved": round(tfs, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tfs / 2500.0, 4), "traffic": None,
            "algorithmic_flops_per_launch": int(flops), "algorithmic_bytes_per_launch": int(byts), "ms_per_launch": round(float(ms), 4),
            "flop_per_byte": round(flops / byts, 1), "launches_per_step": 4 if FUSE_CIN else 8,
            "fused_first_layer": {"kernel": "the same kernel, %d -> 2 x %d (classifier + regressor c_in in one launch; forward, and 2 x %d -> %d as the "
                                            "data gradient)" % (c, c, c, c), "in_the_step": bool(FUSE_CIN), "launches_per_step": 2 if FUSE_CIN else 0,
                                  "ms_per_launch": round(float(ms2), 4), "achieved": round(tfs2, 1), "frac": round(tfs2 / 2500.0, 4),
                                  "algorithmic_flops_per_launch": int(2 * flops), "algorithmic_bytes_per_launch": int(byts2),
                                  "ms_of_two_separate_launches": round(2 * float(ms), 4)},
            "note": "in isolation; inside the step the classifier / regressor trunks and the weight-gradient stream share the CUs"}


def measure_traffic_pmc(timeout_s=150, kernel="k_ig3r", order="fwd"):
    """HBM bytes per launch of the roofline kernel from the PMC counters, measured NOW: two rocprofv3 passes (FETCH_SIZE and
    WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "Counter slots") over tools/conv_microbench.py e0_32x32_full (the same
    launch as `conv_roofline`), FETCH_SIZE doubled as the guide prescribes for gfx950 (wide coalesced reads are tallied at half
    their bytes), both in KB. Returns None when rocprofv3 is not installed or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.isfile("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("NNDET_BENCH_PMC", "1") == "0":
        return None
    vals = {}
    tmp = tempfile.mkdtemp(prefix="nndet_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", MICRO_ORDER=order, MICRO_ITERS="6")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "conv_microbench.py"), "e0_32x32_full"]
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            tot, n = 0.0, 0
            for dbp in glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True):
                db = sqlite3.connect(dbp)
                cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
                kcol = "kernel_name" if "kernel_name" in cols else "name"
                vcol = "value" if "value" in cols else "counter_value"
                for k, c, v in db.execute(f"select {kcol}, counter_name, {vcol} from counters_collection"):
                    if c == ctr and kernel in k and "k_wgrad3s" not in k:
                        tot += float(v); n += 1
            if n == 0:
                return None
            vals[ctr] = (tot / n, n)
        rd = vals["FETCH_SIZE"][0] * 1024.0 * 2.0
        wr = vals["WRITE_SIZE"][0] * 1024.0
        return {"hbm_bytes": int(rd + wr), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "dispatches": vals["FETCH_SIZE"][1],
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, measured in this run (FETCH_SIZE x 2: gfx950 correction)"}
    except Exception as e:                                        # noqa: BLE001 -- any failure: fall back to the committed figure
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
    traffic, traffic_src = None, None
    head = batch == 4 and tuple(P) == (160, 160, 96) and dtype == torch.bfloat16 and os.environ.get("NNDET_IG3R", "1") != "0"
    if head and MEASURE_PMC[0]:
        pm = measure_traffic_pmc()
        if pm is not None and "hbm_bytes" in pm:
            traffic, traffic_src = pm["hbm_bytes"], pm
        elif pm is not None:
            traffic_src = pm
    if traffic is None:
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        for name in ("round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json"):
            tj = os.path.join(pdir, name)
            if os.path.isfile(tj) and batch == 4 and tuple(P) == (160, 160, 96) and dtype == torch.bfloat16:
                with open(tj) as f:
                    tr = json.load(f)
                    key = "k_ig3r_e0" if os.environ.get("NNDET_IG3R", "1") != "0" and "k_ig3r_e0" in tr else "k_ig3_cfgA_e0"
                    traffic = int(tr[key]["hbm_bytes"])
                    traffic_src = dict(traffic_src or {}, fallback="profiles/%s (committed PMC summary, NOT re-measured in this run)" % name)
                break
    # 432 FLOP per algorithmic byte is above the MFMA/HBM ridge (2500 TF/s / 8 TB/s = 312): the kernel is priced against the
    # dense bf16 MFMA peak; the HBM view (algorithmic bytes / time) is reported next to it
    dn = str(dtype).replace("torch.", "").replace("bfloat16", "bf16").replace("float16", "f16").replace("float32", "f32")
    kname = ("k_ig3r<%s> (persistent, weights in registers)" % dn) if (os.environ.get("NNDET_IG3R", "1") != "0" and dtype != torch.float32) \
        else "k_ig3<%s,WR=1,MT=2,NT=8>" % dn
    return {"bound": "mfma", "kernel": kname + " conv3d 3x3x3 32->32 @%dx%dx%d, batch %d (forward)" % (*P, batch),
            "achieved": round(tfs, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tfs / 2500.0, 4),
            "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flops_per_launch": int(flops), "algorithmic_bytes_per_launch": int(alg_bytes),
            "ms_per_launch": round(float(ms), 4), "algorithmic_GBs": round(gbs, 1), "hbm_frac_of_8TBs": round(gbs / 8000.0, 4),
            "measured_mfma_ceiling_TFs": 1900.0}


def wgrad_roofline(route, plan, batch, dtype, device, in_step=8, alone=12):
    """The kernel that dominates the profile and ends the step: the 3x3x3 / stride 1 weight gradient, k_wgrad3d -- its largest
    launch, 32 -> 32 channels at full resolution (encoder.stages.0.convs.0.1). HIP events bracket exactly that kernel on the stream it
    is launched on (the weight-gradient stream; the library records the caller's events around the one launch whose tile count
    matches, include/nndet_amd.h: nndet_probe_wgrad3d): `in_step` extra training steps after the timed region give its duration INSIDE
    the step (next to the data-gradient chain it shares the chip with), `alone` back-to-back launches its duration alone.
    achieved = algorithmic FLOPs per launch / the in-step duration."""
    import ctypes
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
    from nndetection_amd.layout import cpad
    P = tuple(plan["patch_size"])
    c = plan["arch"]["start_channels"]
    tiles = batch * ((P[0] + 3) // 4) * ((P[1] + 7) // 8) * ((P[2] + 7) // 8)
    lib = L.load()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()                          # (torch creates the hipEvent at the first record; the library re-records them)
    torch.cuda.synchronize()
    h0, h1 = e0.cuda_event, e1.cuda_event

    def probed(fn):
        lib.nndet_probe_wgrad3d(tiles, h0, h1)
        fn()
        torch.cuda.synchronize()
        lib.nndet_probe_wgrad3d(0, None, None)
        return e0.elapsed_time(e1)

    ms_step = None
    if route is not None and os.environ.get("NNDET_WGRAD3D", "1") != "0":
        probed(route.step)
        v = sorted(probed(route.step) for _ in range(in_step))
        ms_step = {"mean": sum(v) / len(v), "median": v[len(v) // 2], "min": v[0], "max": v[-1], "n": len(v)}
    m = ConvInstanceRelu(3, c, c, 3, stride=1, padding=1, add_norm=False, add_act=False).to(device)
    x = torch.randn(batch, *P, cpad(c), device=device).to(dtype)
    d = _desc(x, c, c, m.k, m.s, m.p, False)
    dy = torch.randn(batch, d.out_d, d.out_h, d.out_w, d.cout_p, device=device).to(dtype)
    dw = torch.zeros_like(m.conv.weight)
    wsb = lib.nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
    ws = L.workspace(wsb, x.device)
    st = L.stream()
    call = lambda: L.call("nndet_conv3d_backward_weight", ctypes.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), None, L.ptr(ws), wsb, st)
    for _ in range(20):
        call()
    va = sorted(probed(call) for _ in range(alone))
    ms_alone = sum(va) / len(va)
    del x, dy
    nvox = batch * P[0] * P[1] * P[2]
    esz = torch.tensor([], dtype=dtype).element_size()
    flops = 2.0 * nvox * 27 * c * c
    alg_bytes = 2 * nvox * cpad(c) * esz + 27 * c * c * 4
    ms = ms_step["mean"] if ms_step is not None else ms_alone
    tfs = flops / (ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    if MEASURE_PMC[0] and batch == 4 and P == (160, 160, 96) and dtype == torch.bfloat16:
        pm = measure_traffic_pmc(kernel="k_wgrad3", order="wgrad")      # (k_wgrad3e, or k_wgrad3d under NNDET_WGRAD3D=1)
        if pm is not None and "hbm_bytes" in pm:
            traffic, traffic_src = pm["hbm_bytes"], pm
        else:
            traffic_src = pm
    dn = str(dtype).replace("torch.", "").replace("bfloat16", "bf16").replace("float16", "f16")
    out = {"bound": "mfma", "kernel": "k_wgrad3e<%s> (round 6 form of k_wgrad3d) weight gradient of conv3d 3x3x3 32->32 @%dx%dx%d, batch %d (encoder.stages.0.convs.0.1)" % (dn, *P, batch),
           "achieved": round(tfs, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tfs / 2500.0, 4), "traffic": traffic, "traffic_source": traffic_src,
           "algorithmic_flops_per_launch": int(flops), "algorithmic_bytes_per_launch": int(alg_bytes),
           "ms_per_launch": round(float(ms), 4), "timed": "inside the training step (HIP events on the weight-gradient stream around this one kernel, %d steps after the timed region)"
           % (ms_step["n"] if ms_step else 0) if ms_step is not None else "alone (back-to-back launches)",
           "in_step_ms": {k: (round(v, 4) if k != "n" else v) for k, v in ms_step.items()} if ms_step is not None else None,
           "alone_ms": round(float(ms_alone), 4), "alone_TFLOPs": round(flops / (ms_alone * 1e-3) / 1e12, 1),
           "alone_frac": round(flops / (ms_alone * 1e-3) / 1e12 / 2500.0, 4),
           "flop_per_byte": round(flops / alg_bytes, 1), "measured_mfma_ceiling_TFs": 1900.0,
           "note": "k_wgrad3d is the family with the most kernel time per step (11 uniform + 3 ragged launches); this is its largest launch. Inside the step it "
                   "shares the chip with the data-gradient chain of the main stream, so its in-step duration is longer than alone."}
    if traffic is not None:
        out["traffic_over_algorithmic"] = round(traffic / alg_bytes, 3)
    return out


def box_microbench_config5(device):
    """BASELINE.json configs[4] at its stated sizes: pairwise IoU / GIoU [2000 x 100000], ATSS with 2 000 GT x 5 levels
    x 100 000 anchors (27 anchors per location, 4 candidates), NMS on 1 000 / 10 000 / 100 000 boxes with distinct scores. Parity at these
    sizes is the tests' job (tests/test_boxes_gpu.py, tests/test_atss_gpu.py: bit-exact); here they are only timed."""
    from nndetection_amd.core.boxes import box_iou, generalized_box_iou, nms, ATSSMatcher

    def rb(rng, n):
        c = rng.uniform(0, 160.0, (n, 3)); s_ = rng.uniform(2.0, 26.0, (n, 3))
        return np.stack([c[:, 0] - s_[:, 0] / 2, c[:, 1] - s_[:, 1] / 2, c[:, 0] + s_[:, 0] / 2, c[:, 1] + s_[:, 1] / 2,
                         c[:, 2] - s_[:, 2] / 2, c[:, 2] + s_[:, 2] / 2], 1).astype(np.float32)

    def timeit(fn, iters=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    rng = np.random.default_rng(0)
    a, g = torch.from_numpy(rb(rng, 100000)).to(device), torch.from_numpy(rb(rng, 2000)).to(device)
    out = {"sizes": "SURVEY 8d config 5: anchors [100000, 6] per level x 5 levels, GT [2000, 6]"}
    pairs = 2000 * 100000
    for key, fn in (("box_iou", box_iou), ("generalized_box_iou", generalized_box_iou)):
        dt = timeit(lambda: fn(g, a))
        out[key] = {"ms": round(dt * 1e3, 4), "Gpairs_per_s": round(pairs / dt / 1e9, 1), "algorithmic_GBs": round((4 * pairs + 24 * 102000) / dt / 1e9, 1),
                    "hbm_frac_of_8TBs": round((4 * pairs + 24 * 102000) / dt / 8e12, 4)}
    out["nms"] = {}
    for n in (1000, 10000, 100000):
        b = torch.from_numpy(rb(rng, n)).to(device)
        sc = torch.from_numpy(((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)).to(device)
        dt = timeit(lambda: nms(b, sc, 0.6), iters=5 if n < 100000 else 2)
        out["nms"][str(n)] = {"ms": round(dt * 1e3, 4), "Mboxes_per_s": round(n / dt / 1e6, 2)}
    rng5 = np.random.default_rng(0)
    a5 = torch.from_numpy(np.concatenate([rb(rng5, 100000) for _ in range(5)], 0)).to(device)
    g5 = torch.from_numpy(rb(rng5, 2000)).to(device)
    m = ATSSMatcher(num_candidates=4, center_in_gt=False)
    dt = timeit(lambda: m(g5, a5, [100000] * 5, 27), iters=3)
    out["atss"] = {"ms": round(dt * 1e3, 3), "G_gt_anchor_pairs_per_s": round(2000 * 500000 / dt / 1e9, 1), "gt": 2000, "anchors": 500000, "levels": 5}
    return out


def config3_lidc192(device, warmup=4, steps=8):
    """BASELINE.json configs[3] on ONE GPU's share: Task012_LIDC-like plan, 192x192x128 patches, fp16 convolutions (+ GradScaler, the
    reference's precision=16), batch 4 per GPU: ms per step and peak HBM (SURVEY 8d config 4 asks for the peak per GPU)."""
    from nndetection_amd.plans import get_plan
    plan = get_plan("lidc192")
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(device)
    r = Route(plan, 4, "f16", device, 0, False)
    dt, last = r.timed(warmup, steps)
    out = {"plan": "lidc192", "patch": list(plan["patch_size"]), "batch_per_gpu": 4, "dtype": "f16 + GradScaler", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "patches_per_s": round(4 * steps / dt, 2),
           "peak_hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2), "final_loss": round(float(last.detach().float().item()), 5)}
    del r
    torch.cuda.empty_cache()
    return out


MEASURE_PMC = [True]
STEP_PMC = [None]


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


def cpu_baseline(plan, device=None, batch=2):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference, oracle/retina_torch.py; kind "port") on this host:
    a batch of TWO patches forward + ATSS + losses + backward (a bounded sample of the same workload; two, not one, so that the
    batch-level hard-negative mining and batch_dice of the parity check below couple images as they do at the benchmarked batch). The thread count is chosen by a
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
    x, tg = synth_batch(plan, batch, torch.float32, "cpu", 0)
    orig = torch.randperm
    torch.randperm = _det_randperm                          # the sampler's permutation, fixed for the parity comparison
    try:
        t0 = time.perf_counter()
        losses, _ = net.train_step(x, tg, evaluation=False)
        sum(losses.values()).backward()
        dt = time.perf_counter() - t0
        out = {"value": round(batch / dt, 4), "unit": "patches/s", "cores": best, "threads": best, "logical_cpus": logical,
               "kind": "port", "thread_sweep_proxy_s": sweep,
               "sample": "one batch of %d patches %dx%dx%d fp32: forward + ATSS + losses + backward of the CPU oracle (%.1f s, %d threads)"
                         % (batch, *P, dt, best)}
        parity = None
        if device is not None:
            from nndetection_amd.ptmodule import build_model
            hip = build_model(plan)
            hip.load_state_dict(net.state_dict())
            hip.to(device)
            tgd = {"target_boxes": [b.to(device) for b in tg["target_boxes"]], "target_classes": [c.to(device) for c in tg["target_classes"]],
                   "target_seg": tg["target_seg"].to(device)}
            lg, _ = hip.train_step(x.to(device), tgd, evaluation=False)
            lc = {k: float(v.detach()) for k, v in losses.items()}
            lgv = {k: float(v) for k, v in lg.items()}
            diff = max(abs(lc[k] - lgv[k]) for k in lc)
            parity = {"what": "losses of the HIP fp32 kernels vs the CPU oracle on the cpu_baseline batch of %d patches (same weights, same sampler permutation)" % batch,
                      "patches": batch,
                      "losses_cpu_oracle": {k: round(v, 6) for k, v in lc.items()}, "losses_hip_fp32": {k: round(v, 6) for k, v in lgv.items()},
                      "max_abs_diff": diff, "tolerance": 1e-4, "ok": bool(diff <= 1e-4 and set(lc) == set(lgv))}
            del hip
    finally:
        torch.randperm = orig
    return out, parity


def torch_rocm_baseline_child(plan_name, batch, warmup=3, steps=10):
    """stock PyTorch-ROCm on THIS GPU -- the oracle's module tree (plain torch.nn Conv3d / ConvTranspose3d /
    InstanceNorm3d / GroupNorm / ReLU, i.e. MIOpen + ATen kernels, the layers the reference builds) with the reference's training
    arithmetic: torch.autocast(float16) + torch.amp.GradScaler (scripts/train.py:277-278 precision=16), losses as the reference
    computes them, torch.optim.SGD(nesterov). Runs in a child process (own MIOpen state, bounded by a timeout) and prints one JSON
    object. The ATSS assignment (numpy in the oracle) is computed ONCE on the host before the timed loop and handed in as device
    tensors: the baseline's time contains no target assignment at all (that favours the baseline). Checker / context leg only."""
    # NNDET_TORCH_BASELINE_MODE=fair: MIOpen's NORMAL find (it benchmarks its solvers per shape; the find-db goes to
    # /tmp), channels_last_3d tensors and >= 10 warm-up steps -- minutes of set-up, so bench.py's default leg stays "fast" (FAST find,
    # NCDHW, 3 warm-up steps) and quotes the committed fair measurement next to it (profiles/round5_torch_rocm_baseline_fair.json).
    tb_mode = os.environ.get("NNDET_TORCH_BASELINE_MODE", "fast")       # fast | normal (NORMAL find, NCDHW) | fair (NORMAL find, channels_last_3d)
    fair = tb_mode == "fair"
    if tb_mode in ("fair", "normal"):
        os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")
        os.environ.setdefault("MIOPEN_USER_DB_PATH", "/tmp/miopen_userdb")
        os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
        warmup = max(warmup, 10)
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")          # no exhaustive per-shape kernel search inside a bounded leg
    os.environ.setdefault("MIOPEN_LOG_LEVEL", "1")
    from oracle.retina_torch import OracleRetinaUNet
    from nndetection_amd.plans import MODEL_CFG_V001, TRAINER_CFG_V001, get_plan
    plan = get_plan(plan_name)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001)
    x, tg = synth_batch(plan, batch, torch.float32, "cpu", 1000)
    assigned = tuple(t.to(dev) for t in net.assign(x.shape, tg))
    net.to(dev)
    x = x.to(dev)
    if fair:
        net.to(memory_format=torch.channels_last_3d)
        x = x.contiguous(memory_format=torch.channels_last_3d)
    tgd = {"target_boxes": None, "target_classes": None, "target_seg": tg["target_seg"].to(dev)}
    opt = torch.optim.SGD(net.parameters(), lr=TRAINER_CFG_V001["initial_lr"], momentum=0.9, nesterov=True, weight_decay=3e-5)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14)

    def step():
        with torch.autocast("cuda", dtype=torch.float16):
            losses, _ = net.train_step(x, tgd, evaluation=False, assigned=assigned)
            loss = sum(losses.values())
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad(set_to_none=True)
        return loss

    t_first = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    for _ in range(max(0, warmup - 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"value": round(batch * steps / dt, 3), "unit": "patches/s", "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
                      "warmup": warmup, "batch": batch, "first_step_s": round(t_first, 1), "final_loss": round(float(last.detach()), 5),
                      "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                      "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE"), "memory_format": "channels_last_3d" if fair else "contiguous (NCDHW)",
                      "mode": tb_mode, "torch": torch.__version__}), flush=True)


def torch_rocm_baseline(plan_name, batch, timeout_s=240):
    """Runs `torch_rocm_baseline_child` in a subprocess; a failure or a timeout costs the leg, never the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--torch-baseline-child", "--plan", plan_name, "--batch", str(batch)]
    base = {"kind": "stock PyTorch-ROCm (MIOpen / ATen) on this GPU: the oracle's torch.nn module tree, fp16 autocast + GradScaler + "
                    "torch.optim.SGD(nesterov), batch %d; target assignment precomputed and excluded; checker leg, outside the timed region" % batch}
    try:
        r = subprocess.run(cmd, cwd=ROOT, timeout=timeout_s, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                res = dict(base, **json.loads(line))
                if res.get("mode") == "fast":                  # the committed NORMAL-find measurements (minutes of MIOpen search each)
                    for key, fn in (("fair_reference", "round5_torch_rocm_baseline_fair.json"), ("normal_find_reference", "round5_torch_rocm_baseline_normal.json")):
                        fpath = os.path.join(ROOT, "profiles", fn)
                        if not os.path.isfile(fpath):
                            continue
                        try:
                            with open(fpath) as f:
                                fr = json.load(f)
                            res[key] = {k: fr.get(k) for k in ("value", "ms_per_step", "miopen_find_mode", "memory_format", "warmup", "steps", "first_step_s")}
                            res[key]["file"] = "profiles/" + fn
                        except Exception:                     # noqa: BLE001
                            pass
                return res
        return dict(base, error="rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
    except Exception as e:                                        # noqa: BLE001
        return dict(base, error="%s: %s" % (type(e).__name__, str(e)[:200]))


def inference_rate(net, x, iters=5):
    """`inference_step` (forward + fused post-processing: top-k on the logits, decode of the survivors, batched NMS) per image."""
    was_training = net.training
    net.eval()                                   # as nnDetection's validation / predict loops call it (Lightning puts the module in eval mode)
    try:
        with torch.no_grad():
            net.inference_step(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                out = net.inference_step(x)
            torch.cuda.synchronize()
    finally:
        net.train(was_training)
    dt = (time.perf_counter() - t0) / iters
    return {"ms_per_image": round(dt * 1e3 / x.shape[0], 3), "batch": int(x.shape[0]), "ms_per_batch": round(dt * 1e3, 3),
            "detections": [int(b.shape[0]) for b in out["pred_boxes"]]}


def dry_run(args, world, rank, force_dist):
    """`--cpu-dry-run` (multi-GPU without hardware). Everything `main()` does around the step for N ranks -- environment
    (RANK / WORLD_SIZE / MASTER_*), process group, reducer with the NNDET_DDP_* knobs and its cross-rank layout check, the step loop,
    barrier + max-over-ranks timing, one JSON line from rank 0 -- with the gloo backend and a CPU stand-in network. Adds what a
    rehearsal can prove: every rank ends with bit-identical parameters although they started different and one rank had no positives."""
    import hashlib
    from nndetection_amd.ddp import GradAllReducer
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ddp_factory = None
    if world > 1 or force_dist:
        ddp_factory = lambda net: GradAllReducer(net, force_overlap=force_dist, overlap=os.environ.get("NNDET_DDP_OVERLAP", "1") != "0", profile=False)
    route = DryRoute(rank, world, ddp_factory)
    dt, last = route.timed(args.warmup, args.steps, world)
    digest = hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in route.net.parameters())).digest()
    mine = torch.tensor(list(digest), dtype=torch.uint8)
    digests, tt, hooks = [mine], torch.tensor([dt], dtype=torch.float64), None
    if world > 1:
        digests = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(digests, mine)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        flag = torch.tensor([int(all(route.ddp.launched_from_hooks))], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        hooks = bool(flag.item())
    if rank == 0:
        red = route.ddp
        print(json.dumps({
            "metric": "patches/sec (fwd+bwd) RetinaUNet", "value": None, "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(float(tt.item()) / max(1, args.steps) * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "none", "dry_run": True,
            "config": {"workload": "CPU DRY RUN of the N-rank launch on a stand-in network over gloo: NOT a measurement of the hot path", "parallelism": "dp%d" % world},
            "params_identical_on_all_ranks": len({bytes(d.tolist()) for d in digests}) == 1, "positive_free_rank": world - 1 if world > 1 else None,
            "all_buckets_launched_from_hooks_on_all_ranks": hooks, "final_loss": round(float(last.detach()), 6),
            "ddp": None if red is None else {"buckets": len(red.buckets), "bucket_numel": [b.numel for b in red.buckets], "layout_digest": red.layout_digest[:16],
                                             "first_bucket_mb": red.first_bucket_mb, "bucket_mb": red.bucket_mb, "in_place": bool(red.inplace),
                                             "copied_last": red.copied_last, "exposed_allreduce_ms": None, "backward_to_ready_ms": None}}), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--plan", default="luna160")
    ap.add_argument("--batch", type=int, default=None, help="patches per GPU (default: the plan's batch size, 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--via-plugin", action="store_true", help="the timed steps go through the plugin's training_step (batch dicts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-routes", action="store_true", help="skip the short plugin / fp32 / fp16 comparison runs")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes for roofline.traffic")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / NMS / CPU legs (only the timed steps)")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the stock PyTorch-ROCm leg (torch_rocm_baseline)")
    ap.add_argument("--torch-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="NOT a measurement: rehearse the N-rank launch (rendezvous, per-rank seeds, reducer layout check, in-place bucket path, "
                         "a positive-free rank, optimizer) on CPU tensors over gloo with a stand-in network; prints a JSON line with value null")
    args = ap.parse_args()
    if args.torch_baseline_child:
        from nndetection_amd.plans import get_plan as _gp
        torch_rocm_baseline_child(args.plan, args.batch or _gp(args.plan)["batch_size"])
        return
    MEASURE_PMC[0] = not args.no_pmc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("NNDET_BENCH_FORCE_DIST") == "1"     # exercise the RCCL / bucket path at world size 1 (testing)
    if args.cpu_dry_run:
        return dry_run(args, world, rank, force_dist)
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
    from nndetection_amd.ddp import GradAllReducer

    plan = get_plan(args.plan)
    batch = args.batch or plan["batch_size"]
    dtype = _TORCH_DT[args.dtype]
    ddp_factory = None
    if world > 1 or force_dist:
        # bucket sizes / dtype: NNDET_DDP_FIRST_MB, NNDET_DDP_BUCKET_MB, NNDET_DDP_BF16 (the sweep switches, nndetection_amd/ddp.py)
        ddp_factory = lambda net: GradAllReducer(net, force_overlap=force_dist, overlap=os.environ.get("NNDET_DDP_OVERLAP", "1") != "0",
                                                 profile=os.environ.get("NNDET_DDP_PROFILE", "1") != "0")
    route = Route(plan, batch, args.dtype, device, rank, args.via_plugin, ddp_factory)
    net = route.net
    dt, last = route.timed(args.warmup, args.steps, world)
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(last.detach().float().item())
    peak_gb = torch.cuda.max_memory_allocated(device) / 2 ** 30
    # two more timed blocks of the same K steps (no warm-up in between): `value` stays the FIRST block (the driver-flag run); the
    # median of the three goes into `timed_blocks` so that one slow block (or a slow chip's first seconds) is visible as such
    blocks_ms = [dt / args.steps * 1e3]
    if not args.no_extras:
        for _ in range(2):
            dtb, _l = route.timed(0, args.steps, world)
            if world > 1:
                tb = torch.tensor([dtb], device=device, dtype=torch.float64)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                dtb = float(tb.item())
            blocks_ms.append(dtb / args.steps * 1e3)
    ddp_prof = None
    if route.ddp is not None and route.ddp.profile:
        route.ddp.profile_collect()                      # (events of the timed steps; read outside the timed region)
        ddp_prof = route.ddp.profile_summary()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "patches/sec (fwd+bwd) %dx%dx%d RetinaUNet" % tuple(plan["patch_size"]), "value": round(batch * world * args.steps / dt, 3),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: Task016_Luna-like plan (SURVEY 8)" if args.plan == "luna160" else
                                    "plan '%s' (not the headline configuration)" % args.plan) +
                                   ", RetinaUNetV001 train step (fwd + ATSS + losses + bwd + SGD), %dx%dx%d patches" % tuple(plan["patch_size"]),
                       "plan": args.plan, "batch_per_gpu": batch, "global_batch": batch * world,
                       "parallelism": "dp%d" % world, "params": sum(p.numel() for p in net.parameters()),
                       "distinct_batches": NBATCH,
                       "route": "plugin training_step (batch dict -> device-side targets -> train_step)" if args.via_plugin else
                                "BaseRetinaNet.train_step on prepared targets",
                       "loss_scaling": "torch.amp.GradScaler (sync-free: fused SGD takes scale / found_inf on the device)" if route.scaler is not None else None},
            "final_loss": round(loss_val, 5), "peak_hbm_gib": round(peak_gb, 2),
        }
        if len(blocks_ms) > 1:
            med = sorted(blocks_ms)[len(blocks_ms) // 2]
            out["timed_blocks"] = {"ms_per_step": [round(b, 3) for b in blocks_ms], "median_ms_per_step": round(med, 3),
                                   "median_patches_per_s": round(batch * world / (med * 1e-3), 2),
                                   "note": "three back-to-back blocks of --steps steps; `value` is the first (the driver-flag run)"}
        if route.scaler is not None:
            out["grad_scale"] = float(route.scaler.get_scale())
        if ddp_prof is not None:
            out["ddp"] = ddp_prof
        # whole-step roofline: algorithmic work of all convolutions / measured step time, per GPU
        fl, by = step_algorithmic_work(net, plan, batch, 4 if args.dtype == "f32" else 2)
        peak_tf = 2500.0 if args.dtype != "f32" else 157.3
        out["step_roofline"] = {"algorithmic_flops_per_step": int(fl), "algorithmic_bytes_per_step": int(by),
                                "achieved_TFLOPs": round(fl / (ms_step * 1e-3) / 1e12, 1), "mfma_peak_TFLOPs": peak_tf,
                                "mfma_frac": round(fl / (ms_step * 1e-3) / 1e12 / peak_tf, 4),
                                "achieved_GBs": round(by / (ms_step * 1e-3) / 1e9, 1), "hbm_peak_GBs": 8000.0,
                                "hbm_frac": round(by / (ms_step * 1e-3) / 1e9 / 8000.0, 4),
                                "floor_ms_mfma": round(fl / (peak_tf * 1e12) * 1e3, 3), "floor_ms_hbm": round(by / 8e12 * 1e3, 3),
                                "note": "convolutions only: each reads its input and writes its output once, forward + data gradient + weight gradient; "
                                        "norm / ReLU / loss / optimizer passes count as zero algorithmic bytes"}
        sr = out["step_roofline"]
        skipped = executed_flops_skipped(net, plan, batch, args.dtype)
        if skipped is not None:
            ex = fl - skipped["flops"]
            sr["executed_flops_per_step"] = int(ex)
            sr["mfma_frac_executed"] = round(ex / (ms_step * 1e-3) / 1e12 / peak_tf, 4)
            sr["executed_note"] = ("algorithmic FLOPs minus the dense layers this implementation does not run in a training step: " + skipped["what"] +
                                   " (the kernels that replace them -- composed 32(+32)->1 and half-resolution convolutions, <= 170 sparse rows -- are not added back, "
                                   "so this is a lower bound of what the matrix cores execute)")
        if not args.no_pmc and not args.no_extras and world == 1:
            # the step's REAL HBM traffic: two rocprofv3 --pmc passes over a short run of this same script (tools/step_traffic.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import step_traffic
                extra = ["--plan", args.plan, "--dtype", args.dtype] + (["--batch", str(batch)] if args.batch else []) + (["--via-plugin"] if args.via_plugin else [])
                st = step_traffic.measure(4, 2, extra) if os.environ.get("NNDET_BENCH_PMC", "1") != "0" else None
            except Exception as e:                                           # noqa: BLE001
                st = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
            if st is not None and "hbm_bytes_per_step" in st:
                STEP_PMC[0] = st
                sr["traffic"] = st["hbm_bytes_per_step"]
                sr["traffic_over_algorithmic"] = round(st["hbm_bytes_per_step"] / by, 3)
                sr["traffic_GBs"] = round(st["hbm_bytes_per_step"] / (ms_step * 1e-3) / 1e9, 1)
                sr["traffic_hbm_frac"] = round(st["hbm_bytes_per_step"] / (ms_step * 1e-3) / 1e9 / 8000.0, 4)
                sr["traffic_source"] = st
            else:
                sr["traffic"] = None
                sr["traffic_source"] = st
                pj = os.path.join(ROOT, "profiles", "round4_step_traffic.json")
                if os.path.isfile(pj) and args.plan == "luna160" and batch == 4 and args.dtype == "bf16":
                    with open(pj) as f:
                        cj = json.load(f)
                    if "hbm_bytes_per_step" in cj:
                        sr["traffic"] = cj["hbm_bytes_per_step"]
                        sr["traffic_over_algorithmic"] = round(cj["hbm_bytes_per_step"] / by, 3)
                        sr["traffic_source"] = dict(st or {}, fallback="profiles/round4_step_traffic.json (committed PMC summary, NOT re-measured in this run)")
        if not args.no_extras:
            x_inf = route.batch["data"].to(dtype) if args.via_plugin else route.x
            out["inference"] = inference_rate(net, x_inf)
            # the dominant kernel (k_wgrad3d), timed INSIDE extra training steps of the same route and alone -- before the route goes away
            try:
                out["roofline"] = wgrad_roofline(route if (world == 1 and dtype != torch.float32) else None, plan, batch, dtype, device)
            except Exception as e:                                           # noqa: BLE001 -- never lose the headline to a side leg
                out["roofline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            del route, x_inf, net
            torch.cuda.empty_cache()
            # the kernel with the most FLOPs per launch, alone: the `roofline` block of rounds 1-5, kept as a chip probe (it did not
            # change since round 2: its alone-time compares the chips of two runs)
            pmc_keep = MEASURE_PMC[0]
            MEASURE_PMC[0] = False                                           # (its traffic: the committed PMC summary; the live passes go to k_wgrad3d)
            out["roofline_ig3r"] = conv_roofline(plan, batch, dtype, device)      # rank 0's GPU; the other ranks are done
            MEASURE_PMC[0] = pmc_keep
            out["chip_probe"] = {"k_ig3r_forward_alone_ms": out["roofline_ig3r"]["ms_per_launch"], "k_ig3r_TFLOPs": out["roofline_ig3r"]["achieved"],
                                 "k_wgrad3d_alone_ms": out["roofline"].get("alone_ms"),
                                 "note": "alone-times of two MFMA-bound kernels on this chip: compare across runs before comparing `value` (the pool's chips differ by +-4 %)"}
            out["roofline_dominant"] = head_trunk_roofline(plan, batch, dtype, device)
            if STEP_PMC[0] is not None:
                # L2-miss traffic of the ragged head-trunk launches INSIDE the training step (4 forward + 4 data-gradient launches), from the
                # whole-step PMC passes above: per launch, next to the algorithmic bytes (FETCH_SIZE counts Infinity-Cache hits too)
                for kn, kv in STEP_PMC[0].get("kernels", {}).items():
                    if "k_ig3<" in kn and "IgItems" in kn and kn.rstrip().endswith("true, false>(IgArgs, IgItems)") and kv["launches_per_step"] >= 5.5:
                        rd = out["roofline_dominant"]
                        rd["traffic"] = int((kv["read_MB"] + kv["write_MB"]) * 1e6 / kv["launches_per_step"])
                        alg = rd["algorithmic_bytes_per_launch"]
                        ff = rd.get("fused_first_layer") or {}
                        if ff.get("in_the_step"):                 # the kernel's launches of a step: 4 of 128 -> 128 and 2 fused ones
                            alg = (4 * alg + 2 * ff["algorithmic_bytes_per_launch"]) / 6.0
                        rd["traffic_over_algorithmic"] = round(rd["traffic"] / alg, 2)
                        rd["traffic_source"] = ("whole-step PMC passes (step_roofline.traffic_source), this kernel's %.1f launches per step; FETCH_SIZE = L2 misses: "
                                                "the 64-row blocks and halos of neighbouring tiles re-fetch the 45 MB input, mostly from the Infinity Cache" % kv["launches_per_step"])
                        break
                for k_ in ("kernels",):
                    sr_src = out["step_roofline"].get("traffic_source")
                    if isinstance(sr_src, dict):
                        sr_src.pop(k_, None)                  # (the per-kernel list is only used here; the line stays readable)
            out["nms"] = nms_rate(device)
            try:
                out["box_microbench"] = box_microbench_config5(device)       # BASELINE.json configs[4] at its stated sizes
            except Exception as e:                                           # noqa: BLE001
                out["box_microbench"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            if world == 1 and args.plan == "luna160" and not args.no_routes:
                try:
                    out["config3_lidc192"] = config3_lidc192(device)         # BASELINE.json configs[3] (one GPU's share)
                except Exception as e:                                       # noqa: BLE001
                    out["config3_lidc192"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            if world == 1 and not args.no_routes:
                # the other routes on short runs in this process: what the drop-in (plugin) delivers next to the direct route, the
                # exact-fp32 kernels (north_star's 1e-4 parity path), fp16 + GradScaler (the reference's own precision=16)
                routes = {}
                for key, dn, vp in (("plugin_" + args.dtype, args.dtype, True), ("direct_f32", "f32", False),
                                    ("plugin_f16_gradscaler", "f16", True), ("direct_" + args.dtype, args.dtype, False)):
                    if key == "direct_" + args.dtype and not args.via_plugin:
                        continue                                             # that IS the headline
                    if key == "plugin_" + args.dtype and args.via_plugin:
                        continue
                    try:
                        routes[key] = side_route(plan, batch, dn, device, vp)
                        routes[key]["vs_headline"] = round(routes[key]["patches_per_s"] / out["value"], 4)
                    except Exception as e:                                   # noqa: BLE001 -- a side leg must not cost the headline
                        routes[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
                out["routes"] = routes
            if not args.no_cpu_baseline and world == 1:                      # the CPU leg only at N = 1 (rank 0)
                torch.cuda.empty_cache()
                out["cpu_baseline"], out["parity_check"] = cpu_baseline(plan, device)
            if not args.no_torch_baseline and world == 1:
                torch.cuda.empty_cache()
                out["torch_rocm_baseline"] = torch_rocm_baseline(args.plan, batch)
                if "value" in out["torch_rocm_baseline"]:
                    out["torch_rocm_baseline"]["headline_over_this"] = round(out["value"] / out["torch_rocm_baseline"]["value"], 2)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
