"""CPU: the reference-free parts of the plugin step mixin (`nndetection_amd.ptmodule.RetinaUNetAMDSteps`): lazy loss scalars, the
precision -> activation dtype mapping, the Lightning-facing data-parallel hooks under a world-size-2 gloo group (driven through a
Lightning stand-in, one rank without positives = without a gradient for the "regressor"), deferred targets bookkeeping.
The step bodies themselves need the HIP library: tests/test_plugin_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_float_behaves_like_the_reference_item():
    """retinaunet/base.py:154,205-218: the per-loss entries are consumed by `np.mean(list)` in training_epoch_end and by format
    strings; Lightning leaves non-tensor entries of the step output alone."""
    from nndetection_amd.ptmodule import lazy_items, LazyFloat
    d = lazy_items({"reg": torch.tensor(0.25), "cls": torch.tensor(1.5, dtype=torch.float64), "seg_ce": torch.tensor([0.5]).sum()})
    assert list(d) == ["reg", "cls", "seg_ce"] and all(isinstance(v, LazyFloat) for v in d.values())
    assert float(d["reg"]) == 0.25 and d["cls"].item() == 1.5
    assert np.mean([d["reg"], d["cls"]]) == 0.875 and np.asarray(d["seg_ce"]).dtype == np.float64
    assert f"{d['cls']:0.3f}" == "1.500" and d["reg"] + 1 == 1.25 and 2 * d["reg"] == 0.5 and d["reg"] < d["cls"]
    assert lazy_items({}) == {}


def test_compute_dtype_mapping(monkeypatch):
    from nndetection_amd.ptmodule import RetinaUNetAMDSteps

    class M(RetinaUNetAMDSteps):
        def __init__(self, **cfg):
            self.trainer_cfg = cfg

    x = torch.zeros(1)
    monkeypatch.delenv("NNDET_AMD_DTYPE", raising=False)
    assert M(precision=32).amd_compute_dtype(x) == torch.float32
    assert M(precision=16).amd_compute_dtype(x) == torch.float32          # a CPU batch is never cast (no HIP path there anyway)
    assert M(precision=16, amd_dtype="bf16").amd_compute_dtype(x) == torch.bfloat16
    assert M(precision=32, amd_dtype="fp16").amd_compute_dtype(x) == torch.float16
    monkeypatch.setenv("NNDET_AMD_DTYPE", "f32")
    assert M(precision=16, amd_dtype="bf16").amd_compute_dtype(x) == torch.float32      # the environment wins
    monkeypatch.setenv("NNDET_AMD_DTYPE", "int8")
    with pytest.raises(ValueError):
        M().amd_compute_dtype(x)


def test_split_counts_and_error_flags():
    from nndetection_amd.core.targets import _split
    boxes = torch.arange(2 * 4 * 6, dtype=torch.float32).view(2, 4, 6)
    classes = torch.arange(8).view(2, 4)
    ids = torch.arange(8, dtype=torch.int32).view(2, 4)
    b, c, i = _split(boxes, classes, ids, [3, 0, 0])
    assert b[0].shape == (3, 6) and b[1].shape == (0, 6) and c[0].tolist() == [0, 1, 2] and i[1].numel() == 0
    with pytest.raises(KeyError):
        _split(boxes, classes, ids, [1, 1, 2])
    with pytest.raises(KeyError):
        _split(boxes, classes, ids, [1, 1, 1])


class _Net(nn.Module):
    """Stand-in for BaseRetinaNet: a trunk every rank uses, a "regressor" only ranks with positives use, a layer nobody uses."""

    def __init__(self):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
        self.regressor = nn.Linear(4, 4)
        self.out_p1 = nn.Linear(4, 4)

    def never_used_parameters(self):
        return list(self.out_p1.parameters())


def _hooks_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from nndetection_amd.ptmodule import RetinaUNetAMDSteps
    from nndetection_amd.ddp import GradAllReducer

    class LightningStandIn(nn.Module):                     # what pl.LightningModule contributes to the MRO: the two hooks, no-ops
        calls = []

        def on_fit_start(self):
            self.calls.append("fit")

        def on_after_backward(self):
            self.calls.append("bwd")

    class Module(RetinaUNetAMDSteps, LightningStandIn):
        def __init__(self):
            super().__init__()
            torch.manual_seed(rank)                        # different initial parameters per rank: on_fit_start must broadcast rank 0's
            self.model = _Net()
            self.trainer_cfg = {"amd_ddp_first_bucket_mb": 1e-4, "amd_ddp_bucket_mb": 2e-4}

    mod = Module()
    mod.on_fit_start()
    assert isinstance(mod._amd_reducer, GradAllReducer) and len(mod._amd_reducer.buckets) >= 2
    w0 = mod.model.trunk[0].weight.detach().clone()
    res = []
    for step in range(2):                                  # two optimisation steps: bucket state must reset between them
        torch.manual_seed(100 + rank + 10 * step)
        h = mod.model.trunk(torch.randn(5, 8))
        loss = h.sum() if rank == 0 else mod.model.regressor(h).sum()       # rank 0: "no positive anchors" -> no regressor gradient
        loss.backward()
        mod.on_after_backward()
        res.append([p.grad.detach().numpy().copy() for p in mod.model.parameters()])
        mod.zero_grad(set_to_none=True)
    # under Lightning's own DDP strategy the hooks must stay out of the way
    class _Strategy:
        pass
    _Strategy.__name__ = "DDPStrategy"
    mod2 = Module()
    mod2.trainer = type("T", (), {"strategy": _Strategy(), "model": None})()
    mod2.on_fit_start()
    q.put((rank, w0.numpy(), res, list(LightningStandIn.calls), mod2._amd_reducer is None))
    dist.barrier()
    dist.destroy_process_group()


def test_lightning_hooks_drive_the_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_hooks_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    (_, w0a, ga, calls, skip_a), (_, w0b, gb, _, skip_b) = res
    assert np.array_equal(w0a, w0b), "on_fit_start did not broadcast rank 0's parameters"
    assert calls[:1] == ["fit"] and calls.count("bwd") == 2, calls          # the parent hooks still run
    assert skip_a and skip_b, "the reducer must not be installed under Lightning's DDP strategy"
    # reference: mean of the two local gradients, the regressor zero-filled on rank 0, out_p1 zero everywhere
    torch.manual_seed(0)
    net = _Net()
    for step in range(2):
        want = []
        for rank in range(2):
            net.zero_grad(set_to_none=True)
            torch.manual_seed(100 + rank + 10 * step)
            h = net.trunk(torch.randn(5, 8))
            (h.sum() if rank == 0 else net.regressor(h).sum()).backward()
            want.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net.parameters()])
        for a, b, w0_, w1_ in zip(ga[step], gb[step], *want):
            assert np.allclose(a, b, atol=1e-6), "ranks disagree after on_after_backward"
            assert np.allclose(a, ((w0_ + w1_) / 2).numpy(), atol=1e-6)
    assert float(np.abs(ga[0][-1]).max()) == 0.0 and float(np.abs(ga[0][-2]).max()) == 0.0


def test_bucket_sweep_switches(monkeypatch):
    """NNDET_DDP_FIRST_MB / NNDET_DDP_BUCKET_MB / NNDET_DDP_BF16: the knobs bench.py's N > 1 runs read for tuning against xGMI."""
    from nndetection_amd.ddp import GradAllReducer
    torch.manual_seed(0)
    model = nn.Sequential(*[nn.Linear(64, 64) for _ in range(8)])           # 8 x 16.6 KB of gradients
    monkeypatch.setenv("NNDET_DDP_FIRST_MB", "0.01")
    monkeypatch.setenv("NNDET_DDP_BUCKET_MB", "0.04")
    a = GradAllReducer(model, force_overlap=True)
    assert len(a.buckets) >= 3 and a.buckets[0].numel * 4 <= 0.02 * 2 ** 20 and a.bucket_dtype == torch.float32
    monkeypatch.setenv("NNDET_DDP_BUCKET_MB", "100")
    monkeypatch.setenv("NNDET_DDP_BF16", "1")
    for h in a._hooks:
        h.remove()
    b = GradAllReducer(model, force_overlap=True)
    assert len(b.buckets) == 2 and b.buckets[1].flat.dtype == torch.bfloat16
    x = torch.randn(4, 64)
    model(x).sum().backward()
    ref = [p.grad.clone() for p in model.parameters()]
    model.zero_grad(set_to_none=True)
    model(x).sum().backward()
    b.finish()
    for p, r in zip(model.parameters(), ref):                                # 16-bit buckets hand back fp32 gradients, rounded once
        assert p.grad.dtype == torch.float32 and torch.allclose(p.grad, r.bfloat16().float(), atol=0, rtol=0)
    s = b.profile_summary()
    assert s["bucket_dtype"] == "bfloat16" and len(s["bucket_mbytes"]) == 2 and s["steps"] == 0
