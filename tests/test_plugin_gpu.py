"""GPU: the registered plugin's step bodies (`nndetection_amd.ptmodule.RetinaUNetAMDSteps`) driven the
way Lightning drives `RetinaUNetV001AMD` -- raw nnDetection batch dicts {data, target (instance ids), instance_mapping} ->
`training_step` / `validation_step` -> backward -> `on_after_backward` -> optimizer -- on the box, where neither nnDetection nor
Lightning exists (`StandaloneRetinaUNetV001AMD` carries the same mixin; tests/test_plugin_boundary.py asserts on CPU that the
registered class uses it). Replaces nndet/ptmodule/retinaunet/base.py:135-181 and the precision wiring of scripts/train.py:265-289.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _module(plan_name="tiny", **trainer_over):
    from nndetection_amd.plans import get_plan, MODEL_CFG_V001, TRAINER_CFG_V001
    from nndetection_amd.ptmodule import StandaloneRetinaUNetV001AMD
    p = get_plan(plan_name)
    plan = {"architecture": p["arch"], "anchors": p["anchors"], "patch_size": p["patch_size"], "batch_size": p["batch_size"]}
    torch.manual_seed(0)
    mod = StandaloneRetinaUNetV001AMD(copy.deepcopy(MODEL_CFG_V001), dict(TRAINER_CFG_V001, **trainer_over), plan).cuda()
    return mod, p


def _batch(p, seed=0, batch=None, empty_image=None):
    """An nnDetection training batch: image fp32 [B,1,D,H,W], instance-id volume [B,1,D,H,W] (float, like the data loader's),
    one {instance id: class} mapping per image (it may hold ids that are not in the patch)."""
    P, B = p["patch_size"], batch or p["batch_size"]
    g = torch.Generator().manual_seed(seed)
    data = torch.randn(B, 1, *P, generator=g)
    rng = np.random.default_rng(seed + 1)
    tgt = np.zeros((B, 1, *P), np.float32)
    maps = []
    for b in range(B):
        m = {}
        if b != empty_image:
            for i in range(1, 3 if b % 2 == 0 else 2):
                c = rng.uniform(0.25, 0.75, 3) * np.asarray(P); s = rng.uniform(4, 9, 3)
                lo = np.maximum(c - s / 2, 0).astype(int); hi = np.minimum(c + s / 2, np.asarray(P)).astype(int) + 1
                tgt[b, 0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = i
                m[i] = 0
        m[7] = 0                                       # an instance of the case that lies outside this patch
        maps.append(m)
    return {"data": data.cuda(), "target": torch.from_numpy(tgt).cuda(), "instance_mapping": maps}


def test_training_step_equals_direct_route_fp32(monkeypatch):
    """fp32 batch, no autocast, precision 32: `training_step` (deferred targets, lazy loss scalars) gives exactly the losses and
    gradients of prepare_targets -> BaseRetinaNet.train_step on the same batch; the extra dict entries behave like floats."""
    from nndetection_amd.core.targets import prepare_targets
    from nndetection_amd.ptmodule import LazyFloat
    from tests.gpu_util import det_randperm
    monkeypatch.setattr(torch, "randperm", det_randperm)
    mod, p = _module(precision=32)
    batch = _batch(p)
    out = mod.training_step(batch, 0)
    assert mod.amd_last_dtype == torch.float32
    assert set(out) == {"loss", "reg", "cls", "seg_ce", "seg_dice"} and out["loss"].requires_grad
    assert all(isinstance(out[k], LazyFloat) for k in ("reg", "cls", "seg_ce", "seg_dice"))
    out["loss"].backward()
    mod.on_after_backward()
    g1 = {n: q.grad.clone() for n, q in mod.model.named_parameters() if q.grad is not None}
    mod.zero_grad(set_to_none=True)
    images, targets = prepare_targets(batch["data"], batch["target"], batch["instance_mapping"])
    losses, _ = mod.model.train_step(images, targets, evaluation=False)
    sum(losses.values()).backward()
    for k, v in losses.items():
        assert float(out[k]) == float(v.detach()), k
    assert abs(np.mean([out["reg"], out["cls"]]) - (float(out["reg"]) + float(out["cls"])) / 2) < 1e-12     # training_epoch_end's np.mean
    assert "%.3f" % float(out["cls"]) == f"{out['cls']:.3f}"
    g0 = {n: q.grad for n, q in mod.model.named_parameters() if q.grad is not None}
    assert set(g0) == set(g1) and len(g0) > 40
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * (float(g0[n].abs().max()) + 1e-12), n


@pytest.mark.parametrize("mode", ["autocast-f16", "autocast-bf16", "precision16-no-autocast", "forced-f32"])
def test_precision_selects_the_low_precision_kernels(mode):
    """scripts/train.py:277-278 `pl.Trainer(precision=16, amp_backend='native')`: training_step runs inside torch.autocast(float16)
    with a GradScaler. The activations the conv kernels see must then BE float16 (not the fp32 batch), parameters / gradients stay
    fp32, and a scaled backward + scaler.step updates the weights with finite values."""
    kw = {"precision": 16}
    if mode == "forced-f32":
        kw["amd_dtype"] = "f32"
    mod, p = _module(**kw)
    want = {"autocast-f16": torch.float16, "autocast-bf16": torch.bfloat16, "precision16-no-autocast": torch.bfloat16,
            "forced-f32": torch.float32}[mode]
    seen = []
    hooks = [mod.model.encoder.register_forward_hook(lambda m, i, o: seen.extend(("enc", t.dtype) for t in o if t is not None)),
             mod.model.decoder.register_forward_hook(lambda m, i, o: seen.extend(("dec", t.dtype) for t in o if t is not None))]
    opt, _ = mod.configure_optimizers()
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, enabled=mode == "autocast-f16")
    first = dict(mod.model.named_parameters())["encoder.stages.0.convs.0.0.conv.weight"]
    w0 = first.detach().clone()
    batch = _batch(p, seed=3)
    assert batch["data"].dtype == torch.float32
    import contextlib
    ctx = torch.autocast("cuda", dtype=want) if mode.startswith("autocast") else contextlib.nullcontext()
    with ctx:
        out = mod.training_step(batch, 0)
    for h in hooks:
        h.remove()
    assert mod.amd_last_dtype == want
    assert seen and all(dt == want for _, dt in seen), seen
    assert out["loss"].dtype == torch.float32 and torch.isfinite(out["loss"])
    scaler.scale(out["loss"]).backward()
    mod.on_after_backward()
    grads = [q.grad for q in mod.model.parameters() if q.grad is not None]
    assert len(grads) > 40 and all(g.dtype == torch.float32 for g in grads)
    scaler.step(opt)
    scaler.update()
    assert all(torch.isfinite(q).all() for q in mod.model.parameters())
    if mode == "autocast-f16":
        assert scaler.get_scale() == 1024.0                      # no inf / nan was found: the step was not skipped
    assert not torch.equal(w0, first.detach())


def test_low_precision_step_tracks_fp32_step():
    """The same batch through the plugin in fp32, bf16 and fp16 (autocast): the four losses of the low-precision routes stay within
    1e-2 (bf16) / 2e-3 (fp16) relative of the fp32 route -- i.e. precision=16 runs the SAME network, not a different one."""
    from tests.gpu_util import det_randperm
    orig = torch.randperm
    torch.randperm = det_randperm
    try:
        res = {}
        for name, dt in (("f32", None), ("bf16", torch.bfloat16), ("f16", torch.float16)):
            mod, p = _module(precision=32 if dt is None else 16)
            batch = _batch(p, seed=5)
            import contextlib
            with (torch.autocast("cuda", dtype=dt) if dt is not None else contextlib.nullcontext()):
                out = mod.training_step(batch, 0)
            res[name] = {k: float(v) for k, v in out.items() if k != "loss"}
    finally:
        torch.randperm = orig
    for name, tol in (("bf16", 1e-2), ("f16", 2e-3)):
        for k, v in res["f32"].items():
            assert abs(res[name][k] - v) <= tol * max(1.0, abs(v)), (name, k, res[name][k], v)


def test_validation_step_and_image_without_instances():
    mod, p = _module(precision=32)
    batch = _batch(p, seed=2, empty_image=1)
    out = mod.validation_step(batch, 0)
    assert set(out) == {"loss", "reg", "cls", "seg_ce", "seg_dice"} and all(isinstance(v, float) for v in out.values())
    (prediction, targets), = mod.evaluated
    assert set(prediction) == {"pred_boxes", "pred_scores", "pred_labels", "pred_seg"}
    assert len(prediction["pred_boxes"]) == batch["data"].shape[0] and prediction["pred_seg"].shape[1] == 2
    assert targets["target_boxes"][1].shape == (0, 6) and targets["target_boxes"][0].shape[0] >= 1
    tr = mod.training_step(batch, 1)                            # the deferred route with an image without objects
    tr["loss"].backward()
    assert torch.isfinite(tr["loss"]) and float(tr["cls"]) > 0


def test_conv_block_honours_autocast():
    """B2: a conv block called with an fp32 activation inside torch.autocast computes in the autocast dtype, like
    nn.Conv3d under autocast does in the reference (nndet/arch/conv.py:54-143); outside autocast fp32 stays fp32."""
    from nndetection_amd.arch.conv import ConvInstanceRelu
    torch.manual_seed(1)
    m = ConvInstanceRelu(3, 32, 64, 3, stride=1, padding=1).cuda()
    x = torch.randn(2, 32, 8, 8, 8, device="cuda", requires_grad=True)
    assert m(x).dtype == torch.float32
    for dt in (torch.float16, torch.bfloat16):
        with torch.autocast("cuda", dtype=dt):
            y = m(x)
        assert y.dtype == dt
        y2 = m(x.detach().to(dt))
        assert torch.equal(y, y2)
        y.float().sum().backward()
        assert x.grad is not None and x.grad.dtype == torch.float32 and torch.isfinite(x.grad).all()
        assert m.conv.weight.grad.dtype == torch.float32
        x.grad = None
        m.zero_grad(set_to_none=True)
