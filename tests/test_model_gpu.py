"""GPU: the HIP RetinaUNet against the CPU oracle and the reference golden on the `tiny` plan.
fp32 kernels: losses 1e-4 (north_star tolerance), every parameter gradient norm 1e-3 relative, detections
(box coords, scores: 1e-4 ABSOLUTE; class ids exact). bf16 kernels: bounds stated in tests/test_parity_full_gpu.py."""
import os

import numpy as np
import pytest
import torch

from oracle.detweights import fill_state
from oracle.retina_torch import OracleRetinaUNet
from nndetection_amd.plans import get_plan, MODEL_CFG_V001
from tests.gpu_util import det_randperm

pytestmark = pytest.mark.gpu


def _load(golden_dir):
    gn = np.load(os.path.join(golden_dir, "net_tiny_golden.npz"))
    plan = get_plan("tiny")
    B = plan["batch_size"]
    tg = {"target_boxes": [torch.from_numpy(gn[f"gt_boxes_{i}"]) for i in range(B)],
          "target_classes": [torch.from_numpy(gn[f"gt_classes_{i}"]) for i in range(B)],
          "target_seg": torch.from_numpy(gn["target_seg"].astype(np.float32))}
    return gn, plan, tg


def _hip_model(plan, ora):
    from nndetection_amd.ptmodule import build_model
    net = build_model(plan)
    net.load_state_dict(ora.state_dict())      # strict: identical keys / shapes as the reference
    return net.cuda()


def _cuda_targets(tg):
    return {"target_boxes": [t.cuda() for t in tg["target_boxes"]], "target_classes": [t.cuda() for t in tg["target_classes"]],
            "target_seg": tg["target_seg"].cuda()}


def test_tiny_fp32_vs_oracle_and_golden(golden_dir, monkeypatch):
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    x = torch.from_numpy(gn["x"])
    losses, pred = net.train_step(x.cuda(), _cuda_targets(tg), evaluation=True)
    for k in ("reg", "cls", "seg_ce", "seg_dice"):
        assert abs(losses[k].item() - float(gn[f"loss_{k}"])) < 1e-4, (k, losses[k].item(), float(gn[f"loss_{k}"]))
    # the training-only path (fused segmentation head, no prediction) must give the same losses as the golden as well
    losses_t, pred_t = net.train_step(x.cuda(), _cuda_targets(tg), evaluation=False)
    assert pred_t is None
    for k in ("reg", "cls", "seg_ce", "seg_dice"):
        assert abs(losses_t[k].item() - float(gn[f"loss_{k}"])) < 1e-4, (k, losses_t[k].item(), float(gn[f"loss_{k}"]))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    norms = {k: (p.grad.norm().item() if p.grad is not None else -1.0) for k, p in net.named_parameters()}
    bad = {}
    for k, ref in zip(gn["grad_names"], gn["grad_norms"]):
        k, ref = str(k), float(ref)
        if k.startswith("decoder.out.P0") is False and ref < 0:
            continue                                  # parameter without gradient in the reference (none on `tiny`)
        if abs(norms[k] - ref) > 1e-3 * max(1e-3, abs(ref)):
            bad[k] = (norms[k], ref)
    assert not bad, bad
    for k in ("encoder.stages.0.convs.0.0.conv.weight", "head.regressor.conv_out.conv.bias",
              "decoder.up.P1.conv.weight", "segmenter.conv_out.conv.weight"):
        got = dict(net.named_parameters())[k].grad.detach().cpu().reshape(-1)[:512].numpy()
        ref = gn["grad::" + k]
        assert np.abs(got - ref).max() <= 1e-3 * max(1e-6, np.abs(ref).max()), k
    for b in range(plan["batch_size"]):
        assert pred["pred_boxes"][b].shape == gn[f"det_boxes_{b}"].shape
        from tests.test_parity_full_gpu import assert_detections_match
        assert_detections_match(pred["pred_boxes"][b].cpu().numpy(), pred["pred_scores"][b].cpu().numpy(), pred["pred_labels"][b].cpu().numpy(),
                                gn[f"det_boxes_{b}"], gn[f"det_scores_{b}"], gn[f"det_labels_{b}"], f"tiny image {b}")
    assert abs(pred["pred_seg"].double().sum().item() - float(gn["pred_seg_sum"])) < 1e-3 * float(gn["pred_seg_sum"])


def test_tiny_bf16_close_to_fp32(golden_dir, monkeypatch):
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    x = torch.from_numpy(gn["x"]).cuda().to(torch.bfloat16)
    losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
    sum(losses.values()).backward()
    for k in ("reg", "cls", "seg_ce", "seg_dice"):
        ref = float(gn[f"loss_{k}"])
        assert abs(losses[k].item() - ref) < 0.05 * max(1.0, abs(ref)), (k, losses[k].item(), ref)
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


def test_inference_step_and_state_dict_roundtrip(golden_dir):
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"])
    out = net.inference_step(x.cuda())
    ref = ora.inference_step(x)
    for b in range(x.shape[0]):
        from tests.test_parity_full_gpu import assert_detections_match
        rb, rs = np.asarray(ref["pred_boxes"][b]), np.asarray(ref["pred_scores"][b])
        assert_detections_match(out["pred_boxes"][b].cpu().numpy(), out["pred_scores"][b].cpu().numpy(), out["pred_labels"][b].cpu().numpy(),
                                rb, rs, np.asarray(ref["pred_labels"][b]), f"inference image {b}")
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ora.load_state_dict(sd)                        # keys / shapes load back into the reference-shaped oracle


@pytest.mark.parametrize("sync_free", [False, True], ids=["compact", "syncfree"])
def test_no_gt_image_and_optimizer_step(sync_free, monkeypatch):
    """A batch without any GT (all anchors background, no 'reg' loss: comb.py:397-401) and one SGD step. The compact loss path
    reproduces the reference's dict (no "reg" key, regressor gradients None = the DDP zero-fill case); the default
    sync-free path cannot know the count on the host: "reg" is an exact 0 and the regressor gradients are zeros."""
    from nndetection_amd.ptmodule import build_model, configure_optimizer
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    monkeypatch.setattr(DetectionHeadHNMNative, "sync_free", sync_free)
    plan = get_plan("tiny")
    torch.manual_seed(0)
    net = build_model(plan).cuda()
    opt, sched = configure_optimizer(net)
    x = torch.randn(2, 1, *plan["patch_size"], device="cuda")
    tg = {"target_boxes": [torch.zeros(0, 6, device="cuda")] * 2, "target_classes": [torch.zeros(0, device="cuda")] * 2,
          "target_seg": torch.zeros(2, *plan["patch_size"], device="cuda")}
    losses, _ = net.train_step(x, tg, evaluation=False)
    if sync_free:
        assert set(losses) == {"reg", "cls", "seg_ce", "seg_dice"} and float(losses["reg"].detach()) == 0.0
    else:
        assert "reg" not in losses and set(losses) == {"cls", "seg_ce", "seg_dice"}
    sum(losses.values()).backward()
    g = net.head.regressor.conv_out.conv.weight.grad
    assert (g is not None and not g.any()) if sync_free else g is None
    before = net.encoder.stages[0].convs[0][0].conv.weight.detach().clone()
    opt.step(); sched.step()
    assert not torch.equal(before, net.encoder.stages[0].convs[0][0].conv.weight)


def test_head_side_streams_match_sequential(golden_dir):
    """The (classifier, regressor) x level branches on side streams (default) == the sequential order: identical losses
    (the same kernels on the same data), gradients up to the summation order of the shared head weights."""
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda()
    res = {}
    old = DetectionHeadHNMNative.multi_stream
    try:
        for mode in (True, False, True):
            DetectionHeadHNMNative.multi_stream = mode
            net.zero_grad(set_to_none=True)
            torch.manual_seed(5)                       # the sampler's randperm
            losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(({k: float(v.detach()) for k, v in losses.items()},
                                             {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    finally:
        DetectionHeadHNMNative.multi_stream = old
    (l_ms, g_ms), (l_ms2, g_ms2) = res[True]
    (l_seq, g_seq), = res[False]
    assert l_ms == l_seq == l_ms2, (l_ms, l_seq, l_ms2)
    assert set(g_ms) == set(g_seq)
    for n in g_seq:
        d = float((g_ms[n] - g_seq[n]).abs().max()), float((g_ms2[n] - g_seq[n]).abs().max())
        scale = float(g_seq[n].abs().max()) + 1e-12
        assert max(d) <= 2e-5 * scale, (n, d, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_head_gather_matches_per_level_flatten(golden_dir, dtype):
    """One gather launch per head branch (csrc/headio.hip: flatten + Scale + cat of all levels, forward and backward) == the
    reference's per-level permute / contiguous / view / Scale / cat chain in torch: identical predictions and losses (the same
    values are copied), gradients equal up to the order of the d(scale) sum."""
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    with torch.no_grad():
        for i, sc in enumerate(net.head.regressor.scales):
            sc.scale.fill_(1.0 + 0.25 * i)              # the golden weights leave every Scale at 1
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    res = {}
    old, old_items = DetectionHeadHNMNative.gather_levels, DetectionHeadHNMNative.items_levels
    DetectionHeadHNMNative.items_levels = False        # per-level convolutions on both sides: this test is about the gather alone
    try:
        for mode in (True, False):
            DetectionHeadHNMNative.gather_levels = mode
            net.zero_grad(set_to_none=True)
            torch.manual_seed(5)
            with torch.no_grad():
                pred, _, _ = net(x)
            losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            res[mode] = (pred, {k: float(v.detach()) for k, v in losses.items()},
                         {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    finally:
        DetectionHeadHNMNative.gather_levels, DetectionHeadHNMNative.items_levels = old, old_items
    (p1, l1, g1), (p0, l0, g0) = res[True], res[False]
    assert p1["box_logits"].dtype == torch.float32 and p1["box_logits"].shape == p0["box_logits"].shape
    assert torch.equal(p1["box_logits"], p0["box_logits"]) and torch.equal(p1["box_deltas"], p0["box_deltas"])
    assert l1 == l0, (l1, l0)
    assert set(g1) == set(g0) and any("scales" in n for n in g1)
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-12
        tol = 2e-5 if dtype == torch.float32 else 2e-3      # bf16: dY of conv_out is rounded once (gather) instead of after the Scale mul
        assert float((g1[n] - g0[n]).abs().max()) <= tol * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)


def test_optimizer_steps_track_the_oracle(golden_dir, monkeypatch):
    """TRAINING parity, not single-step parity: SGD(nesterov) steps on the golden batch. Reference arithmetic = the torch oracle driven
    by torch.optim.SGD (what the reference configures, nndet/ptmodule/retinaunet/base.py:300-336); here = the HIP model driven by
    nndetection_amd.optim.SGDNesterov (torch._fused_sgd_). The losses of steps 0..2 agree to 2e-4 (measured 1e-7 .. 3e-6; from step 3
    on a flipped hard-negative pick makes the two runs drift apart like any two fp32 runs) and the weights after 3 steps to 1e-4
    relative -- which they only do if every kernel sees the UPDATED parameters: the convolutions read packed copies cached per
    parameter version, and fused optimizer kernels do not advance the version counters (round 3: stale copies had gone unnoticed by
    the single-step parity tests; with stale copies step 1 already deviates by 9 %)."""
    from nndetection_amd.optim import SGDNesterov
    from nndetection_amd.ptmodule import get_params_no_wd_on_norm
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    x = torch.from_numpy(gn["x"])
    lr = 0.02
    opt_o = torch.optim.SGD(get_params_no_wd_on_norm(ora, 3e-5), lr, momentum=0.9, nesterov=True)
    opt_h = SGDNesterov(get_params_no_wd_on_norm(net, 3e-5), lr, momentum=0.9, nesterov=True)
    w0 = {n: p.detach().clone() for n, p in ora.named_parameters()}
    devs = []
    for it in range(3):
        lo, _ = ora.train_step(x, tg, evaluation=False)
        opt_o.zero_grad(); sum(lo.values()).backward(); opt_o.step()
        lh, _ = net.train_step(x.cuda(), _cuda_targets(tg), evaluation=False)
        opt_h.zero_grad(); sum(lh.values()).backward()
        v_before = {n: p._version for n, p in net.named_parameters() if p.grad is not None}
        opt_h.step()
        assert all(p._version > v_before[n] for n, p in net.named_parameters() if n in v_before), "parameter versions must advance"
        devs.append(max(abs(float(lh[k].detach()) - float(lo[k].detach())) / max(1.0, abs(float(lo[k].detach()))) for k in lo))
    assert max(devs) <= 2e-4, devs
    moved, worst = 0, (0.0, "")
    for n, p in net.named_parameters():
        ref = dict(ora.named_parameters())[n].detach()
        step = float((ref - w0[n]).abs().max())             # how far the oracle moved this parameter in 3 steps
        moved += step > 1e-5
        if step > 0:                                        # deviation relative to the distance travelled
            worst = max(worst, (float((p.detach().cpu() - ref).abs().max()) / step, n))
    print("largest weight deviation / distance travelled:", worst)
    assert worst[0] <= 5e-2, worst              # (measured 1.1e-2: the atomically reduced gradients differ in the last bits run to run)
    assert moved > 40                                     # the comparison is not vacuous: the parameters did change

    # A foreign FUSED optimizer (torch.optim.SGD(fused=True): what ptmodule.amd_fuse_sgd switches the reference's optimizer to) does
    # not advance the counters at all: a training-mode forward pass re-packs unconditionally, train() <-> eval() drops the caches.
    # Reference = a FRESH model (empty caches) loaded with the trained weights.
    def fresh():
        m = _hip_model(plan, ora)
        m.load_state_dict(net.state_dict())
        return m

    opt_f = torch.optim.SGD(get_params_no_wd_on_norm(net, 3e-5), lr, momentum=0.9, nesterov=True, fused=True)
    for it in range(2):
        lh, _ = net.train_step(x.cuda(), _cuda_targets(tg), evaluation=False)
        opt_f.zero_grad(); sum(lh.values()).backward(); opt_f.step()
    l2, _ = net.train_step(x.cuda(), _cuda_targets(tg), evaluation=False)
    l3, _ = fresh().train_step(x.cuda(), _cuda_targets(tg), evaluation=False)
    for k in l3:
        assert abs(float(l2[k].detach()) - float(l3[k].detach())) <= 1e-5 * max(1.0, abs(float(l3[k].detach()))), (k, float(l2[k]), float(l3[k]))
    net.eval()
    m = fresh().eval()
    with torch.no_grad():
        p1, p2 = net.inference_step(x.cuda()), m.inference_step(x.cuda())
    assert len(p1["pred_boxes"]) == len(p2["pred_boxes"])
    for a_, b_ in zip(p1["pred_boxes"], p2["pred_boxes"]):
        assert a_.shape == b_.shape and float((a_ - b_).abs().max()) <= 1e-4 if a_.numel() else True
    for a_, b_ in zip(p1["pred_scores"], p2["pred_scores"]):
        assert float((a_ - b_).abs().max()) <= 1e-5 if a_.numel() else True


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_sparse_backward_of_the_head_output_convolutions(golden_dir, dtype, monkeypatch):
    """csrc/sparse_out.hip: the gradient w.r.t. box_logits / box_deltas is zero except at the <= 170 sampled anchors; with the hints of
    `_DetLossFn.backward` the gather backward and the two output convolutions' data / weight gradients touch only those rows. Same
    losses, same gradients for EVERY parameter (the head trunks, decoder and encoder see the data gradient) as the dense backward --
    fp32: summation order only; 16-bit: the dense route rounds the output gradient to 16 bits, the sparse one does not."""
    import nndetection_amd.arch.heads as H
    from nndetection_amd import _lib as L
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    calls = {"n": 0}
    real = L.call

    def counting(name, *a):
        if name in ("nndet_head_out_sparse_scatter", "nndet_conv_out_sparse_backward", "nndet_conv_out_sparse_forward"):
            calls["n"] += 1
        return real(name, *a)

    monkeypatch.setattr(L, "call", counting)
    tol = 2e-5 if dtype == torch.float32 else (3e-2 if dtype == torch.bfloat16 else 4e-3)
    ltol = 1e-6 if dtype == torch.float32 else (2e-3 if dtype == torch.bfloat16 else 3e-4)
    # The sparse regressor forward computes the sampled deltas in fp32 from the 16-bit trunk output, the dense route rounds them to
    # 16 bits first. GIoU's min / max have KINKS: when a decoded coordinate of a sampled positive ties with the ground truth's (with
    # scale 1.0 on level 0 one positive of the golden batch decodes to z2 = 13.48468 against a ground-truth z2 of 13.48468; with 0.9 and
    # the activations of k_ig3s another one does), the two roundings legitimately pick different branches of the gradient (25 % on
    # regressor.conv_out.weight, nothing elsewhere). A tie is a coincidence of one scale value: the full-sparse mode is run with two level-0
    # scales and must agree with the dense route for at least one of them; backward-only sparsity (same forward, no tie possible) for both.
    full_sparse_ok = []
    for base in (0.9, 0.95):
        with torch.no_grad():
            for i, sc in enumerate(net.head.regressor.scales):
                sc.scale.fill_(base + 0.25 * i)
        res = {}
        # (sparse backward, sparse regressor forward): everything / backward only / the reference's dense route
        for mode, expect in (((True, True), 4), ((True, False), 4), ((False, False), 0)):
            monkeypatch.setattr(H, "SPARSE_OUT", mode[0])
            monkeypatch.setattr(H, "SPARSE_REG", mode[1])
            net.zero_grad(set_to_none=True)
            calls["n"] = 0
            losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
            (sum(losses.values()) * (256.0 if dtype == torch.float16 else 1.0)).backward()
            torch.cuda.synchronize()
            # (T, T): cls scatter + cls conv backward + reg sparse forward + reg conv backward; (T, F): 2 x (scatter + conv backward)
            assert calls["n"] == expect, (mode, calls)
            res[mode] = ({k: float(v.detach()) for k, v in losses.items()},
                         {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None})
        dense = res[(False, False)]
        for mode in ((True, True), (True, False)):
            got = res[mode]
            for k, v in dense[0].items():
                assert abs(got[0][k] - v) <= ltol * max(1.0, abs(v)), (base, mode, k, got[0][k], v)
            assert set(got[1]) == set(dense[1]) and any("regressor.conv_out" in n for n in got[1])
            bad = {n: float((got[1][n] - g0).abs().max()) / (float(g0.abs().max()) + 1e-12) for n, g0 in dense[1].items()
                   if float((got[1][n] - g0).abs().max()) > tol * (float(g0.abs().max()) + 1e-12)}
            if mode == (True, True):
                full_sparse_ok.append((base, bad))
            else:
                assert not bad, (base, mode, bad)
        assert res[(True, False)][0] == dense[0]            # backward-only sparsity leaves the forward pass untouched
    assert any(not bad for _, bad in full_sparse_ok), full_sparse_ok
    # evaluation (a prediction is asked for) always takes the dense forward route
    monkeypatch.setattr(H, "SPARSE_OUT", True); monkeypatch.setattr(H, "SPARSE_REG", True)
    calls["n"] = 0
    with torch.no_grad():
        _, pred = net.train_step(x, _cuda_targets(tg), evaluation=True)
    assert calls["n"] == 0 and len(pred["pred_boxes"]) == x.shape[0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_segmentation_branch_fusion_in_train_step(golden_dir, dtype, monkeypatch):
    """A training step without prediction leaves decoder.out.P0 to the segmentation branch (`UFPNModular.defer_out0`,
    `_SegBranchFn`, csrc/segbranch.hip: one composed 32 -> 1 convolution + loss). Same losses and the same gradient for EVERY
    parameter as the two-layer route (16-bit: that route rounds the 32-channel map and the logits, the fused one does not); the
    fp32 step and a step that asks for a prediction keep the two layers."""
    from nndetection_amd.arch import segmenter as S
    from nndetection_amd import _lib as L
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    res = {}
    # (whole branch incl. the level-0 lateral and the last top-down step, nndet_segbranch_forward_up) / (... without the top-down step,
    # nndet_segbranch_forward2) / (output conv + head only) / the separate layers
    for mode in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
        monkeypatch.setattr(S, "SEG_BRANCH", mode[0]); monkeypatch.setattr(S, "SEG_LATERAL", mode[1]); monkeypatch.setattr(S, "SEG_UP", mode[2])
        net.zero_grad(set_to_none=True)
        calls.clear()
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        (sum(losses.values()) * (256.0 if dtype == torch.float16 else 1.0)).backward()
        torch.cuda.synchronize()
        lp = dtype != torch.float32
        # (this model's head reads decoder level 1, so its last top-down step cannot be absorbed: the first two modes coincide here;
        # tests/test_parity_full_gpu.py::test_luna160_absorbed_top_down_step runs the route on the architecture that allows it)
        up_ok = net._seg_up_ok(x)
        assert not up_ok
        assert ("nndet_segbranch_forward_up" in calls) == (mode == (True, True, True) and lp and up_ok), (mode, dtype)
        assert ("nndet_segbranch_forward2" in calls) == (mode[:2] == (True, True) and lp and not (mode[2] and up_ok)), (mode, dtype)
        assert ("nndet_segbranch_forward" in calls) == (mode == (True, False, False) and lp), (mode, dtype)
        assert ("nndet_segbranch_backward" in calls) == (mode[0] and lp)
        res[mode] = ({k: float(v.detach()) for k, v in losses.items()},
                     {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None})
    res[False] = res[(False, False, False)]
    # (fp32: the same kernels both times, only the atomics' summation order differs)
    tol = 2e-5 if dtype == torch.float32 else (3e-2 if dtype == torch.bfloat16 else 4e-3)
    ltol = 1e-6 if dtype == torch.float32 else (2e-3 if dtype == torch.bfloat16 else 3e-4)
    for mode in ((True, True, True), (True, True, False), (True, False, False)):
        for k, v in res[False][0].items():
            assert abs(res[mode][0][k] - v) <= ltol * max(1.0, abs(v)), (mode, k, res[mode][0][k], v)
        assert set(res[mode][1]) == set(res[False][1]) and "decoder.out.P0.0.conv.weight" in res[mode][1] \
            and "decoder.lateral.P0.0.conv.weight" in res[mode][1] and "decoder.up.P1.conv.weight" in res[mode][1] \
            and "decoder.up.P1.conv.bias" in res[mode][1]
        for n, g0 in res[False][1].items():
            d = float((res[mode][1][n] - g0).abs().max())
            assert d <= tol * (float(g0.abs().max()) + 1e-12) + 1e-7, (mode, n, d, float(g0.abs().max()))
    monkeypatch.setattr(S, "SEG_BRANCH", True); monkeypatch.setattr(S, "SEG_LATERAL", True); monkeypatch.setattr(S, "SEG_UP", True)
    calls.clear()
    with torch.no_grad():
        _, pred = net.train_step(x, _cuda_targets(tg), evaluation=True)
    assert "nndet_segbranch_forward" not in calls and pred["pred_seg"].shape[1] == 2
    assert net.decoder.defer_out0 is False and net.decoder.absorb_lat0 is False and net.decoder.absorb_up0 is False


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_inference_step_segmentation_from_the_composed_convolution(golden_dir, dtype, monkeypatch):
    """inference_step in 16 bits takes the segmentation probabilities from the training step's composed convolution (decoder level 0 and
    the logits never exist; `FgBgSegmenter.logit_difference`). Same detections as with the separate layers; probabilities no further from
    the fp32 evaluation than those of the separate 16-bit layers (which round the 32-channel map and the logits) allow; fp32 keeps the
    separate layers; the composed parameter tensors are cached between calls and rebuilt when a parameter changes."""
    from nndetection_amd.core.retina import BaseRetinaNet
    from nndetection_amd import _lib as L
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora).eval()
    x = torch.from_numpy(gn["x"]).cuda()
    ref32 = net.inference_step(x)["pred_seg"].double()
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(BaseRetinaNet, "seg_infer_fused", fused)
        calls.clear()
        out[fused] = net.inference_step(x.to(dtype))
        hit = any(c.startswith("nndet_segbranch_forward") for c in calls)
        assert hit == (fused and dtype != torch.float32), (fused, dtype, hit)
        assert out[fused]["pred_seg"].shape == ref32.shape and out[fused]["pred_seg"].dtype == torch.float32
        assert net.decoder.defer_out0 is False and net._seg_infer is False
    for b in range(x.shape[0]):
        assert torch.equal(out[True]["pred_boxes"][b], out[False]["pred_boxes"][b]) and torch.equal(out[True]["pred_scores"][b], out[False]["pred_scores"][b])
    e_f = float((out[True]["pred_seg"].double() - ref32).abs().max())
    e_s = float((out[False]["pred_seg"].double() - ref32).abs().max())
    assert float((out[True]["pred_seg"].sum(1) - 1).abs().max()) < 1e-6
    if dtype == torch.float32:
        assert e_f == 0.0 and e_s == 0.0
        return
    assert e_f <= 1.5 * e_s + 1e-4, (e_f, e_s)
    # cache: a second call builds nothing; a changed parameter (in place, no version bump: the fused optimizers' route) does
    from nndetection_amd.arch import segmenter as S
    from nndetection_amd.arch.conv import bump_param_generation
    monkeypatch.setattr(BaseRetinaNet, "seg_infer_fused", True)
    key0 = S._compose_cache.get("key")
    again = net.inference_step(x.to(dtype))
    assert torch.equal(again["pred_seg"], out[True]["pred_seg"]) and S._compose_cache.get("key") == key0
    with torch.no_grad():
        net.segmenter.conv_out.conv.bias.add_(torch.tensor([0.0, 1.0], device="cuda"))
    bump_param_generation()
    moved = net.inference_step(x.to(dtype))["pred_seg"]
    assert S._compose_cache.get("key") != key0 or not net._seg_up_ok(x.to(dtype))
    assert float((moved[:, 1] - out[True]["pred_seg"][:, 1]).min()) > 0.0          # every foreground probability went up


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_early_consumer_of_the_stage0_output(golden_dir, dtype, monkeypatch):
    """arch/conv.py EARLY_CONSUMER: the first convolution of encoder stage 1 reads the PRE-norm output of stage 0 (norm + ReLU applied
    while staging, `in_affine`) while the normalised tensor is written on an auxiliary stream. Same arithmetic on the same values: the
    losses and every parameter gradient agree with the plain route to the noise of the statistics' atomics, the stage-0 output is
    the same tensor bit for bit, the tag does not leak out of the encoder, and the affine launch really is the one that runs."""
    from nndetection_amd.arch import conv as C
    from nndetection_amd import _lib as L
    monkeypatch.setenv("NNDET_IG3S", "0")             # like with like: a consumer that applies the norm on load cannot stage by LDS-DMA (k_ig3s)
    monkeypatch.setattr(C, "NORM_INPUT_FUSE", False)  # (round 6: the route that replaced this one, tested below)
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    res = {}
    for on in (True, False, True):
        monkeypatch.setattr(C, "EARLY_CONSUMER", on)
        net.zero_grad(set_to_none=True)
        calls.clear()
        with torch.no_grad():
            feats = net.encoder(x)
        torch.cuda.synchronize()
        assert ("nndet_affine_apply" in calls) == on
        assert all(not hasattr(f, "_nndet_pre") for f in feats)
        f0 = feats[0].detach().float().cpu().clone()
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        (sum(losses.values()) * (256.0 if dtype == torch.float16 else 1.0)).backward()
        torch.cuda.synchronize()
        res.setdefault(on, []).append((f0, {k: float(v.detach()) for k, v in losses.items()},
                                       {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}))
    tol = 2e-5 if dtype == torch.float32 else (3e-2 if dtype == torch.bfloat16 else 4e-3)
    ltol = 1e-6 if dtype == torch.float32 else (2e-3 if dtype == torch.bfloat16 else 3e-4)
    ref = res[False][0]
    for f0, ls, gr in res[True]:
        assert torch.equal(f0, ref[0])
        for k, v in ref[1].items():
            assert abs(ls[k] - v) <= ltol * max(1.0, abs(v)), (k, ls[k], v)
        assert set(gr) == set(ref[2])
        for n, g0 in ref[2].items():
            d = float((gr[n] - g0).abs().max())
            assert d <= tol * (float(g0.abs().max()) + 1e-12) + 1e-7, (n, d, float(g0.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_stage1_convolution_writes_the_normalised_stage0_output(golden_dir, dtype, monkeypatch):
    """arch/conv.py NORM_INPUT_FUSE (round 6): stage 0's last block only computes its coefficient table; the first (stride-2) convolution
    of stage 1 reads the pre-norm tensor and writes the normalised one on the way (nndet_conv3d_forward_norm_input, k_ig3s<.., PRE>),
    no nndet_norm_apply pass for that block. The stage-0 output and every deeper feature map are the same tensors bit for bit, losses and
    gradients agree to the noise of the statistics' atomics, the tag does not leak, and with the fused kernel switched off
    (NNDET_IG3S=0) the consumer falls back to the plain materialising pass with the same values."""
    from nndetection_amd.arch import conv as C
    from nndetection_amd import _lib as L
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    res = {}
    for mode in ("fused", "off", "fused", "fallback"):
        monkeypatch.setattr(C, "NORM_INPUT_FUSE", mode != "off")
        monkeypatch.setenv("NNDET_IG3S", "0" if mode == "fallback" else "1")
        net.zero_grad(set_to_none=True)
        calls.clear()
        with torch.no_grad():
            feats = net.encoder(x)
        torch.cuda.synchronize()
        assert ("nndet_conv3d_forward_norm_input" in calls) == (mode == "fused")
        assert ("nndet_affine_apply" in calls) == (mode == "fallback")
        assert all(not hasattr(f, "_nndet_pre") for f in feats)
        fs = [f.detach().float().cpu().clone() for f in feats]
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        (sum(losses.values()) * (256.0 if dtype == torch.float16 else 1.0)).backward()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((fs, {k: float(v.detach()) for k, v in losses.items()},
                                         {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}))
    # the unwritten buffer exists only inside the encoder's loop: a stage called on its own, or one that is observed through a forward
    # hook, takes the plain route and returns a written tensor
    monkeypatch.setattr(C, "NORM_INPUT_FUSE", True)
    monkeypatch.setenv("NNDET_IG3S", "1")
    with torch.no_grad():
        calls.clear()
        alone = net.encoder.stages[0](x)
        torch.cuda.synchronize()
        assert "nndet_conv3d_forward_norm_input" not in calls and not hasattr(alone, "_nndet_pre")
        assert torch.equal(alone.detach().float().cpu(), res["off"][0][0][0])
        seen = []
        h = net.encoder.stages[0].convs[-1].register_forward_hook(lambda m, i, o: seen.append(o.detach().float().cpu().clone()))
        calls.clear()
        feats = net.encoder(x)
        torch.cuda.synchronize()
        h.remove()
        assert "nndet_conv3d_forward_norm_input" not in calls and len(seen) == 1 and torch.equal(seen[0], res["off"][0][0][0])
        calls.clear()
        net.encoder(x)
        assert "nndet_conv3d_forward_norm_input" in calls             # (hook gone: fused again)
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    ltol = 2e-3 if dtype == torch.bfloat16 else 3e-4
    ref = res["off"][0]
    for mode in ("fused", "fallback"):
        for fs, ls, gr in res[mode]:
            assert torch.equal(fs[0], ref[0][0]), mode                 # the normalised stage-0 output itself
            for f, r in zip(fs[1:], ref[0][1:]):                       # (deeper stages: the same kernels on the same values, up to the atomics)
                assert float((f - r).abs().max()) <= (tol if mode == "fused" else 4 * tol) * float(r.abs().max())
            for k, v in ref[1].items():
                assert abs(ls[k] - v) <= ltol * max(1.0, abs(v)), (mode, k, ls[k], v)
            assert set(gr) == set(ref[2])
            if mode == "fallback":            # (NNDET_IG3S=0: k_igemm rounds the stride-2 output its own way, the gradients are another sample)
                continue
            for n, g0 in ref[2].items():
                dd = float((gr[n] - g0).abs().max())
                assert dd <= tol * (float(g0.abs().max()) + 1e-12) + 1e-7, (mode, n, dd, float(g0.abs().max()))


def test_head_levels_are_written_into_the_ragged_batch_buffer(golden_dir, monkeypatch):
    """arch/decoder.py `_ragged_out`: the out convolutions of the levels the detection head reads write into consecutive slices of ONE
    allocation, so `cat_levels` (the head's ragged [rows, C_p] batch) is a view instead of a 45 MB copy -- same values as the
    concatenation of separately allocated outputs, gradients flow to every level."""
    from nndetection_amd.arch import decoder as D
    from nndetection_amd.arch.pyramid import cat_levels
    from nndetection_amd.layout import phys
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda().to(torch.bfloat16)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(D.UFPNModular, "ragged_out", on)
        net.zero_grad(set_to_none=True)
        outs = net.decoder(net.encoder(x))
        torch.cuda.current_stream().wait_event(net.decoder.tail_event) if net.decoder.tail_event is not None else None
        fm = [outs[l] for l in net.decoder_levels]
        t2d, meta = cat_levels(fm)
        assert (t2d.data_ptr() == phys(fm[0])[0].data_ptr()) == on           # a view of the first level's buffer / a fresh copy
        (t2d.float() * torch.linspace(-1, 1, t2d.shape[1], device="cuda")).sum().backward()
        torch.cuda.synchronize()
        res[on] = (t2d.detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in net.named_parameters() if p.grad is not None})
    assert torch.equal(res[True][0], res[False][0])
    assert set(res[True][1]) == set(res[False][1]) and len(res[True][1]) > 20
    for n, g in res[False][1].items():
        assert float((res[True][1][n] - g).abs().max()) <= 2e-2 * float(g.abs().max()) + 1e-7, n      # (same kernels; atomics' order only)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_fused_input_gradient_accumulation(golden_dir, dtype, monkeypatch):
    """Encoder stage outputs feed the next stage and the decoder lateral. With set_fuse_grad_accum the second data gradient is added
    into the first one's buffer (nndet_conv3d_backward_data_acc, or its _normred form that also accumulates the producer's norm-backward
    sums in 16-bit types) instead of autograd adding two tensors: same gradients (fp32: the
    same single rounding of a + b; bf16: one rounding less), and the accumulate entry point really is the one that runs."""
    from nndetection_amd import _lib as L
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    calls = []
    orig_call = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), orig_call(name, *a))[1])
    res = {}
    for mode in (True, False):
        net.encoder.set_fuse_grad_accum(mode)
        net.zero_grad(set_to_none=True)
        calls.clear()
        torch.manual_seed(5)
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        res[mode] = ({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None},
                     calls.count("nndet_conv3d_backward_data_acc") + calls.count("nndet_conv3d_backward_data_acc_normred"))
    (g1, n1), (g0, n0) = res[True], res[False]
    assert n0 == 0 and n1 == net.encoder.num_stages - 1, (n0, n1)
    assert set(g1) == set(g0)
    tol = 2e-5 if dtype == torch.float32 else 3e-2      # bf16: the sum is rounded once instead of twice; dgamma / dbeta are cancelling sums
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-12
        assert float((g1[n] - g0[n]).abs().max()) <= tol * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)


def test_toy64_config0_fp32_vs_reference_golden(golden_dir, monkeypatch):
    """BASELINE.json configs[0] on the GPU (fp32 kernels): losses within the 1e-4 of north_star, every gradient norm 1e-3 relative,
    detections (boxes, scores 1e-4; class ids exact) against what the unmodified reference produced on the CPU."""
    from tests.gpu_util import synth_inputs
    gn = np.load(os.path.join(golden_dir, "net_toy64_golden.npz"))
    plan = get_plan("toy64")
    x, tg = synth_inputs(plan)
    assert abs(float(x.double().sum()) - float(gn["x_checksum"])) < 1e-6
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    losses, pred = net.train_step(x.cuda(), _cuda_targets(tg), evaluation=True)
    for k in ("reg", "cls", "seg_ce", "seg_dice"):
        assert abs(losses[k].item() - float(gn[f"loss_{k}"])) < 1e-4, (k, losses[k].item(), float(gn[f"loss_{k}"]))
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    norms = {k: (p.grad.norm().item() if p.grad is not None else -1.0) for k, p in net.named_parameters()}
    bad = {}
    for k, ref in zip(gn["grad_names"], gn["grad_norms"]):
        k, ref = str(k), float(ref)
        if abs(norms[k] - ref) > 1e-3 * max(1e-3, abs(ref)):
            bad[k] = (norms[k], ref)
    assert not bad, bad
    for b in range(plan["batch_size"]):
        pb, ps, pl = pred["pred_boxes"][b].cpu().numpy(), pred["pred_scores"][b].cpu().numpy(), pred["pred_labels"][b].cpu().numpy()
        from tests.test_parity_full_gpu import assert_detections_match
        assert_detections_match(pb, ps, pl, gn[f"det_boxes_{b}"], gn[f"det_scores_{b}"], gn[f"det_labels_{b}"], f"toy64 image {b}")


def test_ddp_overlap_path_on_one_gpu_with_multistream_head(golden_dir):
    """VERDICT r1 item 8 / ADVICE (medium): GradAllReducer(force_overlap=True) runs the hook -> bucket copy -> gradient-view
    path at world size 1 while the detection head accumulates its parameter gradients on side streams. Two training steps
    with and without it must give the same gradients (up to the summation order of shared head weights)."""
    from nndetection_amd.ddp import GradAllReducer
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    assert DetectionHeadHNMNative.multi_stream, "the multi-stream head is the default"
    gn, plan, tg = _load(golden_dir)
    x = torch.from_numpy(gn["x"]).cuda()
    res = {}
    for mode in ("plain", "overlap"):
        ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
        net = _hip_model(plan, ora)
        ddp = GradAllReducer(net, first_bucket_mb=0.05, bucket_mb=0.2, force_overlap=True) if mode == "overlap" else None
        if ddp is not None:
            assert len(ddp.buckets) >= 3
        from nndetection_amd.ptmodule import configure_optimizer
        opt, sched = configure_optimizer(net)
        grads = []
        for step in range(2):
            torch.manual_seed(5 + step)
            losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
            sum(losses.values()).backward()
            if ddp is not None:
                assert ddp._next >= len(ddp.buckets) - 1          # the buckets were launched from the hooks, during backward
                ddp.finish()
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
            opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
        res[mode] = grads
    for step in range(2):
        a, b = res["plain"][step], res["overlap"][step]
        assert set(a) <= set(b)                                    # overlap mode zero-fills the never-used parameters
        # step 0: identical weights -> identical kernels, only the summation order of atomically reduced gradients differs.
        # step 1: the weights differ in their last bits after step 0 (that summation order), and WHICH negatives the sampler
        # draws from the boundary of its top-k pool reacts to that; a stale / unsynchronised bucket would be off by O(1).
        tol = 2e-5 if step == 0 else 2e-3
        for n in a:
            scale = float(a[n].abs().max()) + 1e-12
            assert float((a[n] - b[n]).abs().max()) <= tol * scale, (step, n)


def test_ddp_in_place_gradients_and_a_batch_without_positives(golden_dir, monkeypatch):
    """VERDICT r4 item 8a / 8b on the real model (forced world-1 overlap path on one GPU). (a) The conv / norm nodes write their parameter
    gradients INTO the reducer's bucket memory (static gradient-pool layout): after backward + finish() almost no gradient had to be copied
    (the two fused first trunk layers, the absorbed segmentation branch and the Scale parameters come from other memory) and every
    p.grad lives inside the flat buffer. (b) On the compact loss route (the reference's: no "reg" key without positives,
    nndet/arch/heads/comb.py:397-401) a batch WITHOUT any ground-truth box leaves the 12 regressor tensors without a gradient: the head
    tells the reducer during the forward pass (_lib.notify_no_grad), so every bucket is still launched from the gradient hooks and the
    regressor gradients come out as zeros. Gradients equal those of a run without the reducer."""
    from nndetection_amd.ddp import GradAllReducer
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    from nndetection_amd import _lib as L
    gn, plan, tg = _load(golden_dir)
    x = torch.from_numpy(gn["x"]).cuda()
    empty = {"target_boxes": [torch.zeros((0, 6)) for _ in tg["target_boxes"]], "target_classes": [torch.zeros((0,)) for _ in tg["target_classes"]],
             "target_seg": torch.zeros_like(tg["target_seg"])}
    monkeypatch.setattr(torch, "randperm", det_randperm)
    res = {}
    for mode in ("plain", "ddp"):
        ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
        net = _hip_model(plan, ora)
        ddp = GradAllReducer(net, first_bucket_mb=0.05, bucket_mb=0.2, force_overlap=True) if mode == "ddp" else None
        out = {}
        try:
            for case, targets, sync_free in (("positives", tg, True), ("no positives, compact route", empty, False)):
                monkeypatch.setattr(DetectionHeadHNMNative, "sync_free", sync_free)
                net.zero_grad(set_to_none=True)
                losses, _ = net.train_step(x, _cuda_targets(targets), evaluation=False)
                assert ("reg" in losses) == (case == "positives")
                sum(losses.values()).backward()
                if ddp is not None:
                    n_before = ddp._next
                    ddp.finish()
                    assert n_before == len(ddp.buckets) and all(ddp.launched_from_hooks), (case, n_before, ddp.launched_from_hooks)
                    lo, hi = ddp._flat_all.data_ptr(), ddp._flat_all.data_ptr() + ddp._flat_all.numel() * 4
                    assert all(lo <= p.grad.data_ptr() < hi for p in net.parameters()), case
                    n_params = sum(len(b.params) for b in ddp.buckets)
                    assert ddp.copied_last <= 24, (case, ddp.copied_last, n_params)       # of 92: everything else was written in place
                    out[case + " copied"] = ddp.copied_last
                torch.cuda.synchronize()
                out[case] = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in net.named_parameters()}
        finally:
            if ddp is not None:
                ddp.close()
        res[mode] = out
    assert L.grad_pool.static is None
    for case in ("positives", "no positives, compact route"):
        for n, g in res["plain"][case].items():
            h = res["ddp"][case][n]
            assert h is not None, n
            if g is None:                                  # no gradient without the reducer = zeros with it
                assert float(h.abs().max()) == 0.0, (case, n)
                continue
            assert float((g - h).abs().max()) <= 2e-5 * (float(g.abs().max()) + 1e-12), (case, n)
    reg_none = [n for n, g in res["plain"]["no positives, compact route"].items() if g is None and n.startswith("head.regressor")]
    assert len(reg_none) == 8 + len(plan["arch"]["decoder_levels"]), reg_none    # 3 + 3 + 2 trunk / output tensors + one Scale per level (12 at luna160)


def test_lean_sgd_matches_torch_sgd_on_gpu():
    """a19: the foreach SGD(nesterov) + LinearWarmupPolyLR pair against torch.optim.SGD + the reference's schedule on the GPU,
    on the real parameter groups of the model (no weight decay on norm parameters), 5 steps with synthetic gradients."""
    from nndetection_amd.ptmodule import build_model, configure_optimizer
    plan = get_plan("tiny")
    torch.manual_seed(0)
    a = build_model(plan).cuda()
    b = build_model(plan).cuda()
    b.load_state_dict(a.state_dict())
    oa, sa = configure_optimizer(a, lean=True)
    ob, sb = configure_optimizer(b, lean=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(5):
        for pa, pb in zip(a.parameters(), b.parameters()):
            if it % 2 == 1 and pa.ndim == 0:
                continue                                           # parameters without gradient in some steps
            gr = torch.randn(pa.shape, device="cuda", generator=g)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step(); sa.step(); oa.zero_grad(set_to_none=True)
        ob.step(); sb.step(); ob.zero_grad(set_to_none=True)
        assert abs(oa.param_groups[0]["lr"] - ob.param_groups[0]["lr"]) < 1e-12
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-7, rtol=1e-6), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_deferred_norm_equals_materialised(golden_dir, monkeypatch, dtype):
    """with deferred normalisation (default) the consumers apply InstanceNorm / GroupNorm + ReLU while staging
    their input (NndetConv.in_affine) and `k_norm_apply` never runs; with NNDET_DEFER_NORM=0 every block materialises its
    activation. Same arithmetic (fmaf, max, one rounding) on both routes -> identical losses; gradients identical up to the
    summation order of the atomically reduced ones."""
    import nndetection_amd.arch.conv as C
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    monkeypatch.setattr(DetectionHeadHNMNative, "items_levels", False)     # the ragged head path has no deferred variant: compare like with like
    monkeypatch.setattr(C, "FUSED_STEM", False)       # (the fused stem block normalises the UNROUNDED conv output: other arithmetic, own test)
    import nndetection_amd.arch.segmenter as S
    monkeypatch.setattr(S, "SEG_LATERAL", False)      # (absorbing the level-0 lateral needs a materialised encoder output: ditto)
    monkeypatch.setenv("NNDET_IG3S", "0")             # (k_ig3s stages by LDS-DMA and has no deferred variant: the same kernel on both routes)
    monkeypatch.setattr(C, "NORM_RED_FUSE", False)    # (the norm-backward sums from k_dgs's epilogue exist on the materialised route only and
    #                                                    associate differently: tests/test_conv_gpu.py::test_norm_backward_sums_from_the_strided_data_gradient)
    gn, plan, tg = _load(golden_dir)
    x = torch.from_numpy(gn["x"]).cuda().to(dtype)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(C, "DEFER_NORM", mode)
        ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
        net = _hip_model(plan, ora)
        monkeypatch.setattr(torch, "randperm", det_randperm)
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        res[mode] = ({k: float(v.detach()) for k, v in losses.items()},
                     {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None})
        # the deferred route really is deferred: the first encoder conv hands on a tagged tensor
        with torch.no_grad():
            y = net.encoder.stages[0].convs[0][0](x)
        assert (C.deferred(y) is not None) == mode
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    for n, g in res[False][1].items():
        d = float((res[True][1][n] - g).abs().max())
        assert d <= 1e-5 * (float(g.abs().max()) + 1e-12), (n, d)


def test_materialize_of_deferred_activation(monkeypatch):
    """A deferred activation handed to a consumer that cannot apply the norm on load (here: plain torch code) is materialised by
    `materialize`; values and gradients equal the non-deferred block."""
    import nndetection_amd.arch.conv as C
    from nndetection_amd.arch.conv import ConvInstanceRelu
    monkeypatch.setattr(C, "DEFER_NORM", True)
    torch.manual_seed(0)
    m = ConvInstanceRelu(3, 32, 32, 3, padding=1).cuda()
    x = torch.randn(2, 32, 9, 10, 12, device="cuda")
    outs = {}
    for defer in (False, True):
        m.defer_output = defer
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        assert (C.deferred(y) is not None) == defer
        y = C.materialize(y)
        (y * torch.arange(y.numel(), device="cuda").view_as(y).float().cos()).sum().backward()
        outs[defer] = (y.detach().clone(), xi.grad.clone(), m.conv.weight.grad.clone(), m.norm.weight.grad.clone())
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)


def _rand_loss_inputs(M, B, C, seed):
    """logits / deltas / labels / matched boxes / anchors of B images with M anchors each + padded index lists"""
    g = torch.Generator().manual_seed(seed)
    anchors = torch.rand(M, 3, generator=g) * 40
    size = torch.rand(M, 3, generator=g) * 20 + 2
    an = torch.stack([anchors[:, 0], anchors[:, 1], anchors[:, 0] + size[:, 0], anchors[:, 1] + size[:, 1], anchors[:, 2],
                      anchors[:, 2] + size[:, 2]], 1).cuda()
    logits = (torch.randn(B * M, C, generator=g) * 2).cuda().requires_grad_(True)
    deltas = (torch.randn(B * M, 6, generator=g) * 0.5)
    deltas[3, 2] = 6.0; deltas[7, 5] = 4.2                                     # above bbox_xform_clip = log(1000 / 16) = 4.135
    deltas = deltas.cuda().requires_grad_(True)
    labels = torch.randint(0, C + 1, (B * M,), generator=g).float().cuda()
    gt = an.repeat(B, 1) + torch.randn(B * M, 6, generator=g).cuda()
    gt[:, 2:4] = torch.maximum(gt[:, 2:4], gt[:, 0:2] + 1); gt[:, 5] = torch.maximum(gt[:, 5], gt[:, 4] + 1)
    return an, logits, deltas, labels, gt


@pytest.mark.parametrize("C,npos,nneg", [(1, 5, 9), (3, 7, 20), (1, 0, 4)], ids=["c1", "c3", "nopos"])
def test_fused_detection_loss_matches_torch_ops(C, npos, nneg):
    """csrc/boxes.hip k_detloss / k_detloss_scatter against the torch expressions of DetectionHeadHNMNative._compute_loss_sync_free
    (decode_single, giou_diag, binary_cross_entropy_with_logits, masked sums): losses 1e-6, gradients 1e-5 relative."""
    from nndetection_amd.arch.heads import _DetLossFn
    from nndetection_amd.core.boxes.coder import decode_single, BBOX_XFORM_CLIP
    from nndetection_amd.core.boxes.ops import giou_diag
    M, B, P, Q = 64, 2, 10, 24
    an, logits, deltas, labels, gt = _rand_loss_inputs(M, B, C, 11 + C)
    fg = torch.where(labels > 0)[0][:npos]
    if npos:
        fg = torch.cat([torch.tensor([3, 7], device="cuda"), fg])[:npos].unique()       # the clamped rows are among the positives
        labels[fg] = labels[fg].clamp(min=1)
    bgr = torch.where(labels == 0)[0][:nneg]
    pos = torch.full((P,), -1, dtype=torch.int64, device="cuda"); pos[:fg.numel()] = fg
    neg = torch.full((Q,), -1, dtype=torch.int64, device="cuda"); neg[:bgr.numel()] = bgr
    counts = torch.tensor([fg.numel(), bgr.numel(), 0, 0], dtype=torch.int64, device="cuda")
    cfg = {"eps": 1e-7, "clip": BBOX_XFORM_CLIP, "reg_w": 1.0, "reg_mean": False, "cls_w": 1.0, "cls_mean": True}
    out = _DetLossFn.apply(logits, deltas, pos, neg, counts, labels, gt, an, cfg)
    (out[0] * 1.7 + out[1] * 0.6).backward()
    g_l, g_d = logits.grad.clone(), deltas.grad.clone()
    logits.grad = None; deltas.grad = None
    # torch reference on the compact lists
    pi, ni = pos[:fg.numel()], neg[:bgr.numel()]
    pred = decode_single(deltas[pi], an[pi % M])
    reg = -1 * giou_diag(pred, gt[pi], eps=1e-7).sum() / max(1, pi.numel()) if pi.numel() else deltas.sum() * 0
    idx = torch.cat([pi, ni])
    onehot = torch.nn.functional.one_hot(labels[idx].long(), C + 1)[:, 1:].float()
    cls = torch.nn.functional.binary_cross_entropy_with_logits(logits[idx], onehot, reduction="mean")
    (reg * 1.7 + cls * 0.6).backward()
    assert abs(float(out[0].detach()) - float(reg.detach())) <= 1e-6 * max(1.0, abs(float(reg.detach()))), (float(out[0].detach()), float(reg.detach()))
    assert abs(float(out[1].detach()) - float(cls.detach())) <= 1e-6 * max(1.0, abs(float(cls.detach()))), (float(out[1].detach()), float(cls.detach()))
    assert float((g_l - logits.grad).abs().max()) <= 1e-5 * float(logits.grad.abs().max())
    if pi.numel():
        assert float((g_d - deltas.grad).abs().max()) <= 1e-5 * float(deltas.grad.abs().max())
        assert float(g_d[3, 2]) == 0.0 and float(deltas.grad[3, 2]) == 0.0                 # clamped delta: no gradient on both routes
    else:
        assert not g_d.any()


def test_fused_detection_loss_in_train_step(golden_dir, monkeypatch):
    """One training step of the tiny model with the fused loss tail (default) and with the torch ops: same losses, same gradients."""
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    x = torch.from_numpy(gn["x"]).cuda()
    monkeypatch.setattr(torch, "randperm", det_randperm)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(DetectionHeadHNMNative, "fused_loss", mode)
        net.zero_grad(set_to_none=True)
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        res[mode] = ({k: float(v.detach()) for k, v in losses.items()},
                     {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    (l1, g1), (l0, g0) = res[True], res[False]
    for k in l0:
        assert abs(l1[k] - l0[k]) <= 1e-6 * max(1.0, abs(l0[k])), (k, l1[k], l0[k])
    assert set(g1) == set(g0)
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-12
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-5 * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_stream_overlap_does_not_change_training(golden_dir, monkeypatch, dtype):
    """Four optimizer steps on toy64; at every step the SAME parameters go through a train step with every overlap feature on (head
    branches, target assignment, segmentation branch and the full-resolution decoder tail on side streams, weight gradients on their
    own stream) and with everything on the caller's stream: same losses, same gradients for every parameter. A missing stream dependency shows up as a difference
    here -- the kernels and their inputs are identical, only the order of atomically reduced sums may differ (fp32: 2e-5 of the
    tensor maximum; bf16: storage noise of the atomically accumulated statistics, 2e-2). Parameters after several steps are NOT
    compared: two runs of the same configuration already differ by 4e-4 there (tools/diag_overlap.py: chaotic amplification)."""
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    from nndetection_amd.core.retina import BaseRetinaNet
    from nndetection_amd.ptmodule import build_model, configure_optimizer
    from nndetection_amd.arch.decoder import UFPNModular
    from tests.gpu_util import synth_inputs
    plan = get_plan("toy64")
    x, tg = synth_inputs(plan)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    torch.manual_seed(0)
    net = build_model(plan).cuda()
    opt, sched = configure_optimizer(net)
    for g in opt.param_groups:
        g["lr"] = 1e-2
    ltol, gtol = (2e-5, 2e-5) if dtype == torch.float32 else (1e-3, 2e-2)
    for it in range(4):
        res = {}
        for mode in (True, False):
            monkeypatch.setattr(DetectionHeadHNMNative, "multi_stream", mode)
            monkeypatch.setattr(BaseRetinaNet, "overlap_aux", mode)
            monkeypatch.setattr(L.wgrad_streams, "enabled", mode)
            monkeypatch.setattr(UFPNModular, "split_tail", mode)
            opt.zero_grad(set_to_none=True)
            losses, _ = net.train_step(x.cuda().to(dtype), _cuda_targets(tg), evaluation=False)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            res[mode] = ({k: float(v.detach()) for k, v in losses.items()},
                         {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
        (l1, g1), (l0, g0) = res[True], res[False]
        for k in l0:
            assert abs(l1[k] - l0[k]) <= ltol * max(1.0, abs(l0[k])), (it, k, l1[k], l0[k])
        assert set(g1) == set(g0)
        for n in g0:
            d = float((g1[n] - g0[n]).abs().max())
            assert d <= gtol * (float(g0[n].abs().max()) + 1e-9), (it, n, d, float(g0[n].abs().max()))
        opt.step()


def test_foreign_fused_optimizer_invalidates_every_parameter_cache():
    """`amd_fuse_sgd` switches the reference's torch.optim.SGD to fused=True, and torch._fused_sgd_ writes the
    parameters WITHOUT advancing their version counters. Everything cached from a parameter -- packed / cast weights, padded biases,
    the STEM's (1 input channel) padded bias included -- must be invalidated by the optimizer step itself (the step post-hook bumps
    `arch.conv.PARAM_GENERATION`), not only by the unconditional re-pack of a training-mode `BaseRetinaNet.forward`: here the blocks
    are called directly (no prepack_all) and must see the updated parameters after every step."""
    import torch.nn.functional as F
    from nndetection_amd.arch.conv import ConvInstanceRelu, PARAM_GENERATION
    from nndetection_amd.ptmodule import amd_fuse_sgd
    torch.manual_seed(0)
    stem = ConvInstanceRelu(3, 1, 20, 3, stride=1, padding=1, add_norm=False, add_act=False).cuda()     # bias=True: conv.py:113
    conv = ConvInstanceRelu(3, 32, 40, 3, stride=1, padding=1, add_norm=False, add_act=False).cuda()
    assert stem.conv.bias is not None and conv.conv.bias is not None
    params = list(stem.parameters()) + list(conv.parameters())
    opt = torch.optim.SGD(params, lr=0.5, momentum=0.9, nesterov=True)
    assert amd_fuse_sgd(opt) and opt.param_groups[0]["fused"]
    x1 = torch.randn(1, 1, 12, 10, 8, device="cuda")
    x2 = torch.randn(1, 32, 12, 10, 8, device="cuda")
    for it in range(3):
        y1, y2 = stem(x1), conv(x2)
        r1 = F.conv3d(x1, stem.conv.weight, stem.conv.bias, padding=1)
        r2 = F.conv3d(x2, conv.conv.weight, conv.conv.bias, padding=1)
        assert float((y1 - r1).detach().abs().max()) <= 2e-5 * float(r1.detach().abs().max()), ("stem", it)
        assert float((y2 - r2).detach().abs().max()) <= 2e-5 * float(r2.detach().abs().max()), ("conv", it)
        opt.zero_grad()
        (y1.square().mean() + y2.square().mean()).backward()
        gen, v = PARAM_GENERATION[0], [p._version for p in params]
        before = [p.detach().clone() for p in params]
        opt.step()
        assert PARAM_GENERATION[0] == gen + 1, "the step post-hook did not run"
        assert all(float((p.detach() - b).abs().max()) > 0 for p, b in zip(params, before)), "the step changed every parameter"
    with pytest.raises(ValueError):
        opt.add_param_group({"params": [torch.zeros(3, requires_grad=True)]})           # CPU tensor into a fused optimizer


def test_lazy_target_assignment_equals_the_gathered_one(golden_dir, monkeypatch):
    """Round 4: the batched ATSS kernel writes the anchors' labels itself (nndet_atss3d_assign_batched_f32) and the
    matched boxes stay a `MatchedBoxes` (GT boxes + matches; the detection loss reads <= 42 rows through nndet_detloss_matched_f32)
    instead of the clamp / gather / compare / multiply chain over [B, M] tensors and the [B, M, 6] gather of
    nndet/core/retina.py:262-287. Same labels and boxes bit for bit -- incl. an image without objects and non-zero classes -- and the
    same losses / gradients of a training step."""
    from nndetection_amd.core import retina as R
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    x = torch.from_numpy(gn["x"]).cuda()
    with torch.no_grad():
        _, anchors, _ = net(x)
    boxes = [b.cuda() for b in tg["target_boxes"]]
    classes = [torch.arange(b.shape[0], dtype=torch.float32).cuda() % 3 for b in boxes]      # classes 0, 1, 2, ...
    for variant in ("as is", "first image empty"):
        bx_ = [boxes[0][:0]] + boxes[1:] if variant == "first image empty" else boxes
        cl_ = [classes[0][:0]] + classes[1:] if variant == "first image empty" else classes
        lab0, mb0 = net.assign_targets_to_anchors(anchors, bx_, cl_, lazy=False)
        lab1, mb1 = net.assign_targets_to_anchors(anchors, bx_, cl_, lazy=True)
        assert isinstance(mb1, R.MatchedBoxes) and not isinstance(mb0, R.MatchedBoxes)
        for a, b in zip(lab0, lab1):
            assert torch.equal(a, b), variant
        for a, b, l in zip(mb0, mb1.materialize(), lab0):
            assert torch.equal(a[l > 0], b[l > 0]), variant               # (rows of unmatched anchors are never read: index 0 in both)
        assert sum(int((l > 0).sum()) for l in lab0) > 0
    res = {}
    for lazy in (True, False):
        monkeypatch.setattr(R, "LAZY_TARGETS", lazy)
        net.zero_grad(set_to_none=True)
        losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        res[lazy] = ({k: float(v.detach()) for k, v in losses.items()},
                     {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    # (two training steps of the SAME configuration differ in the last bits: the statistics / bias / stem sums are atomics whose order
    # varies from run to run, DESIGN.md section 8 -- an exact comparison here failed in 2 of 4 runs)
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 1e-6 * max(1.0, abs(v)), (k, res[True][0][k], v)
    for n, g in res[False][1].items():
        assert float((res[True][1][n] - g).abs().max()) <= 1e-4 * float(g.abs().max()) + 1e-12, n


def test_copied_factorised_gradient_fails_loudly(golden_dir, monkeypatch):
    """VERDICT r3 ("works, tested, will not scale past this model"): the fused segmentation head hands its input gradient on in factorised
    form through an unwritten tensor registered under its address. If anything between the head and decoder.out.P0 replaces that tensor
    (here: a tensor hook that clones the gradient) the receiving node must raise -- not train on undefined memory."""
    from nndetection_amd import _lib as L
    from nndetection_amd.arch import segmenter as S
    gn, plan, tg = _load(golden_dir)
    ora = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    net = _hip_model(plan, ora)
    monkeypatch.setattr(torch, "randperm", det_randperm)
    monkeypatch.setattr(S, "SEG_BRANCH", False)                  # the rank-1 route (the composed branch does not use the side channel)
    x = torch.from_numpy(gn["x"]).cuda().to(torch.bfloat16)
    orig = net.segmenter.forward

    def hooked(fm, fused=False):
        out = orig(fm, fused=fused)
        if "seg_input" in out:
            src = out["seg_input"]
            y = src * 1.0                                         # an extra autograd node whose backward hands on a fresh tensor
            for a in ("_nndet_rank1_ok", "_nndet_padded"):
                if hasattr(src, a):
                    setattr(y, a, getattr(src, a))
            out["seg_input"] = y
        return out

    losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
    sum(losses.values()).backward()                               # the normal route works
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    net.zero_grad(set_to_none=True)
    monkeypatch.setattr(net.segmenter, "forward", hooked)
    losses, _ = net.train_step(x, _cuda_targets(tg), evaluation=False)
    with pytest.raises((L.NndetError, RuntimeError), match="factorised gradient"):
        sum(losses.values()).backward()
    torch.cuda.synchronize()
