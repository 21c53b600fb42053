"""CPU: the oracle restatement (oracle/) against the golden vectors produced from the real reference
(tests/golden/make_golden.py). Bit-exact for box / index work, 1e-5 for the network."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from oracle.detweights import fill_state
from oracle.retina_torch import OracleRetinaUNet
from nndetection_amd.plans import get_plan, MODEL_CFG_V001


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "boxes_golden.npz"))


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(a, b, equal_nan=True)


def test_iou_giou_bit_exact(g):
    biteq(bx.box_iou(g["iou_b1"], g["iou_b2"]), g["iou"])
    biteq(bx.box_iou(g["iou_b1"], g["iou_b2"], 1e-6), g["iou_eps"])
    biteq(bx.generalized_box_iou(g["iou_b1"], g["iou_b2"]), g["giou"])
    biteq(bx.generalized_box_iou(g["iou_b1"], g["iou_b2"], 1e-7), g["giou_eps"])


def test_center_distance(g):
    d = bx.box_center_dist(g["iou_b1"], g["iou_b2"])
    ulp = np.abs(d.view(np.int32).astype(np.int64) - g["cdist"].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1          # torch-CPU sqrt is not correctly rounded (see make_golden.py)
    biteq((d.astype(np.float64) ** 2).astype(np.float32) * 0 + g["cdist_sq"], g["cdist_sq"])


def test_anchors_bit_exact(g):
    W = [(4, 8, 16), (8, 16, 32), (16, 32, 64)]
    a, npl = bx.anchors_for_image((48, 40, 24), [(12, 10, 6), (6, 5, 3), (3, 3, 3)], W, W, W)
    biteq(a, g["anchors"])
    assert npl == list(g["anchors_per_level"])


def test_atss_matches(g):
    _, m = bx.atss_match(g["atss_gt"], g["anchors"], list(g["anchors_per_level"]), 27, 4)
    biteq(m, g["atss_matches"])
    assert (m >= 0).sum() > 100
    _, m0 = bx.atss_match(np.zeros((0, 6), np.float32), g["anchors"], list(g["anchors_per_level"]), 27, 4)
    assert (m0 == -1).all()


@pytest.mark.parametrize("key", ["300_0.6", "1500_0.1", "1500_0.6"])
def test_nms_bit_exact(g, key):
    thr = float(key.split("_")[1])
    biteq(bx.nms(g[f"nms_boxes_{key}"], g[f"nms_scores_{key}"], thr), g[f"nms_keep_{key}"])


def test_batched_nms_and_edges(g):
    biteq(bx.batched_nms(g["bnms_boxes"], g["bnms_scores"], g["bnms_cls"], 0.5), g["bnms_keep"])
    assert bx.nms(np.zeros((0, 6)), np.zeros((0,)), 0.5).shape == (0,)
    assert bx.batched_nms(np.zeros((0, 6)), np.zeros((0,)), np.zeros((0,)), 0.5).shape == (0,)
    one = np.asarray([[0, 0, 1, 1, 0, 1]], np.float32)
    assert list(bx.nms(one, np.asarray([0.3], np.float32), 0.5)) == [0]


def test_atss_center_in_gt_iou_matcher_and_2d_nms(g):
    """Round 4: what the reference's matcher / native API offers beyond RetinaUNetV001 -- ATSSMatcher(center_in_gt=True)
    (matcher/atss.py:101-107), IoUMatcher (matcher/iou.py:43-107), 2D NMS (nms.cu:22-34,54-96 / nms_cpu on [N, 4] boxes)."""
    npl = [int(v) for v in g["anchors_per_level"]]
    gt_c = g["atss_gt_center_in_gt"]
    _, m = bx.atss_match(gt_c, g["anchors"], npl, 27, 4, center_in_gt=True)
    biteq(m, g["atss_matches_center_in_gt"])
    _, m_plain = bx.atss_match(gt_c, g["anchors"], npl, 27, 4)
    assert (m_plain >= 0).sum() > (m >= 0).sum() > 0
    for low, high, lq in ((0.3, 0.5, False), (0.4, 0.6, True)):
        biteq(bx.iou_match(gt_c, g["anchors"], low, high, lq)[1], g[f"ioumatch_{low}_{high}_{int(lq)}"])
    assert bx.iou_match(gt_c[:0], g["anchors"], 0.3, 0.5, True)[1].tolist() == [-1] * len(g["anchors"])
    for thr in (0.1, 0.5):
        biteq(bx.nms2d(g["nms2d_boxes"], g["nms2d_scores"], thr), g[f"nms2d_keep_{thr}"])


def test_decode_clip(g):
    d = bx.decode_single(g["dec_rel"], g["anchors"])
    assert np.abs(d - g["dec_boxes"]).max() < 1e-4      # exp() is library dependent
    biteq(bx.clip_boxes_to_image(g["dec_boxes"], (48, 40, 24)), g["clip_boxes"])


def test_sampler_counts():
    # HardNegativeSamplerBatched(32, 0.33, min_neg=1, pool 20), B = 4: num_pos <= int(128*0.33) = 42
    assert bx.hnm_counts(1000, 10 ** 6, 4, 32, 0.33, 1, 20) == (42, 85, 1700)
    assert bx.hnm_counts(0, 10 ** 6, 4, 32, 0.33, 1, 20) == (0, 2, 40)
    assert bx.hnm_counts(5, 3, 2, 32, 0.33, 1, 20) == (5, 3, 3)


def det_randperm(n, *a, **k):
    return torch.arange(n - 1, -1, -1, device=k.get("device", None))


def test_network_tiny_against_reference_golden(golden_dir, monkeypatch):
    gn = np.load(os.path.join(golden_dir, "net_tiny_golden.npz"))
    plan = get_plan("tiny")
    net = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    B = plan["batch_size"]
    tg = {"target_boxes": [torch.from_numpy(gn[f"gt_boxes_{i}"]) for i in range(B)],
          "target_classes": [torch.from_numpy(gn[f"gt_classes_{i}"]) for i in range(B)],
          "target_seg": torch.from_numpy(gn["target_seg"].astype(np.float32))}
    monkeypatch.setattr(torch, "randperm", det_randperm)
    losses, pred = net.train_step(torch.from_numpy(gn["x"]), tg, evaluation=True)
    for k, v in losses.items():
        assert abs(v.item() - float(gn[f"loss_{k}"])) < 1e-5, k
    sum(losses.values()).backward()
    norms = {k: (p.grad.norm().item() if p.grad is not None else -1.0) for k, p in net.named_parameters()}
    for k, ref in zip(gn["grad_names"], gn["grad_norms"]):
        assert abs(norms[str(k)] - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), k
    # the never-used out conv has no gradient in the reference either
    assert norms["decoder.out.P0.0.conv.weight"] >= 0
    for b in range(B):
        assert np.allclose(pred["pred_boxes"][b], gn[f"det_boxes_{b}"], atol=1e-4)
        assert np.allclose(pred["pred_scores"][b], gn[f"det_scores_{b}"], atol=1e-5)
        assert np.array_equal(pred["pred_labels"][b], gn[f"det_labels_{b}"])


def test_network_toy64_config0_against_reference_golden(golden_dir, monkeypatch):
    """BASELINE.json configs[0] (64^3 patches, 5 stages, the reference's own CPU-runnable case): the oracle reproduces the losses,
    all 80 gradient norms and the detections the UNMODIFIED reference produced in the build container. The inputs are regenerated
    from the seed (tests/gpu_util.synth_inputs) and pinned by a checksum."""
    from tests.gpu_util import synth_inputs
    gn = np.load(os.path.join(golden_dir, "net_toy64_golden.npz"))
    plan = get_plan("toy64")
    x, tg = synth_inputs(plan)
    assert abs(float(x.double().sum()) - float(gn["x_checksum"])) < 1e-6, "the seeded input differs from the one the golden was made with"
    net = fill_state(OracleRetinaUNet(plan["arch"], plan["anchors"], MODEL_CFG_V001))
    monkeypatch.setattr(torch, "randperm", det_randperm)
    losses, pred = net.train_step(x, tg, evaluation=True)
    for k, v in losses.items():
        assert abs(v.item() - float(gn[f"loss_{k}"])) < 1e-5, k
    sum(losses.values()).backward()
    norms = {k: (p.grad.norm().item() if p.grad is not None else -1.0) for k, p in net.named_parameters()}
    for k, ref in zip(gn["grad_names"], gn["grad_norms"]):
        assert abs(norms[str(k)] - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), k
    for b in range(plan["batch_size"]):
        assert np.allclose(pred["pred_boxes"][b], gn[f"det_boxes_{b}"], atol=1e-4)
        assert np.allclose(pred["pred_scores"][b], gn[f"det_scores_{b}"], atol=1e-5)
        assert np.array_equal(pred["pred_labels"][b], gn[f"det_labels_{b}"])


def test_target_preparation_oracle_vs_reference_golden(golden_dir):
    """the oracle restatement of FindInstances / Instances2Boxes / Instances2Segmentation against what the
    unmodified reference transforms produced (tests/golden/make_golden.py:golden_targets)."""
    import ast
    g = np.load(os.path.join(golden_dir, "targets_golden.npz"))
    tgt = g["target"].astype(np.float32)
    maps = [ast.literal_eval(str(m)) for m in g["maps"]]
    for b in range(tgt.shape[0]):
        ob, oc, oi, osem = bx.instances_to_targets(tgt[b, 0], maps[b])
        assert np.array_equal(ob, g[f"boxes_{b}"]) and np.array_equal(oc, g[f"classes_{b}"]) and np.array_equal(oi, g[f"ids_{b}"])
        assert np.array_equal(osem.astype(np.uint8), g[f"seg_{b}"])


def test_wbc_oracle_vs_reference_golden(golden_dir):
    """the oracle restatement of batched_wbc / wbc against the unmodified reference (make_golden.py:golden_wbc)."""
    g = np.load(os.path.join(golden_dir, "wbc_golden.npz"))
    for tag, kw in (("a", dict(iou_thresh=0.3, score_thresh=0.0, use_area=False, missing_weight=1.0)),
                    ("b", dict(iou_thresh=0.1, score_thresh=0.2, use_area=True, missing_weight=0.5))):
        ob, os_, ol = bx.batched_wbc(g["boxes"], g["scores"], g["labels"], g["weights"], kw["iou_thresh"], g["n_exp"],
                                     kw["score_thresh"], kw["use_area"], kw["missing_weight"])
        assert ob.shape == g[f"out_boxes_{tag}"].shape
        assert np.allclose(ob, g[f"out_boxes_{tag}"], rtol=1e-5, atol=1e-5) and np.allclose(os_, g[f"out_scores_{tag}"], rtol=1e-5, atol=1e-6)
        assert np.array_equal(ol, g[f"out_labels_{tag}"])


def test_postproc_oracles_vs_reference_golden(golden_dir):
    """postproc_golden.npz (tests/golden/make_golden.py golden_postproc): the reference's postprocess_detections_single_image,
    HardNegativeSamplerBatched (randperm := reversed arange) and BoxEnsemblerSelective.postprocess_image, each against its numpy
    restatement -- bit-exact (index / compare work on fp32 values that both sides compute with the same operations)."""
    g = np.load(os.path.join(golden_dir, "postproc_golden.npz"))
    for tag in ("c1", "c3"):
        C, topk, dets, *shape = [int(v) for v in g[f"pp_{tag}_cfg"]]
        b, p, l = bx.postprocess_single_image(g[f"pp_{tag}_boxes"].copy(), g[f"pp_{tag}_probs"], shape, C, topk, float(g[f"pp_{tag}_thr"]), 0.01, 0.6, dets)
        assert np.array_equal(b, g[f"pp_{tag}_out_boxes"]) and np.array_equal(p, g[f"pp_{tag}_out_scores"])
        assert np.array_equal(l, g[f"pp_{tag}_out_labels"])
    pos, neg, _ = bx.hnm_select_reversed(g["hnm_labels"], g["hnm_fg"], len(g["hnm_per_img"]), 32, 0.33, 1, 20)
    assert np.array_equal(pos, g["hnm_pos"]) and np.array_equal(neg, g["hnm_neg"])
    b, p, l, w = bx.ensembler_postprocess_image(g["ens_boxes"], g["ens_probs"], g["ens_labels"], g["ens_weights"], tuple(g["ens_shape"]),
                                                1000, 0.1, 0.01, 0.1, 100)
    assert np.array_equal(b, g["ens_out_boxes"]) and np.array_equal(p, g["ens_out_probs"])
    assert np.array_equal(l, g["ens_out_labels"]) and np.array_equal(w, g["ens_out_weights"])


def test_atss_blocked_equals_dense():
    """`atss_match_blocked` (the oracle of the config-5-sized ATSS GPU test) == `atss_match` (pinned to the reference above) for every
    block size, incl. a GT centre exactly on the anchor lattice (distance ties at the k-th candidate) and single-row blocks."""
    rng = np.random.default_rng(5)
    W = [(4, 8, 16), (8, 16, 32), (16, 32, 64)]
    anchors, npl = bx.anchors_for_image((64, 48, 40), [(16, 12, 10), (8, 6, 5), (4, 3, 3)], W, W, W)
    for G in (1, 7, 40, 130):
        c = rng.uniform(0, 48, (G, 3)); s = rng.uniform(4, 24, (G, 3))
        gt = np.stack([c[:, 0] - s[:, 0] / 2, c[:, 1] - s[:, 1] / 2, c[:, 0] + s[:, 0] / 2, c[:, 1] + s[:, 1] / 2,
                       c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1).astype(np.float32)
        if G > 1:
            gt[0] = [10, 6, 22, 18, 2, 14]
        _, ref = bx.atss_match(gt, anchors, npl, 27, 4)
        for rows, threads in ((1, 1), (16, 4), (100, 1)):
            got = bx.atss_match_blocked(gt, anchors, npl, 27, 4, rows=rows, threads=threads)
            assert np.array_equal(ref, got), (G, rows)
    assert (bx.atss_match_blocked(np.zeros((0, 6), np.float32), anchors, npl, 27, 4) == -1).all()


# ---- differentiable GIoU (round 6): the torch restatement oracle/boxes_torch.py against the reference's autograd
def test_giou_gradients_torch_oracle_vs_reference_fixture(golden_dir):
    from oracle import boxes_torch as bt
    g = np.load(os.path.join(golden_dir, "giou_grad_golden.npz"))
    cot = torch.from_numpy(g["pw_cot"])
    for eps, tag in ((0.0, ""), (1e-7, "_eps")):
        b1, b2 = torch.from_numpy(g["pw_b1"]).requires_grad_(), torch.from_numpy(g["pw_b2"]).requires_grad_()
        bt.generalized_box_iou(b1, b2, eps=eps).backward(cot)
        np.testing.assert_allclose(b1.grad.numpy(), g[f"pw_ga{tag}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(b2.grad.numpy(), g[f"pw_gb{tag}"], rtol=1e-5, atol=1e-6)
    for n in (1, 42, 300):
        for red in ("sum", "mean"):
            p = torch.from_numpy(g[f"loss{n}_pred"]).requires_grad_()
            loss = bt.giou_loss(p, torch.from_numpy(g[f"loss{n}_tgt"]), eps=1e-7, reduction=red, loss_weight=2.0)
            loss.backward()
            assert abs(loss.item() - float(g[f"loss{n}_{red}"])) <= 1e-6 * max(1.0, abs(float(g[f"loss{n}_{red}"])))
            np.testing.assert_allclose(p.grad.numpy(), g[f"loss{n}_{red}_grad"], rtol=1e-5, atol=1e-7)
        # the matrix values of the torch oracle are the numpy oracle's, bit for bit
        m = bt.generalized_box_iou(torch.from_numpy(g[f"loss{n}_pred"]), torch.from_numpy(g[f"loss{n}_tgt"]), eps=1e-7).numpy()
        biteq(m, bx.generalized_box_iou(g[f"loss{n}_pred"], g[f"loss{n}_tgt"], 1e-7))
