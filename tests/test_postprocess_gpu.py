"""GPU parity of the fused post-processing front end (csrc/postproc.hip) against the numpy oracle
`oracle.boxes_np.postprocess_single_image` (restatement of nndet/core/retina.py:332-379, pinned by the reference-generated
detections of tests/golden/net_*_golden.npz through the end-to-end tests)."""
import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from tests.gpu_util import rand_boxes, t

pytestmark = pytest.mark.gpu


def _oracle(boxes, probs, shape, C, **kw):
    return bx.postprocess_single_image(boxes, probs, shape, num_classes=C, **kw)


@pytest.mark.parametrize("M,C,topk,thr,seed", [(5000, 1, 1000, 0.0, 0), (20000, 3, 10000, 0.3, 1), (777, 2, 10000, 0.0, 2),
                                                (40000, 1, 10000, 0.0, 3)])
def test_fused_postprocess_probs_and_boxes_bit_exact(M, C, topk, thr, seed):
    """Decoded boxes + probabilities in (the single-image contract of the reference): everything after that is integer /
    comparison work, so boxes, scores and labels must be IDENTICAL to the oracle -- including score ties (lowest flat index
    first), boxes that the clip makes smaller than `remove_small_boxes`, and several images per call."""
    from nndetection_amd.core.boxes import postprocess_batch
    rng = np.random.default_rng(seed)
    shape = (64, 48, 40)
    B = 3
    boxes = np.stack([rand_boxes(rng, M, extent=(70, 54, 44), smin=3, smax=20) for _ in range(B)])
    probs = rng.uniform(0.0, 1.0, (B, M, C)).astype(np.float32)
    probs[:, 5:40] = probs[:, 4:5]                          # ties on purpose
    probs[1] *= 0.25                                        # image 1: everything below the threshold of the 0.3 case
    boxes[0, :50, 2] = boxes[0, :50, 0] + 0.005             # thinner than min_size 0.01
    gb, gs, gl = postprocess_batch(t(probs), t(boxes), None, shape, C, topk, thr, 0.01, 0.5, 100, scores_are_probs=True)
    for b in range(B):
        rb, rs, rl = _oracle(boxes[b], probs[b], shape, C, topk=topk, score_thresh=thr, min_size=0.01, nms_thresh=0.5,
                             detections_per_img=100)
        assert gb[b].shape == rb.shape, (b, gb[b].shape, rb.shape)
        assert np.array_equal(gl[b].cpu().numpy(), rl), f"labels differ in image {b}"
        assert np.array_equal(gs[b].cpu().numpy(), rs), f"scores differ in image {b}"
        assert np.array_equal(gb[b].cpu().numpy(), rb), f"boxes differ in image {b}"


def test_fused_postprocess_logits_deltas_anchors():
    """The in-network contract: logits + regression deltas + shared anchors (sigmoid and decode inside the kernel, only for the
    survivors). exp / sigmoid are library functions, so values are compared at 1e-4 (north_star) and the selection itself
    (labels, counts) exactly; the seed is one where no decision sits within an ulp of a threshold."""
    from nndetection_amd.core.boxes import postprocess_batch
    rng = np.random.default_rng(7)
    shape = (64, 48, 40)
    W = [(4, 8, 16), (8, 16, 32)]
    anchors, _ = bx.anchors_for_image(shape, [(16, 12, 10), (8, 6, 5)], W, W, W)
    M, C, B = anchors.shape[0], 2, 2
    # logits on a grid (random order): neighbouring probabilities differ by >> 1 ulp, so the ranking does not depend on whose
    # exp() rounds which way
    logits = rng.permutation(np.linspace(-6.0, 2.0, B * M * C)).reshape(B, M, C).astype(np.float32)
    deltas = (rng.standard_normal((B, M, 6)) * 0.3).astype(np.float32)
    deltas[0, 11, 3] = 9.0                                  # exp clamp
    gb, gs, gl = postprocess_batch(t(logits), t(deltas), t(anchors), shape, C, 10000, 0.0, 0.01, 0.6, 100)
    for b in range(B):
        dec = bx.decode_single(deltas[b], anchors)
        rb, rs, rl = _oracle(dec, bx.sigmoid(logits[b]), shape, C, topk=10000, score_thresh=0.0, min_size=0.01, nms_thresh=0.6,
                             detections_per_img=100)
        assert gb[b].shape == rb.shape
        assert np.array_equal(gl[b].cpu().numpy(), rl)
        assert np.abs(gs[b].cpu().numpy() - rs).max() <= 1e-6
        assert np.abs(gb[b].cpu().numpy() - rb).max() <= 1e-4


def test_fused_postprocess_edge_cases():
    from nndetection_amd.core.boxes import postprocess_batch
    rng = np.random.default_rng(1)
    boxes = rand_boxes(rng, 300, extent=(30, 30, 30), smin=3, smax=10)[None]
    probs = rng.uniform(0.1, 0.9, (1, 300, 1)).astype(np.float32)
    # no top-k, no threshold, no small-box filter, no detection cap: plain batched NMS of everything
    gb, gs, gl = postprocess_batch(t(probs), t(boxes), None, None, 1, None, None, None, 0.4, None, scores_are_probs=True)
    keep = bx.nms(boxes[0], probs[0, :, 0], 0.4)
    assert np.array_equal(gb[0].cpu().numpy(), boxes[0][keep]) and np.array_equal(gs[0].cpu().numpy(), probs[0, keep, 0])
    # everything filtered by the score threshold -> empty results, correct dtypes
    gb, gs, gl = postprocess_batch(t(probs), t(boxes), None, (30, 30, 30), 1, 100, 0.95, 0.01, 0.4, 10, scores_are_probs=True)
    assert gb[0].shape == (0, 6) and gs[0].shape == (0,) and gl[0].dtype == torch.int64
    # M smaller than one workgroup's batch, top-k larger than M
    gb, gs, gl = postprocess_batch(t(probs[:, :7]), t(boxes[:, :7]), None, (30, 30, 30), 1, 10000, 0.0, 0.01, 0.4, 100, scores_are_probs=True)
    rb, rs, rl = _oracle(boxes[0, :7], probs[0, :7], (30, 30, 30), 1, topk=10000, score_thresh=0.0, min_size=0.01, nms_thresh=0.4)
    assert np.array_equal(gb[0].cpu().numpy(), rb) and np.array_equal(gs[0].cpu().numpy(), rs)


def test_detector_single_image_contract():
    """BaseRetinaNet.postprocess_detections_single_image(boxes, probs, image_shape) keeps the reference signature."""
    from nndetection_amd.plans import get_plan
    from nndetection_amd.ptmodule import build_model
    net = build_model(get_plan("tiny"))
    rng = np.random.default_rng(2)
    boxes = rand_boxes(rng, 2000, extent=(32, 32, 24), smin=3, smax=10)
    probs = rng.uniform(0, 1, (2000, 1)).astype(np.float32)
    b, p, l = net.postprocess_detections_single_image(t(boxes), t(probs), (32, 32, 24))
    rb, rs, rl = _oracle(boxes, probs, (32, 32, 24), 1, topk=10000, score_thresh=0.0, min_size=0.01, nms_thresh=0.6)
    assert np.array_equal(b.cpu().numpy(), rb) and np.array_equal(p.cpu().numpy(), rs) and np.array_equal(l.cpu().numpy(), rl)


def test_reference_fixtures_single_image_sampler_and_ensembler_stage(golden_dir):
    """DIRECT fixtures generated by the unmodified reference (tests/golden/postproc_golden.npz):
    `BaseRetinaNet.postprocess_detections_single_image` (nndet/core/retina.py:332-379), `HardNegativeSamplerBatched`
    (nndet/core/boxes/sampler.py:237-270, randperm := reversed arange) and the ensembler's `postprocess_image`
    (nndet/inference/ensembler/detection.py:166-217) -- each through the fused HIP pass, bit-exact."""
    import os
    from types import SimpleNamespace
    from nndetection_amd.core.retina import BaseRetinaNet
    from nndetection_amd.core.boxes import HardNegativeSamplerBatched
    from nndetection_amd.inference import postprocess_image_fused, amd_box_ensembler
    g = np.load(os.path.join(golden_dir, "postproc_golden.npz"))
    for tag in ("c1", "c3"):
        C, topk, dets, *shape = [int(v) for v in g[f"pp_{tag}_cfg"]]
        holder = SimpleNamespace(topk_candidates=topk, score_thresh=float(g[f"pp_{tag}_thr"]), num_foreground_classes=C,
                                 remove_small_boxes=0.01, nms_thresh=0.6, detections_per_img=dets)
        b, p, l = BaseRetinaNet.postprocess_detections_single_image(holder, t(g[f"pp_{tag}_boxes"]), t(g[f"pp_{tag}_probs"]), shape)
        assert np.array_equal(b.cpu().numpy(), g[f"pp_{tag}_out_boxes"]), tag
        assert np.array_equal(p.cpu().numpy(), g[f"pp_{tag}_out_scores"]) and np.array_equal(l.cpu().numpy(), g[f"pp_{tag}_out_labels"])
    s = HardNegativeSamplerBatched(32, 0.33, min_neg=1, pool_size=20)
    s.deterministic = True
    labels = t(g["hnm_labels"])
    pm, nm = s(list(labels.split([int(v) for v in g["hnm_per_img"]])), t(g["hnm_fg"]))
    assert np.array_equal(torch.where(torch.cat(pm))[0].cpu().numpy(), g["hnm_pos"])
    assert np.array_equal(torch.where(torch.cat(nm))[0].cpu().numpy(), g["hnm_neg"])
    args = (t(g["ens_boxes"]), t(g["ens_probs"]), t(g["ens_labels"]), t(g["ens_weights"]))
    b, p, l, w = postprocess_image_fused(*args, tuple(int(v) for v in g["ens_shape"]), 1000, 0.1, 0.01, 0.1, 100)
    assert np.array_equal(b.cpu().numpy(), g["ens_out_boxes"]) and np.array_equal(p.cpu().numpy(), g["ens_out_probs"])
    assert np.array_equal(l.cpu().numpy(), g["ens_out_labels"]) and np.array_equal(w.cpu().numpy(), g["ens_out_weights"])

    class RefLike:                                     # what BoxEnsemblerSelective contributes: .parameters + the torch fallback
        def __init__(self, fn):
            self.parameters = {"model_iou": 0.1, "model_nms_fn": fn, "model_score_thresh": 0.1, "model_topk": 1000,
                               "model_detections_per_image": 100, "remove_small_boxes": 1e-2}

        def postprocess_image(self, *a, **k):
            return "reference path"

    def batched_nms_model(*a, **k):
        raise AssertionError("the fused pass replaces the call of model_nms_fn")

    ens = amd_box_ensembler(RefLike)(batched_nms_model)
    b2, p2, l2, w2 = ens.postprocess_image(*args, tuple(int(v) for v in g["ens_shape"]))
    assert torch.equal(b2, b) and torch.equal(w2, w)
    other = amd_box_ensembler(RefLike)(lambda *a, **k: None)            # any other model_nms_fn keeps the reference's sequence
    assert other.postprocess_image(*args, None) == "reference path"
    e = torch.zeros(0, 6, device="cuda")
    assert postprocess_image_fused(e, e[:, 0], e[:, 0].long(), e[:, 0], None, 1000, 0.0, 0.01, 0.1, 100)[0].shape == (0, 6)
