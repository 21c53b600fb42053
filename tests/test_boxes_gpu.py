"""GPU parity of the box-op kernels against the CPU oracle and the reference golden vectors (bit-exact)."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from oracle import nms_c
from tests.gpu_util import rand_boxes, distinct_scores, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "boxes_golden.npz"))


def biteq(a, b, what=""):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not np.array_equal(a, b, equal_nan=True):
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} differ, first at {bad[0]}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}")


def test_iou_giou_golden(g):
    from nndetection_amd.core.boxes import box_iou, generalized_box_iou
    b1, b2 = t(g["iou_b1"]), t(g["iou_b2"])
    biteq(box_iou(b1, b2), g["iou"], "iou")
    biteq(box_iou(b1, b2, eps=1e-6), g["iou_eps"], "iou eps")
    biteq(generalized_box_iou(b1, b2), g["giou"], "giou")
    biteq(generalized_box_iou(b1, b2, eps=1e-7), g["giou_eps"], "giou eps")


@pytest.mark.parametrize("n,m", [(1, 1), (3, 1023), (17, 1024), (16, 1025), (33, 4100)])
def test_iou_shapes_vs_oracle(n, m):
    from nndetection_amd.core.boxes import box_iou, generalized_box_iou
    rng = np.random.default_rng(n * 7 + m)
    a, b = rand_boxes(rng, n), rand_boxes(rng, m)
    biteq(box_iou(t(a), t(b)), bx.box_iou(a, b), "iou")
    biteq(generalized_box_iou(t(a), t(b), eps=1e-7), bx.generalized_box_iou(a, b, 1e-7), "giou")


# ---- gradient of the full GIoU matrix (generalized_box_iou "must stay differentiable"; ops.py:106-128,162-185)
@pytest.fixture(scope="module")
def gg(golden_dir):
    return np.load(os.path.join(golden_dir, "giou_grad_golden.npz"))


def _close64(got, want64, what, rel=2e-6):
    """fp32 result of the HIP kernel (float64 partial sums, one rounding) against the float64 evaluation of the oracle expression"""
    got = got.detach().double().cpu().numpy()
    want = want64.detach().cpu().numpy()
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= rel, (what, err)


@pytest.mark.parametrize("eps,tag", [(0.0, ""), (1e-7, "_eps")])
def test_giou_pairwise_backward_fixture_random_cotangent(gg, eps, tag):
    """[37 x 501] with a random cotangent, ties included (an identical pair, a pair tied on five coordinates): against the reference's
    own autograd (fp32, fixture) and against autograd of the oracle expression in float64."""
    from nndetection_amd.core.boxes import generalized_box_iou
    from oracle import boxes_torch as bt
    b1, b2 = t(gg["pw_b1"]).requires_grad_(), t(gg["pw_b2"]).requires_grad_()
    m = generalized_box_iou(b1, b2, eps=eps)
    biteq(m.detach(), bx.generalized_box_iou(gg["pw_b1"], gg["pw_b2"], eps), "giou forward under autograd")
    m.backward(t(gg["pw_cot"]))
    o1 = torch.from_numpy(gg["pw_b1"]).double().requires_grad_()
    o2 = torch.from_numpy(gg["pw_b2"]).double().requires_grad_()
    bt.generalized_box_iou(o1, o2, eps=eps).backward(torch.from_numpy(gg["pw_cot"]).double())
    _close64(b1.grad, o1.grad, "d boxes1 vs float64")
    _close64(b2.grad, o2.grad, "d boxes2 vs float64")
    for got, ref in ((b1.grad, gg[f"pw_ga{tag}"]), (b2.grad, gg[f"pw_gb{tag}"])):      # the reference's fp32 autograd: its own summation noise
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    # only one side requires a gradient
    b1n = t(gg["pw_b1"]).requires_grad_()
    generalized_box_iou(b1n, t(gg["pw_b2"]), eps=eps).backward(t(gg["pw_cot"]))
    assert torch.equal(b1n.grad, b1.grad)
    b2n = t(gg["pw_b2"]).requires_grad_()
    generalized_box_iou(t(gg["pw_b1"]), b2n, eps=eps).backward(t(gg["pw_cot"]))
    assert torch.equal(b2n.grad, b2.grad)


@pytest.mark.parametrize("n", [1, 42, 300])
@pytest.mark.parametrize("red", ["sum", "mean"])
def test_reference_giou_loss_formula_trains_through_the_pairwise_op(gg, n, red):
    """The reference's GIoULoss (losses/regression.py:158-161): generalized_box_iou -> torch.diag -> reduction -> weight * -1, on OUR op;
    loss and d loss / d pred against the reference fixture and the float64 oracle; equals the O(P) `giou_diag` route bit for bit."""
    from nndetection_amd.core.boxes import generalized_box_iou, giou_diag
    from oracle import boxes_torch as bt
    pred, tgt = t(gg[f"loss{n}_pred"]).requires_grad_(), t(gg[f"loss{n}_tgt"])
    d = torch.diag(generalized_box_iou(pred, tgt, eps=1e-7), diagonal=0)
    loss = 2.0 * -1 * (d.sum() if red == "sum" else d.mean())
    loss.backward()
    ref = float(gg[f"loss{n}_{red}"])
    assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref))
    o = torch.from_numpy(gg[f"loss{n}_pred"]).double().requires_grad_()
    bt.giou_loss(o, torch.from_numpy(gg[f"loss{n}_tgt"]).double(), eps=1e-7, reduction=red, loss_weight=2.0).backward()
    _close64(pred.grad, o.grad, "d loss / d pred vs float64")
    np.testing.assert_allclose(pred.grad.cpu().numpy(), gg[f"loss{n}_{red}_grad"], rtol=0, atol=2e-5 * np.abs(gg[f"loss{n}_{red}_grad"]).max())
    p2 = t(gg[f"loss{n}_pred"]).requires_grad_()
    d2 = giou_diag(p2, tgt, eps=1e-7)
    (2.0 * -1 * (d2.sum() if red == "sum" else d2.mean())).backward()
    assert torch.equal(d2.detach(), d.detach())
    np.testing.assert_allclose(p2.grad.cpu().numpy(), pred.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_giou_pairwise_backward_empty_and_c_abi_argument_checks():
    from nndetection_amd.core.boxes import generalized_box_iou
    from nndetection_amd import _lib as L
    e = generalized_box_iou(torch.zeros(0, 6, device="cuda", requires_grad=True), torch.rand(4, 6, device="cuda"))
    assert e.numel() == 0
    a = torch.rand(3, 6, device="cuda"); g = torch.rand(3, 3, device="cuda")
    lib = L.load()
    assert lib.nndet_giou3d_pairwise_bwd_f32(L.ptr(a), 3, L.ptr(a), 3, L.ptr(g), 0.0, None, None, L.stream()) == -1   # no output wanted
    assert lib.nndet_giou3d_pairwise_bwd_f32(L.ptr(a), -1, L.ptr(a), 3, L.ptr(g), 0.0, L.ptr(a), None, L.stream()) == -1
    assert lib.nndet_giou3d_pairwise_bwd_f32(L.ptr(a), 0, L.ptr(a), 3, L.ptr(g), 0.0, L.ptr(a), None, L.stream()) == 0


def test_iou_empty():
    from nndetection_amd.core.boxes import box_iou
    e = box_iou(torch.zeros(0, 6, device="cuda"), torch.rand(4, 6, device="cuda"))
    assert e.numel() == 0


def test_anchors_golden(g):
    from nndetection_amd.core.boxes import AnchorGenerator3DS
    W = [(4, 8, 16), (8, 16, 32), (16, 32, 64)]
    gen = AnchorGenerator3DS(width=W, height=W, depth=W, stride=1)
    img = torch.zeros(2, 1, 48, 40, 24, device="cuda")
    fms = [torch.zeros(2, 8, 12, 10, 6, device="cuda"), torch.zeros(2, 8, 6, 5, 3, device="cuda"), torch.zeros(2, 8, 3, 3, 3, device="cuda")]
    anc = gen(img, fms)
    assert len(anc) == 2 and anc[0] is anc[1]
    biteq(anc[0], g["anchors"], "anchors")
    assert gen.get_num_acnhors_per_level() == list(g["anchors_per_level"])
    assert gen.num_anchors_per_location() == [27, 27, 27]


def test_atss_golden(g):
    from nndetection_amd.core.boxes import ATSSMatcher
    m = ATSSMatcher(num_candidates=4, center_in_gt=False, return_match_quality=True)
    mq, matches = m(t(g["atss_gt"]), t(g["anchors"]), list(g["anchors_per_level"]), 27)
    biteq(matches, g["atss_matches"], "atss matches")
    biteq(mq, bx.box_iou(g["atss_gt"], g["anchors"]), "match quality")
    mq0, m0 = m(torch.zeros(0, 6, device="cuda"), t(g["anchors"]), list(g["anchors_per_level"]), 27)
    assert mq0.numel() == 0 and bool((m0 == -1).all())


@pytest.mark.parametrize("G,seed", [(1, 0), (7, 1), (19, 2), (40, 3)])
def test_atss_random_vs_oracle(G, seed):
    """Larger anchor set, many GTs (tile boundaries of the GT loop), incl. GT centres ON the anchor lattice
    (distance ties -> lowest-index rule)."""
    from nndetection_amd.core.boxes import ATSSMatcher
    rng = np.random.default_rng(seed)
    W = [(4, 8, 16), (8, 16, 32), (16, 32, 64)]
    anchors, npl = bx.anchors_for_image((64, 48, 40), [(16, 12, 10), (8, 6, 5), (4, 3, 3)], W, W, W)
    gt = rand_boxes(rng, G, extent=(64, 48, 40), smin=4, smax=24)
    if G > 1:
        gt[0] = [10, 6, 22, 18, 2, 14]      # centre (16, 12, 8): exactly on the level-0 lattice (stride 4)
    _, ref = bx.atss_match(gt, anchors, npl, 27, 4)
    _, got = ATSSMatcher(num_candidates=4, center_in_gt=False)(t(gt), t(anchors), npl, 27)
    biteq(got, ref, f"atss G={G}")


def _config5_boxes(rng, n, extent=160.0, smin=2.0, smax=26.0):
    """centres U(0, 160)^3, sizes U(2, 26) (the survey's probe `rb`)."""
    c = rng.uniform(0, extent, (n, 3)); s = rng.uniform(smin, smax, (n, 3))
    return np.stack([c[:, 0] - s[:, 0] / 2, c[:, 1] - s[:, 1] / 2, c[:, 0] + s[:, 0] / 2, c[:, 1] + s[:, 1] / 2,
                     c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1).astype(np.float32)


@pytest.mark.parametrize("G,levels,per_level", [(130, 2, 20000), (2000, 5, 100000)])
def test_atss_config5_size_vs_blocked_oracle(G, levels, per_level):
    """BASELINE.json configs[4] / SURVEY 8d config 5 at its STATED size: 2 000 GT boxes against 5 levels x 100 000 anchors, 27 anchors per
    location, 4 candidates (k = 108 per level) -> 125 GT tiles and 10 000 (GT, level) radix-select problems in nndet_atss3d_match_f32.
    Bit-exact against the blocked restatement of nndet/core/boxes/matcher/atss.py:48-122 (the dense oracle needs 3 x 4 GB matrices);
    `tests/test_oracle_golden.py::test_atss_blocked_equals_dense` pins the blocked form to the dense one."""
    from nndetection_amd.core.boxes import ATSSMatcher
    rng = np.random.default_rng(0)
    anchors = np.concatenate([_config5_boxes(rng, per_level) for _ in range(levels)], 0)
    gt = _config5_boxes(rng, G)
    npl = [per_level] * levels
    ref = bx.atss_match_blocked(gt, anchors, npl, 27, 4, rows=50, threads=min(16, os.cpu_count() or 1))
    _, got = ATSSMatcher(num_candidates=4, center_in_gt=False)(t(gt), t(anchors), npl, 27)
    biteq(got, ref, f"atss G={G} x {levels} x {per_level}")
    assert int((ref >= 0).sum()) > 20 * G            # the case is not degenerate: tens of positives per GT (27 / 63 per GT measured)


def test_atss_batched_matches_per_image():
    """One pass for the whole batch (nndet_atss3d_match_batched_f32) == per-image oracle matches, including images
    without objects at the start / middle / end of the batch and a batch without any object."""
    from nndetection_amd.core.boxes import ATSSMatcher
    rng = np.random.default_rng(11)
    W = [(4, 8, 16), (8, 16, 32), (16, 32, 64)]
    anchors, npl = bx.anchors_for_image((64, 48, 40), [(16, 12, 10), (8, 6, 5), (4, 3, 3)], W, W, W)
    m = ATSSMatcher(num_candidates=4, center_in_gt=False)
    for counts in [(3, 0, 17, 1), (0, 5, 0), (2, 2), (0, 0), (20, 19, 0)]:
        gts = [rand_boxes(rng, c, extent=(64, 48, 40), smin=4, smax=24) if c else np.zeros((0, 6), np.float32) for c in counts]
        gt_all, got, offs = m.match_batch([t(x) for x in gts], t(anchors), npl, 27)
        assert offs == list(np.concatenate([[0], np.cumsum(counts)]))
        assert gt_all.shape[0] == sum(counts)
        for b, gt in enumerate(gts):
            if len(gt):
                _, ref = bx.atss_match(gt, anchors, npl, 27, 4)
            else:
                ref = np.full((anchors.shape[0],), -1, np.int64)
            biteq(got[b], ref, f"batched atss counts={counts} image {b}")


def test_nms_golden(g):
    from nndetection_amd.core.boxes import nms, batched_nms
    for key in ["300_0.6", "1500_0.1", "1500_0.6"]:
        thr = float(key.split("_")[1])
        biteq(nms(t(g[f"nms_boxes_{key}"]), t(g[f"nms_scores_{key}"]), thr), g[f"nms_keep_{key}"], f"nms {key}")
    biteq(batched_nms(t(g["bnms_boxes"]), t(g["bnms_scores"]), t(g["bnms_cls"]), 0.5), g["bnms_keep"], "batched_nms")


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 1000, 4096, 4097, 8200, 10000])
@pytest.mark.parametrize("thr", [0.1, 0.6])
def test_nms_sizes_vs_oracle(n, thr):
    """chunk (64) and super-chunk (4096) boundaries; clustered boxes so that suppression chains are long."""
    from nndetection_amd.core.boxes import nms
    rng = np.random.default_rng(n)
    b = rand_boxes(rng, n, extent=(50, 50, 30), smin=6, smax=20)
    s = distinct_scores(rng, n)
    biteq(nms(t(b), t(s), thr), nms_c.nms(b, s, thr), f"nms n={n} thr={thr}")


def test_nms_edge_cases():
    from nndetection_amd.core.boxes import nms, batched_nms
    assert nms(torch.zeros(0, 6, device="cuda"), torch.zeros(0, device="cuda"), 0.5).shape == (0,)
    assert batched_nms(torch.zeros(0, 6, device="cuda"), torch.zeros(0, device="cuda"), torch.zeros(0, device="cuda"), 0.5).shape == (0,)
    same = np.tile(np.asarray([[1, 2, 9, 8, 3, 7]], np.float32), (300, 1))
    s = distinct_scores(np.random.default_rng(0), 300)
    k = nms(t(same), t(s), 0.5).cpu().numpy()
    assert list(k) == [int(np.argmax(s))]
    # disjoint boxes: everything survives, ordered by score
    far = np.stack([np.asarray([10 * i, 0, 10 * i + 5, 5, 0, 5], np.float32) for i in range(200)])
    k = nms(t(far), t(s[:200]), 0.5).cpu().numpy()
    assert np.array_equal(k, np.argsort(-s[:200], kind="stable"))
    # degenerate (zero-volume) boxes give NaN IoU against each other: never suppressed (CUDA kernel semantics)
    deg = np.tile(np.asarray([[1, 1, 1, 1, 1, 1]], np.float32), (5, 1))
    assert len(nms(t(deg), t(s[:5]), 0.5)) == 5


def test_nms_100k_properties():
    """BASELINE.json config 5 size: bit-exact against the matrix-free C oracle, plus size-independent properties."""
    from nndetection_amd.core.boxes import nms, box_iou
    rng = np.random.default_rng(5)
    n = 100_000
    b = rand_boxes(rng, n, extent=(160, 160, 160), smin=2, smax=26)
    s = distinct_scores(rng, n)
    keep = nms(t(b), t(s), 0.1)
    kc = keep.cpu().numpy()
    assert np.all(np.diff(s[kc]) < 0), "keep is not sorted by decreasing score"
    again = nms(t(b[kc]), t(s[kc]), 0.1).cpu().numpy()
    assert np.array_equal(again, np.arange(len(kc))), "NMS is not idempotent on its own output"
    sub = kc[:3000]
    iou = box_iou(t(b[sub]), t(b[sub])).cpu().numpy()
    np.fill_diagonal(iou, 0)
    assert iou.max() <= 0.1, "two kept boxes overlap more than the threshold"
    biteq(kc, nms_c.nms(b, s, 0.1), "nms 100k vs C oracle")


def test_decode_clip_and_giou_diag(g):
    from nndetection_amd.core.boxes.coder import decode_clip, decode_single
    from nndetection_amd.core.boxes import giou_diag
    from oracle.retina_torch import giou_t, decode_single_t
    # the REFERENCE's own decode / clip outputs (tests/golden/boxes_golden.npz: dec_rel incl. one delta in the exp clamp).
    # exp is a library function (<= 1 ulp apart between libm / SLEEF / ocml), so: 1e-4 absolute OR 2 ulp relative.
    def close(got, ref, what):
        err = np.abs(got - ref)
        bad = err > np.maximum(1e-4, 2.4e-7 * np.abs(ref))
        assert not bad.any(), (what, float(err.max()), int(bad.sum()))
    an = g["anchors"]
    got = decode_clip(t(g["dec_rel"]), t(an), None).cpu().numpy()
    close(got, g["dec_boxes"], "decode vs reference golden")
    gotc = decode_clip(t(g["dec_rel"]), t(an), (48, 40, 24)).cpu().numpy()
    close(gotc, g["clip_boxes"], "decode + clip vs reference golden")
    assert gotc.min() >= 0 and gotc[:, [0, 2]].max() <= 48 and gotc[:, [1, 3]].max() <= 40 and gotc[:, 4:].max() <= 24
    rng = np.random.default_rng(3)
    anchors = rand_boxes(rng, 5000)
    rel = (rng.standard_normal((2 * 5000, 6)) * 0.5).astype(np.float32)
    rel[11, 5] = 8.0
    ref = bx.decode_single(rel, np.tile(anchors, (2, 1)))
    got = decode_clip(t(rel), t(anchors), None).cpu().numpy()         # anchor row = i % n_anchor
    close(got, ref, "decode vs oracle")
    gotc = decode_clip(t(rel), t(anchors), (160, 160, 96)).cpu().numpy()
    close(gotc, bx.clip_boxes_to_image(ref, (160, 160, 96)), "decode + clip vs oracle")
    # differentiable torch decode on device == oracle decode
    d2 = decode_single(t(rel[:100]), t(anchors[:100])).cpu().numpy()
    close(d2, ref[:100], "torch decode_single on device vs oracle")
    # GIoU diag forward (bit-exact vs oracle diag) and backward (vs autograd of the reference expression)
    p = rand_boxes(rng, 64, extent=(40, 40, 40), smin=5, smax=20)
    q = rand_boxes(rng, 64, extent=(40, 40, 40), smin=5, smax=20)
    pt = t(p).requires_grad_(True)
    out = giou_diag(pt, t(q), eps=1e-7)
    biteq(out.detach(), np.diag(bx.generalized_box_iou(p, q, 1e-7)).copy(), "giou diag")
    (-out.sum()).backward()
    pc = torch.from_numpy(p).requires_grad_(True)
    (-torch.diag(giou_t(pc, torch.from_numpy(q), 1e-7)).sum()).backward()
    assert torch.allclose(pt.grad.cpu(), pc.grad, atol=1e-6, rtol=1e-4), (pt.grad.cpu() - pc.grad).abs().max()


def test_sigmoid_max():
    from nndetection_amd import _lib as L
    x = torch.randn(10001, 3, device="cuda")
    out = torch.empty(10001, device="cuda")
    L.call("nndet_sigmoid_max_f32", L.ptr(x), 10001, 3, L.ptr(out), L.stream())
    assert torch.allclose(out, torch.sigmoid(x).max(1)[0], atol=1e-6)


def test_atss_center_in_gt_matches_reference(golden_dir):
    """ATSSMatcher(center_in_gt=True) -- the reference's default (nndet/core/boxes/matcher/atss.py:101-107; RetinaUNetV001 switches it
    off, nndet/conf/train/v001.yaml:107): a candidate only becomes a positive if the anchor's centre lies inside the GT box, more than
    0.01 from every face. Bit-exact against the reference fixture (GT boxes whose candidates partly / all fail the test) and the
    oracle, single-image and batched entry, with labels."""
    from nndetection_amd.core.boxes import ATSSMatcher
    g = np.load(os.path.join(golden_dir, "boxes_golden.npz"))
    gt, anchors, npl = g["atss_gt_center_in_gt"], g["anchors"], [int(v) for v in g["anchors_per_level"]]
    ref = g["atss_matches_center_in_gt"]
    m = ATSSMatcher(num_candidates=4, center_in_gt=True)
    _, got = m(t(gt), t(anchors), npl, 27)
    assert np.array_equal(got.cpu().numpy(), ref)
    _, plain = ATSSMatcher(num_candidates=4, center_in_gt=False)(t(gt), t(anchors), npl, 27)
    assert (plain.cpu().numpy() >= 0).sum() > (ref >= 0).sum() > 0
    # batched (two images: the fixture's boxes split 4 / 3), with and without labels
    boxes = [t(gt[:4]), t(gt[4:])]
    classes = [torch.arange(4, dtype=torch.float32).cuda(), torch.arange(3, dtype=torch.float32).cuda() + 1]
    gt_all, mb, offs, labels = m.match_batch(boxes, t(anchors), npl, 27, classes=classes)
    _, mb2, _ = m.match_batch(boxes, t(anchors), npl, 27)
    assert torch.equal(mb, mb2)
    for i, (b, c) in enumerate(zip(boxes, classes)):
        _, want = bx.atss_match(b.cpu().numpy(), anchors, npl, 27, 4, center_in_gt=True)
        assert np.array_equal(mb[i].cpu().numpy(), want), i
        lab = np.where(want >= 0, c.cpu().numpy()[np.clip(want, 0, None)] + 1, 0).astype(np.float32)
        assert np.array_equal(labels[i].cpu().numpy(), lab), i


def test_nms_2d_matches_reference(golden_dir):
    """nndet._C.nms accepts [N, 4] boxes as well (nms_kernel / devIoU, nndet/csrc/cuda/nms.cu:22-34,54-96; the Python wrapper sends 2D
    boxes to torchvision.ops.nms, nms.py:70-72). nndet_nms2d_f32 against the reference's `nms_cpu` on 2D boxes (fixture) and the oracle,
    bit-exact keep lists; plus a random case with suppression chains and the empty input."""
    from nndetection_amd.core.boxes import nms
    g = np.load(os.path.join(golden_dir, "boxes_golden.npz"))
    b, s = g["nms2d_boxes"], g["nms2d_scores"]
    for thr in (0.1, 0.5):
        got = nms(t(b), t(s), thr).cpu().numpy()
        assert np.array_equal(got, g[f"nms2d_keep_{thr}"]), thr
    rng = np.random.default_rng(5)
    c = rng.uniform(0, 200, (5000, 2)); sz = rng.uniform(3, 40, (5000, 2))
    bb = np.stack([c[:, 0] - sz[:, 0] / 2, c[:, 1] - sz[:, 1] / 2, c[:, 0] + sz[:, 0] / 2, c[:, 1] + sz[:, 1] / 2], 1).astype(np.float32)
    bb[7] = bb[3]                                       # an identical pair and a zero-area box
    bb[9, 2] = bb[9, 0]
    sc = distinct_scores(rng, 5000)
    assert np.array_equal(nms(t(bb), t(sc), 0.3).cpu().numpy(), bx.nms2d(bb, sc, 0.3))
    assert nms(t(bb[:0]), t(sc[:0]), 0.3).numel() == 0


@pytest.mark.parametrize("cfg", [(0.3, 0.5, False), (0.4, 0.6, True)])
def test_iou_matcher_matches_reference(golden_dir, cfg):
    """IoUMatcher (nndet/core/boxes/matcher/iou.py:20-107) on the reference fixture (bit-exact incl. the -1 / -2 sentinels and the
    low-quality rule), against the oracle on 20 000 anchors x 37 GT boxes, and the no-GT fast path."""
    from nndetection_amd.core.boxes import IoUMatcher
    low, high, lq = cfg
    g = np.load(os.path.join(golden_dir, "boxes_golden.npz"))
    gt, anchors = g["atss_gt_center_in_gt"], g["anchors"]
    m = IoUMatcher(low_threshold=low, high_threshold=high, allow_low_quality_matches=lq)
    _, got = m(t(gt), t(anchors), None, None)
    assert np.array_equal(got.cpu().numpy(), g[f"ioumatch_{low}_{high}_{int(lq)}"])
    rng = np.random.default_rng(3)
    gt2, an2 = rand_boxes(rng, 37, smin=6, smax=30), rand_boxes(rng, 20000, smin=4, smax=34)
    _, got2 = m(t(gt2), t(an2), None, None)
    _, want2 = bx.iou_match(gt2, an2, low, high, lq)
    assert np.array_equal(got2.cpu().numpy(), want2)
    assert (want2 >= 0).sum() > 0 and (want2 == -2).sum() > 0
    mq0, m0 = m(t(gt2[:0]), t(an2), None, None)
    assert mq0.numel() == 0 and bool((m0 == -1).all())
