"""GPU parity of the weighted box clustering (csrc/wbc3d.hip) against the reference-generated fixture and the
numpy oracle. The cluster STRUCTURE (which predictions merge, how many clusters, their labels / order) must be identical;
consolidated values are fp32 sums whose order is unspecified in the reference: 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from tests.gpu_util import rand_boxes, t

pytestmark = pytest.mark.gpu

CASES = (("a", dict(iou_thresh=0.3, score_thresh=0.0, use_area=False, missing_weight=1.0)),
         ("b", dict(iou_thresh=0.1, score_thresh=0.2, use_area=True, missing_weight=0.5)))


def test_wbc_vs_reference_golden(golden_dir):
    from nndetection_amd.inference import batched_wbc, wbc
    g = np.load(os.path.join(golden_dir, "wbc_golden.npz"))
    for tag, kw in CASES:
        b, s, l = batched_wbc(t(g["boxes"]), t(g["scores"]), t(g["labels"]), t(g["weights"]), kw["iou_thresh"], t(g["n_exp"]),
                              kw["score_thresh"], use_area=kw["use_area"], missing_weight=kw["missing_weight"])
        assert b.shape == g[f"out_boxes_{tag}"].shape, (tag, b.shape)
        assert np.array_equal(l.cpu().numpy(), g[f"out_labels_{tag}"])
        assert np.allclose(b.cpu().numpy(), g[f"out_boxes_{tag}"], rtol=1e-5, atol=1e-4)
        assert np.allclose(s.cpu().numpy(), g[f"out_scores_{tag}"], rtol=1e-5, atol=1e-6)
    m = g["labels"] == 1
    b, s = wbc(t(g["boxes"][m]), t(g["scores"][m]), t(g["weights"][m]), t(g["n_exp"][m]), 0.2, 0.1)
    assert np.allclose(b.cpu().numpy(), g["one_boxes"], rtol=1e-5, atol=1e-4) and np.allclose(s.cpu().numpy(), g["one_scores"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,seed", [(1, 0), (65, 1), (3000, 2), (9000, 3)])
def test_wbc_random_vs_oracle(n, seed):
    """Many overlapping predictions (clusters of up to ~40 members), chunk / super-chunk boundaries of the head scan, score
    ties, a zero-volume box (NaN IoU with itself: it disappears, wbc.py:122,141)."""
    from nndetection_amd.inference import batched_wbc
    rng = np.random.default_rng(seed)
    b = rand_boxes(rng, n, extent=(60, 60, 40), smin=6, smax=20)
    s = rng.uniform(0.01, 1.0, n).astype(np.float32)
    if n > 10:
        s[5] = s[4]
        b[7] = [3, 3, 3, 9, 2, 8]
    l = rng.integers(0, 2, n)
    w = rng.uniform(0.2, 1.0, n).astype(np.float32)
    ne = rng.integers(1, 6, n).astype(np.float32)
    gb, gs, gl = batched_wbc(t(b), t(s), t(l), t(w), 0.25, t(ne), 0.05, use_area=True, missing_weight=0.7)
    ob, os_, ol = bx.batched_wbc(b, s, l, w, 0.25, ne, 0.05, True, 0.7)
    assert gb.shape == ob.shape, (gb.shape, ob.shape)
    assert np.array_equal(gl.cpu().numpy(), ol)
    assert np.allclose(gs.cpu().numpy(), os_, rtol=2e-5, atol=1e-6)
    assert np.allclose(gb.cpu().numpy(), ob, rtol=2e-5, atol=2e-4)


def test_model_nms_functions_and_empty():
    from nndetection_amd.inference import batched_nms_model, batched_weighted_nms_model, batched_wbc
    rng = np.random.default_rng(4)
    b = rand_boxes(rng, 500, extent=(40, 40, 40), smin=5, smax=15)
    s = rng.uniform(0, 1, 500).astype(np.float32); w = rng.uniform(0.5, 1, 500).astype(np.float32); l = rng.integers(0, 3, 500)
    kb, ks, kl, kw = batched_nms_model(t(b), t(s), t(l), t(w), 0.4)
    keep = bx.batched_nms(b, s, l, 0.4)
    assert np.array_equal(kb.cpu().numpy(), b[keep]) and np.array_equal(kw.cpu().numpy(), w[keep])
    kb, ks, kl, kw = batched_weighted_nms_model(t(b), t(s), t(l), t(w), 0.4)
    keep = bx.batched_nms(b, s * w, l, 0.4)
    assert np.array_equal(ks.cpu().numpy(), s[keep]) and float(kw.min()) == 1.0
    e = batched_wbc(torch.zeros(0, 6, device="cuda"), torch.zeros(0, device="cuda"), torch.zeros(0, device="cuda"),
                    torch.zeros(0, device="cuda"), 0.3, torch.zeros(0, device="cuda"), 0.0)
    assert e[0].shape == (0, 6) and e[1].shape == (0,)
