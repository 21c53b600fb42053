"""GPU parity of the device-side hard-negative sampler (csrc/sampler.hip) against the numpy oracle (deterministic mode =
the reference under randperm := reversed arange) and distribution-free properties of the random mode."""
import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from tests.gpu_util import t

pytestmark = pytest.mark.gpu


def _case(rng, N, C, n_pos, n_ign=50):
    labels = np.zeros(N, np.float32)
    idx = rng.permutation(N)
    labels[idx[:n_pos]] = rng.integers(1, 3, n_pos)
    labels[idx[n_pos:n_pos + n_ign]] = -1
    logits = (rng.standard_normal((N, C)) * 2 - 2).astype(np.float32)
    logits[idx[-300:]] = logits[idx[-1]]                       # probability ties inside the pool candidates
    return labels, logits


@pytest.mark.parametrize("hist_wgs", ["0", "3"], ids=["block_per_wg", "3_wgs_walk_blocks"])
@pytest.mark.parametrize("N,C,B,n_pos", [(200_000, 1, 4, 500), (200_000, 3, 4, 7), (1_186_650 * 2, 1, 2, 60), (5000, 1, 1, 0), (3000, 2, 64, 900)])
def test_sampler_deterministic_vs_oracle(N, C, B, n_pos, hist_wgs, monkeypatch):
    from nndetection_amd.core.boxes import HardNegativeSamplerBatched
    monkeypatch.setenv("NNDET_SP_HIST_WGS", hist_wgs)      # (round 6: a histogram workgroup may walk several blocks of anchors, labels / scores requested together)
    rng = np.random.default_rng(N + C + n_pos)
    labels, logits = _case(rng, N, C, n_pos)
    probs = bx.sigmoid(logits.max(1))
    s = HardNegativeSamplerBatched(32, 0.33, min_neg=1, pool_size=20)
    s.deterministic = True
    pos, neg = s.sample_indices(t(labels), t(logits), B)
    rp, rn, _ = bx.hnm_select_reversed(labels, probs, B)
    assert np.array_equal(pos.cpu().numpy(), rp), (len(pos), len(rp))
    got_n = neg.cpu().numpy()
    if not np.array_equal(got_n, rn):
        # sigmoid is a library function: an ulp can reorder two pool members with almost equal probability; then the SETS of
        # probabilities must still agree
        assert len(got_n) == len(rn) and np.allclose(np.sort(probs[got_n]), np.sort(probs[rn]), atol=1e-6)


def test_sampler_random_mode_properties_and_mask_contract():
    from nndetection_amd.core.boxes import HardNegativeSamplerBatched
    rng = np.random.default_rng(0)
    labels, logits = _case(rng, 300_000, 1, 400)
    probs = bx.sigmoid(logits.max(1))
    s = HardNegativeSamplerBatched(32, 0.33, min_neg=1, pool_size=20)
    B = 4
    n_pos_exp, n_neg_exp, pool_exp = bx.hnm_counts(int((labels >= 1).sum()), int((labels == 0).sum()), B)
    _, _, pool_idx = bx.hnm_select_reversed(labels, probs, B)
    pool_thr = np.sort(probs[pool_idx])[0]
    seen = []
    for seed in (1, 1, 2):
        torch.manual_seed(seed)
        pos, neg = s.sample_indices(t(labels), t(logits), B)
        pos, neg = pos.cpu().numpy(), neg.cpu().numpy()
        assert len(pos) == n_pos_exp and len(neg) == n_neg_exp
        assert np.all(np.diff(pos) > 0) and np.all(np.diff(neg) > 0)          # ascending, unique
        assert np.all(labels[pos] >= 1) and np.all(labels[neg] == 0)
        assert np.all(probs[neg] >= pool_thr - 1e-6)                          # negatives come from the top-`pool` scores
        seen.append((pos, neg))
    assert np.array_equal(seen[0][0], seen[1][0]) and np.array_equal(seen[0][1], seen[1][1])   # same torch seed -> same sample
    assert not np.array_equal(seen[0][0], seen[2][0]) and not np.array_equal(seen[0][1], seen[2][1])
    # uniformity smoke test: over 200 draws every positive is picked with roughly k/n frequency
    hits = np.zeros(labels.shape[0])
    for seed in range(200):
        torch.manual_seed(1000 + seed)
        pos, _ = s.sample_indices(t(labels), t(logits), B)
        hits[pos.cpu().numpy()] += 1
    f = hits[labels >= 1] / 200.0
    assert abs(f.mean() - n_pos_exp / 400.0) < 1e-9 and f.max() < 0.25 and f.min() > 0.01   # expected 0.105 each
    # the reference's public contract: masks for the whole batch
    torch.manual_seed(3)
    pm, nm = s([t(labels[:150_000]), t(labels[150_000:])], t(probs))
    p2, n2, _ = bx.hnm_counts(int((labels >= 1).sum()), int((labels == 0).sum()), 2)         # two images in this call
    assert pm[0].dtype == torch.uint8 and int(pm[0].sum()) == p2 and int(nm[0].sum()) == n2


@pytest.mark.parametrize("N,C,B,n_pos,det", [(200_000, 1, 4, 500, True), (1_186_650 * 4, 1, 4, 12, False), (5000, 1, 1, 0, True),
                                             (3000, 2, 64, 900, False), (300_000, 1, 8, 3, False)])
def test_sampler_tail_in_one_workgroup_equals_the_radix_sort_tail(N, C, B, n_pos, det, monkeypatch):
    """Round 5: the sampler's tail (three sorts of <= pool-capacity keys, the pool-position keys, the index resolution, the emit) as ONE
    workgroup with an LDS bitonic network (k_sp_tail) against the rocPRIM radix-sort tail of rounds 2-4 (NNDET_SP_TAIL_FUSED=0): the same
    positives, negatives and counts, in deterministic (reversed permutation) and hashed mode, incl. no positive at all, more positives
    than the budget, and the benchmark's 4.7 M anchors."""
    from nndetection_amd.core.boxes import HardNegativeSamplerBatched
    rng = np.random.default_rng(N + C + n_pos)
    labels, logits = _case(rng, N, C, n_pos)
    s = HardNegativeSamplerBatched(32, 0.33, min_neg=1, pool_size=20)
    s.deterministic = det
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("NNDET_SP_TAIL_FUSED", fused)
        torch.manual_seed(7)
        pos, neg, counts = s.sample_device(t(labels), t(logits), B)
        out[fused] = (pos.cpu().numpy(), neg.cpu().numpy(), counts.cpu().numpy())
    for a, b, what in zip(out["1"], out["0"], ("positives", "negatives", "counts")):
        assert np.array_equal(a, b), what
    assert int((out["1"][0] >= 0).sum()) == int(out["1"][2][0]) and int((out["1"][1] >= 0).sum()) == int(out["1"][2][1])
