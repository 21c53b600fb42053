"""GPU parity (bit-exact, integer work) of the device-side target preparation (csrc/targets.hip) against the
reference-generated fixture and the numpy oracle."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from tests.gpu_util import t

pytestmark = pytest.mark.gpu


def test_targets_vs_reference_golden(golden_dir):
    from nndetection_amd.core.targets import instances_to_targets, prepare_targets
    g = np.load(os.path.join(golden_dir, "targets_golden.npz"))
    tgt = g["target"].astype(np.float32)
    maps = [ast.literal_eval(str(m)) for m in g["maps"]]
    boxes, classes, ids, seg = instances_to_targets(t(tgt), maps)
    assert seg.shape == tgt.shape and seg.dtype == torch.float32
    for b in range(tgt.shape[0]):
        assert np.array_equal(boxes[b].cpu().numpy(), g[f"boxes_{b}"]), b
        assert np.array_equal(classes[b].cpu().numpy(), g[f"classes_{b}"]) and classes[b].dtype == torch.int64
        assert np.array_equal(ids[b].cpu().numpy(), g[f"ids_{b}"])
        assert np.array_equal(seg[b, 0].cpu().numpy().astype(np.uint8), g[f"seg_{b}"])
    images, targets = prepare_targets(torch.zeros(3, 1, 24, 20, 16, device="cuda"), t(tgt), maps)
    assert targets["target_seg"].shape == (3, 24, 20, 16) and len(targets["target_boxes"]) == 3
    assert targets["target_boxes"][1].shape == (0, 6)


def test_targets_full_patch_random_vs_oracle():
    """160x160x96 patches (the benchmarked shape), many instances, ids above the LDS fast path (>= 256), touching the border."""
    from nndetection_amd.core.targets import instances_to_targets
    rng = np.random.default_rng(3)
    B, D, H, W = 2, 160, 160, 96
    tgt = np.zeros((B, 1, D, H, W), np.float32)
    maps = []
    for b in range(B):
        m = {}
        for i in list(rng.choice(np.arange(1, 200), 12, replace=False)) + [300, 511, 700]:
            lo = rng.integers(0, [D - 4, H - 4, W - 4]); sz = rng.integers(1, 30, 3)
            hi = np.minimum(lo + sz, [D, H, W])
            tgt[b, 0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = i            # later blobs may overwrite earlier ones
            m[int(i)] = int(rng.integers(0, 3))
        tgt[b, 0, 0, 0, 0] = 1; m[1] = 2
        tgt[b, 0, D - 1, H - 1, W - 1] = 2; m[2] = 1
        m[150_000 % 997] = 0                                                 # a mapping entry whose instance is not in the patch
        maps.append(m)
    boxes, classes, ids, seg = instances_to_targets(t(tgt), maps)
    for b in range(B):
        ob, oc, oi, osem = bx.instances_to_targets(tgt[b, 0], maps[b])
        assert np.array_equal(boxes[b].cpu().numpy(), ob) and np.array_equal(classes[b].cpu().numpy(), oc)
        assert np.array_equal(ids[b].cpu().numpy(), oi)
        assert np.array_equal(seg[b, 0].cpu().numpy(), osem)


def test_targets_errors_and_empty():
    from nndetection_amd.core.targets import instances_to_targets
    tgt = torch.zeros(1, 1, 8, 8, 8, device="cuda")
    boxes, classes, ids, seg = instances_to_targets(tgt, [{}])
    assert boxes[0].shape == (0, 6) and classes[0].shape == (0,) and float(seg.abs().sum()) == 0.0
    tgt[0, 0, 1, 2, 3] = 4
    with pytest.raises(KeyError):
        instances_to_targets(tgt, [{1: 0}])              # id 4 > every key
    with pytest.raises(KeyError):
        instances_to_targets(tgt, [{1: 0, 9: 0}])        # id 4 missing from the mapping
