"""GPU: planner-side consumers: the anchor-search objective against the numpy oracle, and the ROCm-safe memory
estimator on the `tiny` plan."""
import numpy as np
import pytest
import torch

from oracle import boxes_np as bx
from tests.gpu_util import rand_boxes, t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,A", [(1, 1), (300, 27), (5000, 108), (257, 1000)])
def test_anchor_fit_objective_vs_oracle(G, A):
    from nndetection_amd.planning import iou_rowmax, anchor_fit, AnchorFitObjective
    rng = np.random.default_rng(G + A)
    gt = rand_boxes(rng, G, extent=(60, 60, 40), smin=4, smax=30)
    half = rng.uniform(2, 20, (A, 3)).astype(np.float32)
    anchors = np.stack([-half[:, 0], -half[:, 1], half[:, 0], half[:, 1], -half[:, 2], half[:, 2]], 1) + 30.0
    ref = bx.box_iou(gt, anchors.astype(np.float32)).max(1)
    got = iou_rowmax(t(gt), t(anchors.astype(np.float32))).cpu().numpy()
    assert np.array_equal(got, ref)                          # max of bit-exact IoU values
    assert abs(float(anchor_fit(t(gt), t(anchors.astype(np.float32)))) - float(ref.astype(np.float64).mean())) < 1e-6
    assert abs(AnchorFitObjective(torch.from_numpy(gt))(torch.from_numpy(anchors.astype(np.float32))) - float(ref.mean())) < 1e-6


def test_memory_estimator_rocm():
    from nndetection_amd.planning import MemoryEstimatorDetection, smi_memory_allocated, num_gpus, ARCHS
    from nndetection_amd.plans import get_plan
    from nndetection_amd.ptmodule import build_model
    assert num_gpus() >= 1 and smi_memory_allocated(0) > 0
    plan = get_plan("tiny")
    est = MemoryEstimatorDetection(batch_size=2)
    mem, fits = est.estimate(min_shape=plan["patch_size"], target_shape=plan["patch_size"], network=build_model(plan),
                             optimizer_cls=lambda p: torch.optim.SGD(p, 0.01, momentum=0.9), in_channels=1, num_instances=2)
    assert fits and 0 < mem < ARCHS["MI355X"], (mem, getattr(est, "last_error", None))
    small = MemoryEstimatorDetection(target_mem=1024, batch_size=2)            # a 1 KiB "card": cannot fit
    mem2, fits2 = small.estimate(plan["patch_size"], plan["patch_size"], build_model(plan),
                                 optimizer_cls=lambda p: torch.optim.SGD(p, 0.01), in_channels=1)
    assert not fits2 and mem2 > 1024
