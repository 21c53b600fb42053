"""CPU oracle for the DIFFERENTIABLE box ops of the RetinaUNet hot path (torch on the CPU, any float dtype).

TEST INFRASTRUCTURE ONLY: imported by `tests/` and `tests/golden/make_golden.py` as the *checker*. The product
(`nndetection_amd/`) never imports this module and has no CPU fallback.

Restates the reference's autograd expressions so that tests can differentiate them in float64:
`generalized_box_iou` = nndet/core/boxes/ops.py:131-159 (box_iou_union_3d) + 162-185 (generalized_box_iou_3d), `giou_loss` =
nndet/losses/regression.py:147-162 (GIoULoss.forward: diag of the [N, N] matrix, reduction, weight * -1). Unlike the reference entry
point (ops.py:106-128) the dtype is kept (the reference casts to fp32), which is what the float64 arbitration needs. Pinned against
the real reference by `tests/golden/make_golden.py giou_grad` (fixture `tests/golden/giou_grad_golden.npz`).
"""
import torch
from torch import Tensor


def _vol(b: Tensor) -> Tensor:
    """nndet/core/boxes/ops.py:60-72 (box_area_3d)"""
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) * (b[:, 5] - b[:, 4])


def box_iou_union(b1: Tensor, b2: Tensor, eps: float = 0.0):
    """nndet/core/boxes/ops.py:131-159"""
    vol1, vol2 = _vol(b1), _vol(b2)
    x1 = torch.max(b1[:, None, 0], b2[:, 0])
    y1 = torch.max(b1[:, None, 1], b2[:, 1])
    x2 = torch.min(b1[:, None, 2], b2[:, 2])
    y2 = torch.min(b1[:, None, 3], b2[:, 3])
    z1 = torch.max(b1[:, None, 4], b2[:, 4])
    z2 = torch.min(b1[:, None, 5], b2[:, 5])
    inter = ((x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0) * (z2 - z1).clamp(min=0)) + eps
    union = vol1[:, None] + vol2 - inter
    return inter / union, union


def generalized_box_iou(b1: Tensor, b2: Tensor, eps: float = 0.0) -> Tensor:
    """nndet/core/boxes/ops.py:162-185; eps is NOT forwarded to the inner IoU (ops.py:175)"""
    iou, union = box_iou_union(b1, b2)
    x1 = torch.min(b1[:, None, 0], b2[:, 0])
    y1 = torch.min(b1[:, None, 1], b2[:, 1])
    x2 = torch.max(b1[:, None, 2], b2[:, 2])
    y2 = torch.max(b1[:, None, 3], b2[:, 3])
    z1 = torch.min(b1[:, None, 4], b2[:, 4])
    z2 = torch.max(b1[:, None, 5], b2[:, 5])
    vol = ((x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0) * (z2 - z1).clamp(min=0)) + eps
    return iou - (vol - union) / vol


def giou_loss(pred: Tensor, target: Tensor, eps: float = 1e-7, reduction: str = "sum", loss_weight: float = 1.0) -> Tensor:
    """nndet/losses/regression.py:147-162 (+ nndet/losses/base.py reduction_helper: None / 'mean' / 'sum')"""
    d = torch.diag(generalized_box_iou(pred, target, eps=eps), diagonal=0)
    if reduction == "mean":
        d = d.mean()
    elif reduction == "sum":
        d = d.sum()
    return loss_weight * -1 * d
