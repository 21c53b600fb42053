"""Pairwise 3D IoU / GIoU on the GPU (HIP kernels in csrc/boxes.hip).

Same call signatures and semantics as nndet/core/boxes/ops.py:75-185 (3D branch): fp32, autocast
disabled by construction, `tensor([])` for empty inputs, `eps` only added to the intersection (IoU)
resp. the hull volume (GIoU).
"""
import torch
from torch import Tensor

from ... import _lib as L


def _f32c(t: Tensor) -> Tensor:
    return t.detach().float().contiguous()


def _pairwise(name: str, boxes1: Tensor, boxes2: Tensor, eps: float) -> Tensor:
    if boxes1.numel() == 0 or boxes2.numel() == 0:
        return torch.tensor([]).to(boxes1)          # ops.py:96-97
    if boxes1.shape[-1] != 6 or boxes2.shape[-1] != 6:
        raise L.NndetError("only 3D boxes (x1, y1, x2, y2, z1, z2) are on the MI355X hot path")
    a, b = _f32c(boxes1), _f32c(boxes2)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.call(name, L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], float(eps), L.ptr(out), L.stream())
    return out


def box_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 0) -> Tensor:
    """[N,6] x [M,6] -> [N,M] fp32 IoU (nndet/core/boxes/ops.py:75-102)."""
    return _pairwise("nndet_iou3d_pairwise_f32", boxes1, boxes2, eps)


class _GIoUPairwise(torch.autograd.Function):
    """generalized_box_iou is differentiable w.r.t. both box sets, like the reference's autograd expression
    (nndet/core/boxes/ops.py:106-128,162-185; GIoULoss back-propagates through it, nndet/losses/regression.py:158-161):
    forward = the pairwise kernel, backward = `nndet_giou3d_pairwise_bwd_f32` (row sums of the analytic gradient)."""

    @staticmethod
    def forward(ctx, b1, b2, eps):
        out = _pairwise("nndet_giou3d_pairwise_f32", b1, b2, eps)
        if out.numel():
            ctx.save_for_backward(_f32c(b1), _f32c(b2))
        ctx.eps = float(eps)
        ctx.in_dtypes = (b1.dtype, b2.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.saved_tensors:                      # empty input: `tensor([])`, nothing to propagate
            return None, None, None
        a, b = ctx.saved_tensors
        g = g.float().contiguous()
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        L.call("nndet_giou3d_pairwise_bwd_f32", L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(g), ctx.eps,
               L.ptr(ga) if ga is not None else None, L.ptr(gb) if gb is not None else None, L.stream())
        return (ga.to(ctx.in_dtypes[0]) if ga is not None else None, gb.to(ctx.in_dtypes[1]) if gb is not None else None, None)


def generalized_box_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 0) -> Tensor:
    """[N,6] x [M,6] -> [N,M] fp32 GIoU (nndet/core/boxes/ops.py:106-128,162-185)."""
    if boxes1.requires_grad or boxes2.requires_grad:
        return _GIoUPairwise.apply(boxes1, boxes2, eps)
    return _pairwise("nndet_giou3d_pairwise_f32", boxes1, boxes2, eps)


class _GIoUDiag(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, eps):
        a, b = _f32c(pred), _f32c(target)
        out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
        L.call("nndet_giou3d_diag_fwd_f32", L.ptr(a), L.ptr(b), a.shape[0], float(eps), L.ptr(out), L.stream())
        ctx.save_for_backward(a, b)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.float().contiguous()
        ga = torch.empty_like(a)
        L.call("nndet_giou3d_diag_bwd_f32", L.ptr(a), L.ptr(b), L.ptr(g), a.shape[0], ctx.eps, L.ptr(ga), L.stream())
        return ga, None, None


def giou_diag(pred_boxes: Tensor, target_boxes: Tensor, eps: float = 0) -> Tensor:
    """diag(generalized_box_iou(pred, target, eps)) with gradient w.r.t. `pred` -- exactly what GIoULoss
    consumes (nndet/losses/regression.py:147-162), O(P) instead of O(P^2)."""
    if pred_boxes.shape[0] == 0:
        return pred_boxes.new_zeros((0,), dtype=torch.float32)
    return _GIoUDiag.apply(pred_boxes, target_boxes, eps)


def remove_small_boxes(boxes: Tensor, min_size: float) -> Tensor:
    """nndet/core/boxes/ops.py:241-259 (3D)."""
    ws, hs, ds = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1], boxes[:, 5] - boxes[:, 4]
    return torch.where((ws >= min_size) & (hs >= min_size) & (ds >= min_size))[0]


def box_center(boxes: Tensor) -> Tensor:
    """nndet/core/boxes/ops.py:314-327 (3D)."""
    return torch.stack([(boxes[:, 2] + boxes[:, 0]) / 2., (boxes[:, 3] + boxes[:, 1]) / 2., (boxes[:, 5] + boxes[:, 4]) / 2.], dim=1)
