"""Model-level NMS and weighted box clustering on the GPU. Same names, signatures and return values as
nndet/inference/detection/model.py:25-86 and nndet/inference/detection/wbc.py:22-160 (+ the ensemble wrapper of
nndet/inference/detection/ensemble.py), on top of csrc/nms3d.hip and csrc/wbc3d.hip."""
from typing import Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ..core.boxes import batched_nms


def batched_nms_model(boxes: Tensor, scores: Tensor, labels: Tensor, weights: Tensor, iou_thresh: float, *args, **kwargs):
    """model.py:25-54: batched NMS; returns the kept boxes / scores / labels / weights in descending score order."""
    keep = batched_nms(boxes, scores, labels, iou_thresh)
    return boxes[keep], scores[keep], labels[keep], weights[keep]


def batched_weighted_nms_model(boxes: Tensor, scores: Tensor, labels: Tensor, weights: Tensor, iou_thresh: float, *args, **kwargs):
    """model.py:57-86: NMS on scores * weights; the surviving boxes keep their scores, weights become 1."""
    keep = batched_nms(boxes, scores * weights, labels, iou_thresh)
    return boxes[keep], scores[keep], labels[keep], torch.ones_like(weights)[keep]


def _wbc_raw(boxes, scores, labels, weights, n_exp_preds, iou_thresh, score_thresh, use_area, missing_weight):
    n = boxes.shape[0]
    if boxes.shape[1] != 6:
        raise L.NndetError("only 3D boxes are supported by the MI355X weighted box clustering")
    b = boxes.detach().float().contiguous()
    s = scores.detach().float().contiguous()
    w = weights.detach().float().contiguous().to(b.device)
    e = n_exp_preds.detach().float().contiguous().to(b.device)
    lab = labels.detach().to(b.device, torch.int64).contiguous() if labels is not None else None
    dev = b.device
    ob = torch.empty((n, 6), dtype=torch.float32, device=dev)
    os_ = torch.empty((n,), dtype=torch.float32, device=dev)
    ol = torch.empty((n,), dtype=torch.int64, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)
    ws_bytes = L.load().nndet_wbc3d_workspace_bytes(n)
    if ws_bytes == 0:
        raise L.NndetError("nndet_wbc3d_workspace_bytes failed")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    L.call("nndet_wbc3d_f32", L.ptr(b), L.ptr(s), L.ptr(lab), L.ptr(w), L.ptr(e), n, float(iou_thresh), float(score_thresh),
           int(bool(use_area)), float(missing_weight), L.ptr(ob), L.ptr(os_), L.ptr(ol), L.ptr(cnt), L.ptr(ws), ws_bytes, L.stream())
    k = int(cnt.item())
    return ob[:k], os_[:k], ol[:k]


def wbc(boxes: Tensor, scores: Tensor, weights: Tensor, n_exp_preds: Tensor, iou_thresh: float, score_thresh: float,
        use_area: bool = True, missing_weight: float = 1.) -> Tuple[Tensor, Tensor]:
    """Weighted box clustering of ONE class (wbc.py:94-160): -> (consolidated boxes [K, 6], consolidated scores [K]) in cluster
    order (descending score of the cluster's best box)."""
    if boxes.shape[0] == 0:
        return torch.tensor([]).view(-1, boxes.shape[1]).to(boxes), torch.tensor([]).view(-1).to(scores)
    b, s, _ = _wbc_raw(boxes, scores, None, weights, n_exp_preds, iou_thresh, score_thresh, use_area, missing_weight)
    return b.to(boxes.dtype), s.to(scores.dtype)


def batched_wbc(boxes: Tensor, scores: Tensor, labels: Tensor, weights: Tensor, iou_thresh: float, n_exp_preds: Tensor,
                score_thresh: float, use_area: bool = False, missing_weight: float = 1.) -> Tuple[Tensor, Tensor, Tensor]:
    """Weighted box clustering per class (wbc.py:22-91) in ONE pass (the IoU bit is masked by label equality): -> boxes, scores,
    labels grouped by ascending label like the reference's loop over labels.unique(), cluster order inside a label."""
    if boxes.shape[0] == 0:
        return (torch.tensor([]).view(-1, boxes.shape[1]), torch.tensor([]).view(-1), torch.tensor([]).view(-1))
    b, s, l = _wbc_raw(boxes, scores, labels, weights, n_exp_preds, iou_thresh, score_thresh, use_area, missing_weight)
    order = torch.sort(l, stable=True)[1]
    return b[order].to(boxes.dtype), s[order].to(scores.dtype), l[order].to(scores.dtype)


def batched_wbc_ensemble(boxes: Tensor, scores: Tensor, labels: Tensor, weights: Tensor, iou_thresh: float, n_exp_preds: Tensor,
                         score_thresh: float, *args, **kwargs):
    """ensemble_nms_fn signature of the ensemblers (nndet/inference/detection/ensemble.py): + unit weights for the result."""
    b, s, l = batched_wbc(boxes, scores, labels, weights, iou_thresh, n_exp_preds, score_thresh, **kwargs)
    return b, s, l, torch.ones_like(s)
