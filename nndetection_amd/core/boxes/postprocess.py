"""Fused post-processing front end (csrc/postproc.hip): top-k pre-selection on the logits, decode + clip of the survivors
only, score / small-box filters, batched 3D NMS and the final top `detections_per_img` -- for the whole batch, without a
host round trip until the per-image counts are read. Replaces the reference's decode-everything + full sort
(nndet/arch/heads/comb.py:140-158, nndet/core/retina.py:292-379, nndet/core/boxes/nms.py:81-106)."""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from ... import _lib as L
from .coder import BBOX_XFORM_CLIP


def postprocess_batch_raw(scores: Tensor, boxes_or_deltas: Tensor, anchors: Optional[Tensor], image_shape: Optional[Sequence[int]],
                          num_classes: int, topk_candidates: Optional[int], score_thresh: Optional[float],
                          remove_small_boxes: Optional[float], nms_thresh: float, detections_per_img: Optional[int],
                          scores_are_probs: bool = False, bbox_xform_clip: float = BBOX_XFORM_CLIP):
    """Asynchronous part. scores [B, M, C] (logits, or probabilities with scores_are_probs), boxes_or_deltas [B, M, 6]
    (deltas when `anchors` [M, 6] is given, else decoded boxes). Returns padded (boxes [B, D, 6], scores [B, D],
    labels [B, D] int64, counts [B] int64) on the device; rows >= counts[b] are padding."""
    B, M, C = scores.shape
    if C != num_classes:
        raise L.NndetError(f"scores have {C} classes, expected {num_classes}")
    s = scores.detach().float().contiguous()
    d = boxes_or_deltas.detach().float().contiguous()
    if tuple(d.shape) != (B, M, 6):
        raise L.NndetError("only 3D boxes [B, M, 6] are on the MI355X hot path")
    a = None
    if anchors is not None:
        a = anchors.detach().float().contiguous()
        if tuple(a.shape) != (M, 6):
            raise L.NndetError("anchors must be [M, 6] (shared by the images of the batch)")
    topk = int(topk_candidates) if topk_candidates is not None else 0
    K = min(topk, M) if topk > 0 else M * C
    D = int(detections_per_img) if detections_per_img is not None else K
    ws_bytes = L.load().nndet_postprocess3d_workspace_bytes(B, M, C, topk)
    if ws_bytes == 0:
        raise L.NndetError("nndet_postprocess3d_workspace_bytes: invalid problem size")
    dev = s.device
    ws = L.workspace(ws_bytes, dev)
    out_b = torch.empty((B, D, 6), dtype=torch.float32, device=dev)
    out_s = torch.empty((B, D), dtype=torch.float32, device=dev)
    out_l = torch.empty((B, D), dtype=torch.int64, device=dev)
    out_n = torch.empty((B,), dtype=torch.int64, device=dev)
    ix, iy, iz = (float(image_shape[0]), float(image_shape[1]), float(image_shape[2])) if image_shape is not None else (0., 0., 0.)
    L.call("nndet_postprocess3d_f32", L.ptr(s), int(scores_are_probs), L.ptr(d), L.ptr(a), B, M, C, float(bbox_xform_clip),
           ix, iy, iz, topk, float(score_thresh) if score_thresh is not None else 0.0, int(score_thresh is not None),
           float(remove_small_boxes) if remove_small_boxes is not None else 0.0, int(remove_small_boxes is not None),
           float(nms_thresh), D, L.ptr(out_b), L.ptr(out_s), L.ptr(out_l), L.ptr(out_n), L.ptr(ws), ws_bytes, L.stream())
    return out_b, out_s, out_l, out_n


def postprocess_batch(*args, **kwargs) -> Tuple[List[Tensor], List[Tensor], List[Tensor]]:
    """As `postprocess_batch_raw`, sliced to the reference's return type: lists (one entry per image) of boxes [n, 6],
    scores [n], labels [n]. The only host synchronisation is the read of the B counts."""
    out_b, out_s, out_l, out_n = postprocess_batch_raw(*args, **kwargs)
    counts = out_n.tolist()
    return ([out_b[i, :n] for i, n in enumerate(counts)], [out_s[i, :n] for i, n in enumerate(counts)],
            [out_l[i, :n] for i, n in enumerate(counts)])
