"""Single-model stage of nnDetection's inference ensembler on the GPU: `BoxEnsemblerSelective.postprocess_image`
(nndet/inference/ensembler/detection.py:166-217: sort by probability -> first model_topk -> score threshold -> clip to the
tile -> remove small boxes -> model NMS -> first model_detections_per_image) as ONE fused pass of csrc/postproc.hip
(`nndet_postprocess3d_rows_f32`) instead of ~12 torch launches + a full sort per tile and model.

`postprocess_image_fused` is the function; `AMDPostprocessMixin` drops it into an ensembler class without touching the tile
bookkeeping (`amd_box_ensembler(BoxEnsemblerSelective)`), which is what `RetinaUNetV001AMD.get_ensembler_cls("boxes", 3)` returns.
The fused pass covers the default `model_nms_fn` (batched_nms_model, nndet/inference/detection/model.py:25-54); any other function
(e.g. the weighted variant, whose NMS order is scores * weights) keeps the reference's torch sequence with our GPU NMS underneath."""
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .. import _lib as L


def postprocess_image_fused(boxes: Tensor, probs: Tensor, labels: Tensor, weights: Tensor, shape: Optional[Sequence[int]],
                            model_topk: int, model_score_thresh: Optional[float], remove_small_boxes: Optional[float],
                            model_iou: float, model_detections_per_image: Optional[int]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """boxes [N, 6], probs [N], labels [N], weights [N] of one tile / model -> kept (boxes, probs, labels, weights) in descending
    score order. Ties of equal probabilities are broken by the lower row index (the reference's sort leaves them undefined)."""
    n = boxes.shape[0]
    if n == 0:
        return boxes, probs, labels, weights
    if boxes.shape[1] != 6:
        raise L.NndetError("only 3D boxes are on the MI355X hot path")
    dev = boxes.device
    b = boxes.detach().float().contiguous().view(1, n, 6)
    p = probs.detach().float().contiguous().view(1, n)
    lab = labels.detach().to(dev, torch.int32).contiguous().view(1, n)
    topk = int(model_topk) if model_topk is not None else 0
    K = min(topk, n) if topk > 0 else n
    D = min(int(model_detections_per_image), K) if model_detections_per_image is not None else K
    D = max(D, 1)
    ws_bytes = L.load().nndet_postprocess3d_workspace_bytes(1, n, 1, topk)
    if ws_bytes == 0:
        raise L.NndetError("nndet_postprocess3d_workspace_bytes: invalid problem size")
    ws = L.workspace(ws_bytes, dev)
    ob = torch.empty((1, D, 6), dtype=torch.float32, device=dev)
    os_ = torch.empty((1, D), dtype=torch.float32, device=dev)
    ol = torch.empty((1, D), dtype=torch.int64, device=dev)
    oi = torch.empty((1, D), dtype=torch.int64, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)
    ix, iy, iz = (float(shape[0]), float(shape[1]), float(shape[2])) if shape is not None else (0., 0., 0.)
    L.call("nndet_postprocess3d_rows_f32", L.ptr(p), L.ptr(b), L.ptr(lab), 1, n, ix, iy, iz, topk,
           float(model_score_thresh) if model_score_thresh is not None else 0.0, int(model_score_thresh is not None),
           float(remove_small_boxes) if remove_small_boxes is not None else 0.0, int(remove_small_boxes is not None),
           float(model_iou), D, L.ptr(ob), L.ptr(os_), L.ptr(ol), L.ptr(oi), L.ptr(cnt), L.ptr(ws), ws_bytes, L.stream())
    k = int(cnt.item())
    idx = oi[0, :k]
    return ob[0, :k].to(boxes.dtype), os_[0, :k].to(probs.dtype), labels[idx], weights.to(dev)[idx]


class AMDPostprocessMixin:
    """Mix in FRONT of a reference box ensembler (`BoxEnsemblerSelective`): only `postprocess_image` changes."""

    def postprocess_image(self, boxes: Tensor, probs: Tensor, labels: Tensor, weights: Tensor,
                          shape: Optional[Tuple[int]] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        prm: Dict = self.parameters
        fn = prm.get("model_nms_fn")
        fused_ok = boxes.is_cuda and boxes.dim() == 2 and boxes.shape[-1] == 6 and getattr(fn, "__name__", "") == "batched_nms_model"
        if fused_ok:
            return postprocess_image_fused(boxes, probs, labels, weights, shape, prm["model_topk"], prm["model_score_thresh"],
                                           prm["remove_small_boxes"], prm["model_iou"], prm.get("model_detections_per_image", 1000))
        return super().postprocess_image(boxes, probs, labels, weights, shape)


_cache = {}


def amd_box_ensembler(base: type) -> type:
    """`base` (a reference ensembler class) with the fused `postprocess_image`; one subclass per base, created on first use."""
    cls = _cache.get(base)
    if cls is None:
        cls = _cache[base] = type(base.__name__ + "AMD", (AMDPostprocessMixin, base), {"__doc__": AMDPostprocessMixin.__doc__})
    return cls
