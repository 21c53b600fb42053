"""3D NMS on the GPU (csrc/nms3d.hip) -- drop-in for `nndet.core.boxes.nms.nms` / `batched_nms`
(nndet/core/boxes/nms.py:56-106) and for the native `nndet._C.nms` (nndet/csrc/cuda/nms.cu:148-221).
"""
import torch
from torch import Tensor

from ... import _lib as L


def _nms_raw(boxes: Tensor, scores: Tensor, iou_threshold: float):
    """Asynchronous part: returns (keep_padded [N] int64 with -1 tail, n_keep [1] int64 on device)."""
    b = boxes.detach().float().contiguous()
    s = scores.detach().float().contiguous()
    n = b.shape[0]
    keep = torch.empty((n,), dtype=torch.int64, device=b.device)
    n_keep = torch.empty((1,), dtype=torch.int64, device=b.device)
    ws_bytes = L.load().nndet_nms3d_workspace_bytes(n)
    if ws_bytes == 0:
        raise L.NndetError("nndet_nms3d_workspace_bytes failed")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=b.device)
    L.call("nndet_nms2d_f32" if b.shape[1] == 4 else "nndet_nms3d_f32", L.ptr(b), L.ptr(s), n, float(iou_threshold), L.ptr(keep),
           L.ptr(n_keep), L.ptr(ws), ws_bytes, L.stream())
    return keep, n_keep


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """keep indices (int64, decreasing score), same contract as nndet._C.nms: [N, 6] boxes (x1, y1, x2, y2, z1, z2) or [N, 4]
    (the reference sends 2D boxes to torchvision.ops.nms, nms.py:70-72; its own 2D kernel is nms.cu:54-96)."""
    if boxes.shape[0] == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)   # cpu/nms.cpp:24-26
    if boxes.shape[1] not in (4, 6):
        raise L.NndetError("NMS takes [N, 4] or [N, 6] boxes")
    keep, n_keep = _nms_raw(boxes, scores, iou_threshold)
    return keep[: int(n_keep.item())]     # the one host sync (the reference syncs on the mask D2H instead)


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    """nndet/core/boxes/nms.py:81-106: per-class NMS through the coordinate-offset trick (fp32 offsets)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def nms_gpu(dets: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Name the reference looks up at call time (`nndet.core.boxes.nms.nms_gpu`, nms.py:23-27,74-78)."""
    return nms(dets, scores, iou_threshold)
