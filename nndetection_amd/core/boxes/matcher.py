"""ATSS matcher on the GPU (csrc/atss3d.hip). Mirrors ATSSMatcher / Matcher
(nndet/core/boxes/matcher/atss.py:20-122, matcher/base.py:13-91), center_in_gt=False (V001) and True (the reference's default)."""
from typing import Callable, Sequence, Tuple

import ctypes
import torch
from torch import Tensor

from ... import _lib as L
from .ops import box_iou


class ATSSMatcher:
    BELOW_LOW_THRESHOLD: int = -1
    BETWEEN_THRESHOLDS: int = -2

    def __init__(self, num_candidates: int, similarity_fn: Callable = box_iou, center_in_gt: bool = True,
                 return_match_quality: bool = False):
        self.similarity_fn = similarity_fn
        self.num_candidates = num_candidates
        self.min_dist = 0.01
        self.center_in_gt = center_in_gt
        self.return_match_quality = return_match_quality

    def __call__(self, boxes: Tensor, anchors: Tensor, num_anchors_per_level: Sequence[int],
                 num_anchors_per_loc: int) -> Tuple[Tensor, Tensor]:
        """-> (match_quality_matrix, matches [M] int64 with -1 for background).
        The dense [G,M] IoU matrix is only produced when `return_match_quality` is set (the detector only reads
        `.numel() > 0` of it, nndet/core/retina.py:267); otherwise a 1-element placeholder is returned."""
        M = anchors.shape[0]
        if boxes.numel() == 0:                                          # matcher/base.py:51-56
            mq = torch.tensor([]).to(anchors)
            return mq, torch.full((M,), self.BELOW_LOW_THRESHOLD, dtype=torch.int64, device=anchors.device)
        gt = boxes.detach().float().contiguous().to(anchors.device)
        an = anchors.detach().float().contiguous()
        G, Lv = gt.shape[0], len(num_anchors_per_level)
        offs = [0]
        for n in num_anchors_per_level:
            offs.append(offs[-1] + int(n))
        assert offs[-1] == M, "num_anchors_per_level does not sum to the number of anchors"
        k = self.num_candidates * num_anchors_per_loc
        matches = torch.empty((M,), dtype=torch.int64, device=an.device)
        ws_bytes = L.load().nndet_atss3d_workspace_bytes(G, M, Lv, k)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=an.device)
        offs_c = (ctypes.c_int64 * (Lv + 1))(*offs)
        if self.center_in_gt:                   # (the entry that takes the flag also writes labels: a scratch tensor here)
            scratch = torch.empty((M,), dtype=torch.float32, device=an.device)
            L.call("nndet_atss3d_assign_batched_f32", L.ptr(gt), None, G, (ctypes.c_int32 * 2)(0, G), 1, L.ptr(an), M, offs_c, Lv, k, 1,
                   float(self.min_dist), L.ptr(matches), L.ptr(scratch), L.ptr(ws), ws_bytes, L.stream())
        else:
            L.call("nndet_atss3d_match_f32", L.ptr(gt), G, L.ptr(an), M, offs_c, Lv, k, L.ptr(matches), L.ptr(ws), ws_bytes, L.stream())
        mq = self.similarity_fn(gt, an) if self.return_match_quality else an.new_ones(1)
        return mq, matches

    def match_batch(self, boxes: Sequence[Tensor], anchors: Tensor, num_anchors_per_level: Sequence[int],
                    num_anchors_per_loc: int, classes: Sequence[Tensor] = None):
        """All images of a batch against their shared anchors in ONE pass (nndet_atss3d_match_batched_f32).
        -> (gt_all [G,6] concatenated, matches [B, M] with indices local to the image / -1, offsets [B+1]).
        With `classes` (per image, as `boxes`): the kernel also writes the anchors' training labels (class of the matched GT + 1, 0 =
        background; nndet/core/retina.py:262-287) and a fourth value `labels` [B, M] fp32 is returned."""
        M, B = anchors.shape[0], len(boxes)
        dev = anchors.device
        offs_img = [0]
        for b in boxes:                                   # the reference's empty GT is `tensor([[]])` (instances.py:124-125): numel, not shape[0]
            offs_img.append(offs_img[-1] + int(b.numel() // 6))
        G = offs_img[-1]
        an = anchors.detach().float().contiguous()
        nz = [b.detach().to(dev, torch.float32).reshape(-1, 6) for b in boxes if b.numel() > 0]
        gt = torch.cat(nz, 0).contiguous() if nz else an.new_zeros((0, 6))
        Lv = len(num_anchors_per_level)
        offs = [0]
        for n in num_anchors_per_level:
            offs.append(offs[-1] + int(n))
        assert offs[-1] == M, "num_anchors_per_level does not sum to the number of anchors"
        k = self.num_candidates * num_anchors_per_loc
        matches = torch.empty((B, M), dtype=torch.int64, device=dev)
        ws_bytes = L.load().nndet_atss3d_workspace_bytes(max(G, 1), M, Lv, k)
        ws = L.workspace(ws_bytes, dev)
        if classes is not None:
            cl = [c.detach().to(dev, torch.float32).reshape(-1) for c, b in zip(classes, boxes) if b.numel() > 0]
            gt_cls = (cl[0] if len(cl) == 1 else torch.cat(cl, 0)).contiguous() if cl else None
            if gt_cls is not None and gt_cls.numel() != G:
                raise L.NndetError("ATSS: one class per ground-truth box expected")
            labels = torch.empty((B, M), dtype=torch.float32, device=dev)
            L.call("nndet_atss3d_assign_batched_f32", L.ptr(gt) if G else None, L.ptr(gt_cls) if gt_cls is not None else None, G,
                   (ctypes.c_int32 * (B + 1))(*offs_img), B, L.ptr(an), M, (ctypes.c_int64 * (Lv + 1))(*offs), Lv, k,
                   int(bool(self.center_in_gt)), float(self.min_dist), L.ptr(matches), L.ptr(labels), L.ptr(ws), ws_bytes, L.stream())
            return gt, matches, offs_img, labels
        if self.center_in_gt:
            scratch = torch.empty((B, M), dtype=torch.float32, device=dev)
            L.call("nndet_atss3d_assign_batched_f32", L.ptr(gt) if G else None, None, G, (ctypes.c_int32 * (B + 1))(*offs_img), B, L.ptr(an), M,
                   (ctypes.c_int64 * (Lv + 1))(*offs), Lv, k, 1, float(self.min_dist), L.ptr(matches), L.ptr(scratch), L.ptr(ws), ws_bytes,
                   L.stream())
            return gt, matches, offs_img
        L.call("nndet_atss3d_match_batched_f32", L.ptr(gt) if G else None, G, (ctypes.c_int32 * (B + 1))(*offs_img), B,
               L.ptr(an), M, (ctypes.c_int64 * (Lv + 1))(*offs), Lv, k, L.ptr(matches), L.ptr(ws), ws_bytes, L.stream())
        return gt, matches, offs_img


class IoUMatcher:
    """IoU-threshold matcher on the GPU (csrc/atss3d.hip: k_ioum_*). Mirrors IoUMatcher (nndet/core/boxes/matcher/iou.py:20-107): same
    constructor, `__call__(boxes, anchors, num_anchors_per_level, num_anchors_per_loc)` -> (match quality matrix, matches [M] int64 with
    BELOW_LOW_THRESHOLD / BETWEEN_THRESHOLDS). As for ATSS the dense [G, M] IoU matrix is only produced on request."""
    BELOW_LOW_THRESHOLD: int = -1
    BETWEEN_THRESHOLDS: int = -2

    def __init__(self, low_threshold: float, high_threshold: float, allow_low_quality_matches: bool, similarity_fn: Callable = box_iou,
                 return_match_quality: bool = False):
        assert low_threshold <= high_threshold
        self.low_threshold, self.high_threshold = low_threshold, high_threshold
        self.allow_low_quality_matches = allow_low_quality_matches
        self.similarity_fn, self.return_match_quality = similarity_fn, return_match_quality

    def __call__(self, boxes: Tensor, anchors: Tensor, num_anchors_per_level: Sequence[int] = None, num_anchors_per_loc: int = None,
                 **kwargs) -> Tuple[Tensor, Tensor]:
        M = anchors.shape[0]
        if boxes.numel() == 0:                                          # matcher/base.py:51-56
            return torch.tensor([]).to(anchors), torch.full((M,), self.BELOW_LOW_THRESHOLD, dtype=torch.int64, device=anchors.device)
        gt = boxes.detach().float().contiguous().to(anchors.device)
        an = anchors.detach().float().contiguous()
        if gt.shape[1] != 6 or an.shape[1] != 6:
            raise L.NndetError("IoUMatcher: 3D boxes [N, 6] only")
        G = gt.shape[0]
        matches = torch.empty((M,), dtype=torch.int64, device=an.device)
        ws = torch.empty((max(G * 8, 8),), dtype=torch.uint8, device=an.device)
        L.call("nndet_iou_match3d_f32", L.ptr(gt), G, L.ptr(an), M, float(self.low_threshold), float(self.high_threshold),
               int(bool(self.allow_low_quality_matches)), L.ptr(matches), L.ptr(ws), ws.numel(), L.stream())
        mq = self.similarity_fn(gt, an) if self.return_match_quality else an.new_ones(1)
        return mq, matches
