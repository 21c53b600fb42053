"""Box coder. `decode_single` mirrors nndet/core/boxes/coder.py:90-155 (differentiable torch ops on the <= 42
sampled positives of the loss); `BoxCoderND.decode` over ALL anchors runs the fused HIP decode(+clip) kernel."""
import math
from typing import List, Sequence

import torch
from torch import Tensor

from ... import _lib as L

BBOX_XFORM_CLIP = math.log(1000. / 16)   # torchvision BoxCoder default the reference inherits


def decode_single(rel_codes: Tensor, boxes: Tensor, weights: Sequence[float] = (1.,) * 6,
                  bbox_xform_clip: float = BBOX_XFORM_CLIP) -> Tensor:
    boxes = boxes.to(rel_codes.dtype)
    w = boxes[:, 2] - boxes[:, 0]; h = boxes[:, 3] - boxes[:, 1]; d = boxes[:, 5] - boxes[:, 4]
    cx = boxes[:, 0] + 0.5 * w; cy = boxes[:, 1] + 0.5 * h; cz = boxes[:, 4] + 0.5 * d
    dx = rel_codes[:, 0] / weights[0]; dy = rel_codes[:, 1] / weights[1]
    dw = torch.clamp(rel_codes[:, 2] / weights[2], max=bbox_xform_clip)
    dh = torch.clamp(rel_codes[:, 3] / weights[3], max=bbox_xform_clip)
    dz = rel_codes[:, 4] / weights[4]
    dd = torch.clamp(rel_codes[:, 5] / weights[5], max=bbox_xform_clip)
    pcx = dx * w + cx; pcy = dy * h + cy; pcz = dz * d + cz
    pw = torch.exp(dw) * w; ph = torch.exp(dh) * h; pd = torch.exp(dd) * d
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph,
                        pcz - 0.5 * pd, pcz + 0.5 * pd], dim=1)


def decode_clip(rel_codes: Tensor, anchors: Tensor, image_shape=None, bbox_xform_clip: float = BBOX_XFORM_CLIP) -> Tensor:
    """Fused decode (+ clip_boxes_to_image_3d_, clip.py:83-101) of [n,6] deltas against [n_anchor,6] anchors
    (anchor row = i % n_anchor): one pass, 48 B read + 24 B written per box."""
    r = rel_codes.detach().float().contiguous().reshape(-1, 6)
    a = anchors.detach().float().contiguous()
    out = torch.empty_like(r)
    ix, iy, iz = (float(image_shape[0]), float(image_shape[1]), float(image_shape[2])) if image_shape is not None else (0., 0., 0.)
    L.call("nndet_decode_clip3d_f32", L.ptr(r), L.ptr(a), r.shape[0], a.shape[0], float(bbox_xform_clip), ix, iy, iz, L.ptr(out), L.stream())
    return out


class BoxCoderND:
    def __init__(self, weights: Sequence[float], bbox_xform_clip: float = BBOX_XFORM_CLIP):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip
        if any(float(w) != 1.0 for w in weights):
            raise L.NndetError("RetinaUNet uses unit coder weights (retinaunet/base.py:388)")

    def decode_single(self, rel_codes: Tensor, boxes: Tensor) -> Tensor:
        return decode_single(rel_codes, boxes, self.weights, self.bbox_xform_clip)

    def decode(self, rel_codes: Tensor, boxes: List[Tensor]) -> Tensor:
        """All anchors of the batch (nndet/core/boxes/coder.py:217-240). `boxes` holds the same anchor tensor per image."""
        assert isinstance(boxes, (list, tuple))
        same = all(b is boxes[0] for b in boxes)
        anchors = boxes[0] if same else torch.cat(boxes, dim=0)
        return decode_clip(rel_codes, anchors, None, self.bbox_xform_clip)
