"""nndet/core/boxes/clip.py:83-101 (in-place, 3D)."""
import torch


def clip_boxes_to_image_(boxes: torch.Tensor, img_shape):
    s0, s1, s2 = img_shape
    boxes[..., 0::6].clamp_(min=0, max=s0); boxes[..., 1::6].clamp_(min=0, max=s1)
    boxes[..., 2::6].clamp_(min=0, max=s0); boxes[..., 3::6].clamp_(min=0, max=s1)
    boxes[..., 4::6].clamp_(min=0, max=s2); boxes[..., 5::6].clamp_(min=0, max=s2)
    return boxes
