import math
import torch


class BoxCoder:
    """torchvision's published definition: weights + clamp constant log(1000/16)."""

    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode(self, reference_boxes, proposals):
        n = [len(b) for b in reference_boxes]
        t = self.encode_single(torch.cat(reference_boxes, 0), torch.cat(proposals, 0))
        return t.split(n, 0)


class BalancedPositiveNegativeSampler:
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
