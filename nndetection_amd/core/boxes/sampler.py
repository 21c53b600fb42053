"""Batch-level hard-negative sampler. Mirrors HardNegativeSamplerBatched
(nndet/core/boxes/sampler.py:57-98,101-270). The RNG stays torch.randperm exactly as in the reference."""
from typing import List

import torch
from torch import Tensor


class HardNegativeSamplerBatched:
    def __init__(self, batch_size_per_image: int, positive_fraction: float, min_neg: int = 0, pool_size: float = 10):
        self.min_neg = min_neg
        self._batch_size_per_image = batch_size_per_image
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pool_size = pool_size

    def get_num_pos(self, positive: Tensor) -> int:
        return min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))

    def get_num_neg(self, negative: Tensor, num_pos: int) -> int:
        num_neg = int(max(1, num_pos) * abs(1 - 1. / float(self.positive_fraction)))
        return min(negative.numel(), max(num_neg, self.min_neg))

    def __call__(self, target_labels: List[Tensor], fg_probs: Tensor):
        batch_size = len(target_labels)
        self.batch_size_per_image = self._batch_size_per_image * batch_size
        labels = torch.cat(target_labels, dim=0)
        positive = torch.where(labels >= 1)[0]
        negative = torch.where(labels == 0)[0]
        num_pos = self.get_num_pos(positive)
        perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
        pos_mask = torch.zeros_like(labels, dtype=torch.uint8)
        pos_mask[positive[perm1]] = 1
        num_neg = self.get_num_neg(negative, num_pos)
        pool = min(negative.numel(), int(num_neg * self.pool_size))
        _, pool_idx = fg_probs[negative].topk(pool, sorted=True)
        negative = negative[pool_idx]
        perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
        neg_mask = torch.zeros_like(labels, dtype=torch.uint8)
        neg_mask[negative[perm2]] = 1
        return [pos_mask], [neg_mask]
