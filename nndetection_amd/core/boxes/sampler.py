"""Batch-level hard-negative sampler on the device (csrc/sampler.hip). Same constructor, counts and selection rule as
HardNegativeSamplerBatched (nndet/core/boxes/sampler.py:57-98,154-185,237-270): positives = a random subset of the foreground
anchors, negatives = a random subset of the pool of highest-scoring background anchors -- but no torch.where / topk / randperm
round trips: counts stay on the device, the index lists have fixed capacities.

Randomness: the reference draws torch.randperm(n)[:k] (a uniformly random k-subset); here every candidate gets the key
hash(seed, index) and the k smallest keys win -- the same distribution. The seed comes from torch's CPU generator, so
torch.manual_seed makes runs reproducible. Parity tests need the SAME subset as the reference / oracle run under their patched
`torch.randperm := arange(n-1, -1, -1)`: set `HardNegativeSamplerBatched.deterministic = True` (or patch torch.randperm with a
function that carries the attribute `nndet_reversed_arange = True`, which is what tests/gpu_util.det_randperm does)."""
from typing import List, Tuple

import torch
from torch import Tensor

from ... import _lib as L


class HardNegativeSamplerBatched:
    deterministic: bool = False

    def __init__(self, batch_size_per_image: int, positive_fraction: float, min_neg: int = 0, pool_size: float = 10):
        self.min_neg = min_neg
        self._batch_size_per_image = batch_size_per_image
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pool_size = pool_size

    # ---- the reference's count rules (kept as methods: sampler.py:154-185)
    def get_num_pos(self, positive: Tensor) -> int:
        return min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))

    def get_num_neg(self, negative: Tensor, num_pos: int) -> int:
        num_neg = int(max(1, num_pos) * abs(1 - 1. / float(self.positive_fraction)))
        return min(negative.numel(), max(num_neg, self.min_neg))

    def _is_deterministic(self) -> bool:
        return bool(self.deterministic or getattr(torch.randperm, "nndet_reversed_arange", False))

    def sample_device(self, labels: Tensor, scores: Tensor, batch_size: int, scores_are_probs: bool = False):
        """labels [N] (batch concatenated), scores [N, C] logits (or [N] fg probabilities). Asynchronous.
        -> (pos_idx [P_cap] int64, neg_idx [NEG_cap] int64, counts [4] int64 = num_pos, num_neg, #pos, #neg) on the device;
        the index lists are ascending and padded with -1."""
        self.batch_size_per_image = self._batch_size_per_image * batch_size
        lab = labels.detach().float().contiguous()
        sc = scores.detach().float().contiguous()
        N = lab.shape[0]
        C = 1 if scores_are_probs else sc.shape[1]
        p_cap = int(self.batch_size_per_image * self.positive_fraction)
        ratio = abs(1 - 1. / float(self.positive_fraction))
        lib = L.load()
        neg_cap = lib.nndet_hnm_neg_capacity(p_cap, ratio, int(self.min_neg))
        ws_bytes = lib.nndet_hnm_sample_workspace_bytes(p_cap, ratio, int(self.min_neg), float(self.pool_size))
        if neg_cap < 0 or ws_bytes == 0:
            raise L.NndetError("hard-negative sampler: invalid configuration")
        dev = lab.device
        pos = torch.empty((max(p_cap, 1),), dtype=torch.int64, device=dev)
        neg = torch.empty((neg_cap,), dtype=torch.int64, device=dev)
        counts = torch.empty((4,), dtype=torch.int64, device=dev)
        ws = L.workspace(ws_bytes, dev)
        det = self._is_deterministic()
        seed = 0 if det else int(torch.randint(0, 2 ** 62, (1,)).item())        # CPU generator: no device round trip
        L.call("nndet_hnm_sample_f32", L.ptr(lab), L.ptr(sc), int(scores_are_probs), N, C, p_cap, ratio, int(self.min_neg),
               float(self.pool_size), seed, int(det), L.ptr(pos), L.ptr(neg), L.ptr(counts), L.ptr(ws), ws_bytes, L.stream())
        return pos[:p_cap], neg, counts

    def sample_indices(self, labels: Tensor, scores: Tensor, batch_size: int, scores_are_probs: bool = False) -> Tuple[Tensor, Tensor]:
        """(sampled_pos_inds, sampled_neg_inds) exactly as DetectionHeadHNM.select_indices returns them (ascending anchor
        indices). One host read: the two counts."""
        pos, neg, counts = self.sample_device(labels, scores, batch_size, scores_are_probs)
        n_pos, n_neg = counts[:2].tolist()
        return pos[:n_pos], neg[:n_neg]

    def __call__(self, target_labels: List[Tensor], fg_probs: Tensor):
        """The reference's public contract: List[mask], List[mask] (uint8 [sum of anchors]) for the whole batch."""
        labels = torch.cat(target_labels, dim=0)
        pos, neg = self.sample_indices(labels, fg_probs, len(target_labels), scores_are_probs=True)
        pos_mask = torch.zeros_like(labels, dtype=torch.uint8)
        neg_mask = torch.zeros_like(labels, dtype=torch.uint8)
        pos_mask[pos] = 1
        neg_mask[neg] = 1
        return [pos_mask], [neg_mask]
