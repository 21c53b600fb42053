"""Device-side target preparation (csrc/targets.hip): instance volume -> GT boxes, classes and the semantic map in one pass.
Replaces the `pre_trafo` the reference runs at the top of every training / validation step
(nndet/ptmodule/retinaunet/base.py:108-131,141,163: FindInstances -> Instances2Boxes -> Instances2Segmentation,
nndet/io/transforms/instances.py:26-296)."""
from typing import Dict, List, Sequence, Tuple, Union

import torch
from torch import Tensor

from .. import _lib as L


def instances_to_targets(target: Tensor, instance_mapping: Sequence[Dict[Union[str, int], Union[str, int]]], deferred: bool = False):
    """target [B, 1, D, H, W] (or [B, D, H, W]) instance ids; instance_mapping: per image {instance id: class}.
    -> (boxes List[Tensor[n_b, 6]] fp32, classes List[Tensor[n_b]] int64, instance ids List[Tensor[n_b]] int32,
        semantic Tensor like `target`: class + 1 on instances, else 0).
    One host synchronisation (the per-image instance counts, needed to return lists); deferred=True returns a `DeferredTargets`
    instead, whose `resolve()` performs that read later (after the caller has queued work that does not need the counts)."""
    if target.dim() == 5:
        if target.shape[1] != 1:
            raise L.NndetError("instance target must have one channel")
        B, _, D, H, W = target.shape
    elif target.dim() == 4:
        B, D, H, W = target.shape
    else:
        raise L.NndetError("only 3D instance volumes [B, 1, D, H, W] are on the MI355X hot path")
    if len(instance_mapping) != B:
        raise L.NndetError("one instance_mapping per image is required")
    dev = target.device
    maps = [{int(k): int(v) for k, v in m.items()} for m in instance_mapping]
    max_id = max([max(m.keys(), default=0) for m in maps] + [1]) + 1
    max_id = (max_id + 63) // 64 * 64
    table = torch.full((B, max_id), -1, dtype=torch.int32)
    for b, m in enumerate(maps):
        for k, v in m.items():
            if k > 0:
                table[b, k] = v
    # (pinned: a pageable host -> device copy blocks the host until the stream has drained on ROCm)
    table = (table.pin_memory() if dev.type == "cuda" else table).to(dev, non_blocking=True)
    inst = target.detach().reshape(B, -1).float().contiguous()
    seg = torch.empty_like(inst)
    ext = torch.empty((B, max_id, 6), dtype=torch.int32, device=dev)
    boxes = torch.empty((B, max_id, 6), dtype=torch.float32, device=dev)
    classes = torch.empty((B, max_id), dtype=torch.int64, device=dev)
    ids = torch.empty((B, max_id), dtype=torch.int32, device=dev)
    meta = torch.empty((B + 1,), dtype=torch.int32, device=dev)            # counts[B], err
    L.call("nndet_instances_to_targets_f32", L.ptr(inst), B, D, H, W, L.ptr(table), max_id, L.ptr(seg), L.ptr(ext), L.ptr(boxes),
           L.ptr(classes), L.ptr(ids), L.ptr(meta), meta.data_ptr() + 4 * B, L.stream())
    seg = seg.view(target.shape).to(target.dtype)
    if deferred and dev.type == "cuda":
        host = torch.empty((B + 1,), dtype=torch.int32, pin_memory=True)
        host.copy_(meta, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return DeferredTargets(boxes, classes, ids, seg, host, ev, (meta, table, inst))
    return _split(boxes, classes, ids, meta.tolist()) + (seg,)


def _split(boxes, classes, ids, m):
    B = boxes.shape[0]
    counts, err = m[:B], m[B]
    if err & 1:
        raise KeyError("an instance id of the target is larger than every key of instance_mapping")
    if err & 2:
        raise KeyError("an instance of the target has no entry in instance_mapping")     # the reference raises KeyError too
    return ([boxes[b, :n] for b, n in enumerate(counts)], [classes[b, :n] for b, n in enumerate(counts)],
            [ids[b, :n] for b, n in enumerate(counts)])


class DeferredTargets:
    """Targets whose per-image instance COUNTS are still on their way to the host. The lists `target_boxes` / `target_classes` have
    data-dependent lengths, so building them needs one device -> host read per step; read at the top of `training_step` (as
    `meta.tolist()` does) it drains the whole previous step before the first kernel of this one is queued. `BaseRetinaNet.train_step`
    accepts this object instead of the dict: it queues the forward pass first and calls `resolve()` afterwards -- the counts were
    copied to pinned memory by a non-blocking copy in front of the forward pass, so the wait ends as soon as the previous step and
    the target kernel are done, with this step's forward pass already in the queue. `target_seg` does not depend on the counts."""

    def __init__(self, boxes, classes, ids, seg, host, ev, keep):
        self._padded, self._host, self._ev, self._keep = (boxes, classes, ids), host, ev, keep
        self.target_seg = seg[:, 0] if seg.dim() == 5 else seg
        self._dict = None

    def resolve(self) -> dict:
        if self._dict is None:
            self._ev.synchronize()
            b, c, i = _split(*self._padded, self._host.tolist())
            self._dict = {"target_boxes": b, "target_classes": c, "target_ids": i, "target_seg": self.target_seg}
            self._keep = None
        return self._dict

    def __getitem__(self, key):                    # dict-like for callers that do not know about deferral (validation, evaluators)
        return self.target_seg if key == "target_seg" else self.resolve()[key]

    def keys(self):
        return ("target_boxes", "target_classes", "target_seg")


def prepare_targets(data: Tensor, target: Tensor, instance_mapping, deferred: bool = False):
    """The (images, targets) pair `BaseRetinaNet.train_step` takes, from a raw nnDetection batch
    (nndet/ptmodule/retinaunet/base.py:135-154). deferred=True (CUDA only): targets is a `DeferredTargets`."""
    res = instances_to_targets(target, instance_mapping, deferred=deferred)
    if isinstance(res, DeferredTargets):
        return data, res
    boxes, classes, _, seg = res
    return data, {"target_boxes": boxes, "target_classes": classes, "target_seg": seg[:, 0] if seg.dim() == 5 else seg}
