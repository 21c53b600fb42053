"""Device-side target preparation (csrc/targets.hip): instance volume -> GT boxes, classes and the semantic map in one pass.
Replaces the `pre_trafo` the reference runs at the top of every training / validation step
(nndet/ptmodule/retinaunet/base.py:108-131,141,163: FindInstances -> Instances2Boxes -> Instances2Segmentation,
nndet/io/transforms/instances.py:26-296)."""
from typing import Dict, List, Sequence, Tuple, Union

import torch
from torch import Tensor

from .. import _lib as L


def instances_to_targets(target: Tensor, instance_mapping: Sequence[Dict[Union[str, int], Union[str, int]]]):
    """target [B, 1, D, H, W] (or [B, D, H, W]) instance ids; instance_mapping: per image {instance id: class}.
    -> (boxes List[Tensor[n_b, 6]] fp32, classes List[Tensor[n_b]] int64, instance ids List[Tensor[n_b]] int32,
        semantic Tensor like `target`: class + 1 on instances, else 0).
    One host synchronisation (the per-image instance counts, needed to return lists)."""
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
    table = table.to(dev, non_blocking=True)
    inst = target.detach().reshape(B, -1).float().contiguous()
    seg = torch.empty_like(inst)
    ext = torch.empty((B, max_id, 6), dtype=torch.int32, device=dev)
    boxes = torch.empty((B, max_id, 6), dtype=torch.float32, device=dev)
    classes = torch.empty((B, max_id), dtype=torch.int64, device=dev)
    ids = torch.empty((B, max_id), dtype=torch.int32, device=dev)
    meta = torch.empty((B + 1,), dtype=torch.int32, device=dev)            # counts[B], err
    L.call("nndet_instances_to_targets_f32", L.ptr(inst), B, D, H, W, L.ptr(table), max_id, L.ptr(seg), L.ptr(ext), L.ptr(boxes),
           L.ptr(classes), L.ptr(ids), L.ptr(meta), meta.data_ptr() + 4 * B, L.stream())
    m = meta.tolist()
    counts, err = m[:B], m[B]
    if err & 1:
        raise KeyError("an instance id of the target is larger than every key of instance_mapping")
    if err & 2:
        raise KeyError("an instance of the target has no entry in instance_mapping")     # the reference raises KeyError too
    return ([boxes[b, :n] for b, n in enumerate(counts)], [classes[b, :n] for b, n in enumerate(counts)],
            [ids[b, :n] for b, n in enumerate(counts)], seg.view(target.shape).to(target.dtype))


def prepare_targets(data: Tensor, target: Tensor, instance_mapping) -> Tuple[Tensor, dict]:
    """The (images, targets) pair `BaseRetinaNet.train_step` takes, from a raw nnDetection batch
    (nndet/ptmodule/retinaunet/base.py:135-154)."""
    boxes, classes, _, seg = instances_to_targets(target, instance_mapping)
    return data, {"target_boxes": boxes, "target_classes": classes, "target_seg": seg[:, 0] if seg.dim() == 5 else seg}
