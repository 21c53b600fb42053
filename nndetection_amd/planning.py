"""Planner-side consumers of the hot path, ROCm-safe.

* `MemoryEstimatorDetection` -- same constructor / `estimate` / `measure` contract as
  nndet/planning/estimator.py:61-260, which the planner calls with `network_cls.from_config_plan`
  (nndet/planning/architecture/boxes/c002.py:209-212). The reference reads `nvidia-smi` (estimator.py:263-281) and wraps
  the measurement in `cudnn_deterministic`; neither exists on an MI355X box, so `nndet_prep` cannot run there. Here the
  occupied device memory comes from `torch.cuda.mem_get_info` (HIP's hipMemGetInfo), the GPU count from
  `torch.cuda.device_count`, and the reference targets are extended by the MI355X (288 GB HBM3E).
* `anchor_fit` / `AnchorFitObjective` -- the objective of the planner's anchor search
  (nndet/planning/architecture/boxes/base.py:424-484: `box_iou(boxes.cuda(), anchors.cuda()).max(dim=1)[0].mean()`, 15 000
  evaluations) as ONE kernel that never writes the [G, A] matrix (csrc/boxes.hip:k_iou_rowmax); the ground-truth boxes are
  uploaded once.
"""
import copy
import gc
import math
import time
from typing import Callable, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import _lib as L


def b2mb(x): return x / (2 ** 20)
def mb2b(x): return x * (2 ** 20)


# target cards: the reference's entry (estimator.py:42-51) + the MI355X (288 GB HBM3E; the HIP context + the library's code
# objects take ~0.9 GB there as well, measured with torch.cuda.mem_get_info on an idle device)
ARCHS = {"RTX2080TI": 11523260416 - int(mb2b(11)), "MI355X": 288 * 10 ** 9 - int(mb2b(64))}
CUDA_CONTEXT = {"none": 0, "RTX2080TI": int(mb2b(910)), "MI355X": int(mb2b(900))}


def num_gpus() -> int:
    """Number of GPUs (estimator.py:263-267 parses `nvidia-smi -L`)."""
    return torch.cuda.device_count()


def smi_memory_allocated(gpu_id: int = 0) -> int:
    """Bytes in use on the device by ALL processes (estimator.py:270-281 parses `nvidia-smi --query-gpu=memory.used`)."""
    free, total = torch.cuda.mem_get_info(gpu_id)
    return int(total - free)


class MemoryEstimatorDetection:
    def __init__(self, target_mem: Union[float, str] = "MI355X", gpu_id: int = 0, context: Union[float, str] = "MI355X",
                 offset: int = mb2b(768), batch_size: int = 1, mixed_precision: bool = True):
        self.context = CUDA_CONTEXT[context] if isinstance(context, str) else context
        self.offset = offset
        self.block_mem_tensor = None
        self.target_mem = ARCHS[target_mem] if isinstance(target_mem, str) else target_mem
        self.gpu_id = gpu_id
        self.batch_size = batch_size
        self.mixed_precision = mixed_precision

    def create_offset_tensor_on_GPU(self) -> Tensor:
        device = f"cuda:{self.gpu_id}"
        return torch.rand(math.ceil(self.offset / 8), dtype=torch.float64, requires_grad=False, device=device)

    def estimate(self, min_shape: Sequence[int], target_shape: Sequence[int], network, optimizer_cls: Callable = torch.optim.Adam,
                 in_channels: int = None, num_instances: int = 1) -> Tuple[int, bool]:
        """-> (estimated bytes, fits into `target_mem`), estimator.py:107-150."""
        if in_channels is not None:
            min_shape = [in_channels, *min_shape]
            target_shape = [in_channels, *target_shape]
        self.available_mem = torch.cuda.get_device_properties(self.gpu_id).total_memory - smi_memory_allocated(self.gpu_id) + self.context
        fixed, dynamic = self.measure(shape=target_shape, network=copy.deepcopy(network), optimizer_cls=optimizer_cls,
                                      num_instances=num_instances)
        estimated = fixed + dynamic
        self.block_mem_tensor = None
        torch.cuda.empty_cache()
        gc.collect()
        return estimated, estimated < self.target_mem

    def measure(self, shape: Sequence[int], network, optimizer_cls: Callable = torch.optim.Adam, num_instances: int = 1):
        """10 training steps of the network at `shape` = (C, *spatial) (estimator.py:184-260): reserved memory before the first
        step (parameters + optimizer = fixed) and after the last (activations, workspaces = dynamic). The HIP kernels compute in
        bf16 when `mixed_precision` (the reference's autocast fp16) and need no GradScaler."""
        device = torch.device("cuda", self.gpu_id)
        loss = opt = inp = block = None
        try:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(device)
            alloc0 = torch.cuda.memory_allocated(device)
            network = network.to(device)
            empty_mem = torch.cuda.memory_reserved(device)
            opt = optimizer_cls(network.parameters())
            boxes = [[0, 0, 2, 2]]
            if len(shape) == 4:
                boxes[0].extend((0, 2))
            block = self.create_offset_tensor_on_GPU()
            dt = torch.bfloat16 if self.mixed_precision else torch.float32
            for _ in range(10):
                opt.zero_grad()
                inp = {"images": torch.rand((self.batch_size, *shape), device=device).to(dt),
                       "targets": {"target_boxes": [torch.tensor(boxes, device=device, dtype=torch.float).repeat(num_instances, 1)
                                                    for _ in range(self.batch_size)],
                                   "target_classes": [torch.tensor([0] * num_instances, device=device, dtype=torch.float)
                                                      for _ in range(self.batch_size)],
                                   "target_seg": torch.zeros((self.batch_size, *shape[1:]), device=device, dtype=torch.float)}}
                fixed_mem = torch.cuda.memory_reserved(device)
                alloc_fixed = torch.cuda.memory_allocated(device)
                loss_dict, _ = network.train_step(images=inp["images"], targets=inp["targets"], evaluation=False, batch_num=0)
                loss = sum(loss_dict.values())
                loss.backward()
                opt.step()
            torch.cuda.synchronize(device)
            dyn_mem = torch.cuda.memory_reserved(device)
            # The reference measures the caching allocator's RESERVED bytes in a fresh process. In a process whose allocator
            # already holds free blocks (tests, notebooks) reserved memory does not grow; the peak of ALLOCATED bytes is the lower
            # bound that is always valid, so report the larger of the two views.
            fixed = max(fixed_mem - empty_mem, alloc_fixed - alloc0)
            dynamic = max(dyn_mem - fixed_mem, torch.cuda.max_memory_allocated(device) - alloc_fixed)
            empty_mem, fixed_mem, dyn_mem = 0, fixed, fixed + dynamic
        except Exception as e:                                  # out of memory: report "does not fit" like the reference
            self.last_error = e
            empty_mem, fixed_mem, dyn_mem = 0, float("Inf"), float("Inf")
        finally:
            del loss, opt, inp, block
        network.cpu()
        torch.cuda.empty_cache()
        gc.collect()
        return fixed_mem - empty_mem, dyn_mem - fixed_mem


def iou_rowmax(boxes: Tensor, anchors: Tensor, eps: float = 0.0) -> Tensor:
    """max over anchors of IoU per box, [G] fp32, without the [G, A] matrix."""
    a = boxes.detach().float().contiguous()
    b = anchors.detach().float().contiguous().to(a.device)
    if a.shape[-1] != 6 or b.shape[-1] != 6 or b.shape[0] == 0:
        raise L.NndetError("iou_rowmax needs 3D boxes [G, 6] and at least one anchor [A, 6]")
    out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    L.call("nndet_iou3d_rowmax_f32", L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], float(eps), L.ptr(out), L.stream())
    return out


def anchor_fit(boxes: Tensor, anchors: Tensor) -> Tensor:
    """`box_iou(boxes, anchors).max(dim=1)[0].mean()` (base.py:470-471) as a 0-d device tensor."""
    return iou_rowmax(boxes, anchors).mean()


class AnchorFitObjective:
    """Callable for the anchor optimiser's ask / tell loop (base.py:461-474): keeps the ground-truth boxes on the device,
    evaluates `anchor_generator.generate_anchors(*sizes)` + `compute_anchors_for_strides(...)` candidates against them."""

    def __init__(self, boxes: Tensor, device="cuda"):
        self.boxes = boxes.detach().float().contiguous().to(device)

    def __call__(self, anchors: Tensor) -> float:
        return float(anchor_fit(self.boxes, anchors.to(self.boxes.device)).item())
