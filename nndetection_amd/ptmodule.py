"""Plugin surface: `RetinaUNetV001` built from a plan exactly like the reference's
`RetinaUNetModule.from_config_plan` (nndet/ptmodule/retinaunet/base.py:338-675, v001.py:29-38), with every
class attribute pointing at the HIP-backed implementation. Also the optimizer / LR schedule of
`configure_optimizers` (base.py:300-336; nndet/training/optimizer/utils.py:31-72; learning_rate.py:127-184).

When nnDetection itself is importable, `register_with_nndet()` registers `RetinaUNetV001AMD` in its
MODULE_REGISTRY (the `additional_imports` route); the standalone `build_model` is what
bench.py and the tests use (nnDetection's Lightning stack is not installed on the benchmark box).
"""
import copy
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .arch import (Generator, ConvInstanceRelu, ConvGroupRelu, StackedConvBlock2, Encoder, UFPNModular,
                   BCECLassifier, GIoURegressor, DetectionHeadHNMNative, DiCESegmenterFgBg)
from .core.boxes import ATSSMatcher, HardNegativeSamplerBatched, BoxCoderND, get_anchor_generator, box_iou
from .core.retina import BaseRetinaNet
from .plans import MODEL_CFG_V001, TRAINER_CFG_V001


class RetinaUNetV001:
    """Class attributes = the reference's extension points (retinaunet/base.py:74-85, v001.py:30-38)."""
    base_conv_cls = ConvInstanceRelu
    head_conv_cls = ConvGroupRelu
    block = StackedConvBlock2
    encoder_cls = Encoder
    decoder_cls = UFPNModular
    matcher_cls = ATSSMatcher
    head_cls = DetectionHeadHNMNative
    head_classifier_cls = BCECLassifier
    head_regressor_cls = GIoURegressor
    head_sampler_cls = HardNegativeSamplerBatched
    segmenter_cls = DiCESegmenterFgBg

    @classmethod
    def from_config_plan(cls, model_cfg: dict, plan_arch: dict, plan_anchors: dict, log_num_anchors: str = None, **kwargs):
        plan_arch = dict(plan_arch); plan_anchors = dict(plan_anchors)
        plan_arch.update(model_cfg.get("plan_arch_overwrites", {}))
        plan_anchors.update(model_cfg.get("plan_anchors_overwrites", {}))
        dim = plan_arch["dim"]
        coder = BoxCoderND(weights=(1.,) * (dim * 2))
        s_param = not (("aspect_ratios" in plan_anchors) and (plan_anchors["aspect_ratios"] is not None))
        anchor_generator = get_anchor_generator(dim, s_param=s_param)(**copy.deepcopy(plan_anchors))
        conv = Generator(cls.base_conv_cls, dim)
        encoder = cls.encoder_cls(conv=conv, conv_kernels=plan_arch["conv_kernels"], strides=plan_arch["strides"],
                                  block_cls=cls.block, in_channels=plan_arch["in_channels"],
                                  start_channels=plan_arch["start_channels"], stage_kwargs=None,
                                  max_channels=plan_arch.get("max_channels", 320), **model_cfg["encoder_kwargs"])
        decoder = cls.decoder_cls(conv=conv, conv_kernels=plan_arch["conv_kernels"], strides=encoder.get_strides(),
                                  in_channels=encoder.get_channels(), decoder_levels=plan_arch["decoder_levels"],
                                  fixed_out_channels=plan_arch["fpn_channels"], **model_cfg["decoder_kwargs"])
        matcher = cls.matcher_cls(similarity_fn=box_iou, **model_cfg["matcher_kwargs"])
        hconv = Generator(cls.head_conv_cls, dim)
        A = anchor_generator.num_anchors_per_location()[0]
        classifier = cls.head_classifier_cls(conv=hconv, in_channels=plan_arch["fpn_channels"],
                                             internal_channels=plan_arch["head_channels"],
                                             num_classes=plan_arch["classifier_classes"], anchors_per_pos=A,
                                             num_levels=len(plan_arch["decoder_levels"]), **model_cfg["head_classifier_kwargs"])
        regressor = cls.head_regressor_cls(conv=hconv, in_channels=plan_arch["fpn_channels"],
                                           internal_channels=plan_arch["head_channels"], anchors_per_pos=A,
                                           num_levels=len(plan_arch["decoder_levels"]), **model_cfg["head_regressor_kwargs"])
        sampler = cls.head_sampler_cls(**model_cfg["head_sampler_kwargs"])
        head = cls.head_cls(classifier=classifier, regressor=regressor, coder=coder, sampler=sampler,
                            log_num_anchors=None, **model_cfg["head_kwargs"])
        segmenter = None
        if cls.segmenter_cls is not None:
            segmenter = cls.segmenter_cls(Generator(cls.base_conv_cls, dim), seg_classes=plan_arch["seg_classes"],
                                          in_channels=decoder.get_channels(), decoder_levels=plan_arch["decoder_levels"],
                                          **model_cfg["segmenter_kwargs"])
        return BaseRetinaNet(
            dim=dim, encoder=encoder, decoder=decoder, head=head, anchor_generator=anchor_generator, matcher=matcher,
            num_classes=plan_arch["classifier_classes"], decoder_levels=plan_arch["decoder_levels"], segmenter=segmenter,
            detections_per_img=plan_arch.get("detections_per_img", 100), score_thresh=plan_arch.get("score_thresh", 0),
            topk_candidates=plan_arch.get("topk_candidates", 10000),
            remove_small_boxes=plan_arch.get("remove_small_boxes", 0.01), nms_thresh=plan_arch.get("nms_thresh", 0.6))


def build_model(plan: dict, model_cfg: dict = None) -> BaseRetinaNet:
    return RetinaUNetV001.from_config_plan(copy.deepcopy(model_cfg or MODEL_CFG_V001), plan["arch"], plan["anchors"])


NORM_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d,
              nn.LayerNorm, nn.GroupNorm, nn.SyncBatchNorm, nn.LocalResponseNorm)


def get_params_no_wd_on_norm(model: nn.Module, weight_decay: float) -> List[dict]:
    """No weight decay on norm parameters (nndet/training/optimizer/utils.py:31-72)."""
    decay, no_decay = [], []
    for module in model.modules():
        params = [p for p in module.parameters(recurse=False) if p.requires_grad]
        (no_decay if isinstance(module, NORM_TYPES) else decay).extend(params)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def configure_optimizer(model: nn.Module, trainer_cfg: dict = None, lean: bool = True):
    """SGD(nesterov) + linear warm-up -> poly decay, stepped per iteration (retinaunet/base.py:300-336).
    lean=True: the foreach-based implementation of nndetection_amd/optim.py (same arithmetic, far less host time);
    lean=False: torch.optim.SGD + LambdaLR."""
    cfg = dict(TRAINER_CFG_V001 if trainer_cfg is None else trainer_cfg)
    groups = get_params_no_wd_on_norm(model, cfg["weight_decay"])
    total = cfg["max_num_epochs"] * cfg["num_train_batches_per_epoch"]
    warm, warm_lr, lr0, gamma = cfg["warm_iterations"], cfg["warm_lr"], cfg["initial_lr"], cfg["poly_gamma"]
    if lean:
        from .optim import SGDNesterov, LinearWarmupPolyLR
        opt = SGDNesterov(groups, lr0, momentum=cfg["sgd_momentum"], nesterov=cfg["sgd_nesterov"])
        return opt, LinearWarmupPolyLR(opt, warm, warm_lr, gamma, total)
    opt = torch.optim.SGD(groups, lr0, weight_decay=cfg["weight_decay"], momentum=cfg["sgd_momentum"], nesterov=cfg["sgd_nesterov"])

    def factor(it: int) -> float:                 # it = number of scheduler steps so far; the reference uses k = it + 1
        k = it + 1
        if k - 1 < warm:
            return (warm_lr + (lr0 - warm_lr) * k / warm) / lr0
        return (1 - (k - warm) / max(1, total - warm)) ** gamma

    return opt, torch.optim.lr_scheduler.LambdaLR(opt, factor)


_AMD_CLASS_ATTRS = dict(
    base_conv_cls=ConvInstanceRelu, head_conv_cls=ConvGroupRelu, block=StackedConvBlock2, encoder_cls=Encoder,
    decoder_cls=UFPNModular, matcher_cls=ATSSMatcher, head_cls=DetectionHeadHNMNative, head_classifier_cls=BCECLassifier,
    head_regressor_cls=GIoURegressor, head_sampler_cls=HardNegativeSamplerBatched, segmenter_cls=DiCESegmenterFgBg)


class LazyFloat:
    """A Python-number-like view of ONE element of a device tensor that is being copied to pinned host memory asynchronously.
    The reference's `training_step` returns `l.detach().item()` per loss (retinaunet/base.py:154): four host synchronisations
    BETWEEN the forward and the backward pass, i.e. the backward pass is issued into an empty queue (measured 24.9 -> 23.3 ms per
    step when the last such sync left the step). Here all losses go to the host in one non-blocking copy; the value is
    read (event wait) when somebody actually converts it -- `float()`, `np.mean([...])` in `training_epoch_end`, a format string."""
    __slots__ = ("_host", "_i", "_ev", "_v")

    def __init__(self, host, i, ev):
        self._host, self._i, self._ev, self._v = host, i, ev, None

    def item(self) -> float:
        if self._v is None:
            if self._ev is not None:
                self._ev.synchronize()
            self._v = float(self._host[self._i])
            self._host = self._ev = None
        return self._v

    __float__ = item

    def __array__(self, dtype=None, copy=None):
        import numpy as np
        return np.asarray(self.item(), dtype=dtype or np.float64)

    def __repr__(self):
        return repr(self.item())

    def __format__(self, spec):
        return format(self.item(), spec)

    def __add__(self, o): return self.item() + o
    def __radd__(self, o): return o + self.item()
    def __sub__(self, o): return self.item() - o
    def __rsub__(self, o): return o - self.item()
    def __mul__(self, o): return self.item() * o
    def __rmul__(self, o): return o * self.item()
    def __truediv__(self, o): return self.item() / o
    def __rtruediv__(self, o): return o / self.item()
    def __neg__(self): return -self.item()
    def __lt__(self, o): return self.item() < float(o)
    def __le__(self, o): return self.item() <= float(o)
    def __gt__(self, o): return self.item() > float(o)
    def __ge__(self, o): return self.item() >= float(o)
    def __eq__(self, o): return self.item() == o
    def __hash__(self): return hash(self.item())


def lazy_items(values: Dict[str, torch.Tensor]) -> Dict[str, "LazyFloat"]:
    """{key: 0-dim tensor} -> {key: LazyFloat}: one stacked device tensor, one non-blocking copy to pinned memory, one event."""
    keys = list(values)
    if not keys:
        return {}
    dev = values[keys[0]].device
    stacked = torch.stack([values[k].detach().float().reshape(()) for k in keys])
    if dev.type != "cuda" or os.environ.get("NNDET_PLUGIN_LAZY_ITEMS", "1") == "0":
        host = stacked.cpu()
        return {k: LazyFloat(host, i, None) for i, k in enumerate(keys)}
    host = torch.empty(stacked.shape, dtype=torch.float32, pin_memory=True)
    host.copy_(stacked, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return {k: LazyFloat(host, i, ev) for i, k in enumerate(keys)}


_PRECISION_DTYPES = {16: torch.float16, "16": torch.float16, "16-mixed": torch.float16,
                     "bf16": torch.bfloat16, "bf16-mixed": torch.bfloat16}
_NAMED_DTYPES = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                 "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16, "float16": torch.float16}


class RetinaUNetAMDSteps:
    """The step bodies of the registered plugin module, free of any nnDetection / Lightning import so that the SAME code runs in
    `RetinaUNetV001AMD` (mixed in in front of the reference's `RetinaUNetV001`, `register_with_nndet`) and on the GPU box
    (`tests/test_plugin_gpu.py`, `bench.py --via-plugin`) where neither package exists. Expects `self.model` (our
    `BaseRetinaNet`), `self.trainer_cfg` (dict) and -- validation only -- `self.evaluation_step`.

    Replaces retinaunet/base.py:135-181 (`training_step`, `validation_step`) and the Lightning DDP pass-through of
    scripts/train.py:265-289 (`on_fit_start` / `on_after_backward`).

    Precision (scripts/train.py:277-278, conf/train/v001.yaml:32-33: `pl.Trainer(precision=16, amp_backend='native')`): Lightning's
    native AMP runs `training_step` inside `torch.autocast` (float16) and scales the loss with a GradScaler. The HIP kernels take
    their arithmetic type from the activation tensor, so the step casts `batch["data"]` to
      1. `NNDET_AMD_DTYPE` / `trainer_cfg["amd_dtype"]` (f32 | bf16 | f16) if given, else
      2. the autocast dtype when autocast is enabled (float16 under precision=16, bfloat16 under precision='bf16'), else
      3. bfloat16 when `trainer_cfg["precision"]` asks for 16-bit but no autocast context is active (no GradScaler can be assumed
         then: bf16 has fp32's exponent range and needs none), else
      4. the dtype of the batch (fp32: the exact-fp32 MFMA kernels).
    Parameters, gradients and the optimizer stay fp32 in every case (the packed low-precision weight copies are internal), which is
    exactly what autocast gives the reference. `amd_last_dtype` records what the last step ran in."""

    amd_last_dtype: Optional[torch.dtype] = None

    def amd_compute_dtype(self, data: torch.Tensor) -> torch.dtype:
        cfg = getattr(self, "trainer_cfg", None) or {}
        forced = os.environ.get("NNDET_AMD_DTYPE") or cfg.get("amd_dtype")
        if forced:
            try:
                return _NAMED_DTYPES[str(forced).lower()]
            except KeyError:
                raise ValueError(f"amd_dtype / NNDET_AMD_DTYPE must be one of {sorted(_NAMED_DTYPES)}, got {forced!r}")
        if data.is_cuda and torch.is_autocast_enabled():
            return _autocast_dtype()
        want = _PRECISION_DTYPES.get(cfg.get("precision", 32))
        if want is not None and data.is_cuda:
            return torch.bfloat16
        return data.dtype if data.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32

    def amd_prepare(self, batch, deferred: bool):
        from .core.targets import prepare_targets
        data = batch["data"]
        dt = self.amd_compute_dtype(data)
        self.amd_last_dtype = dt
        images, targets = prepare_targets(data, batch["target"], batch["instance_mapping"], deferred=deferred)
        return (images if images.dtype == dt else images.to(dt)), targets

    def training_step(self, batch, batch_idx):
        images, targets = self.amd_prepare(batch, deferred=True)
        with torch.autocast("cuda", enabled=False):        # the kernels already run in the chosen type; torch glue stays fp32
            losses, _ = self.model.train_step(images=images, targets=targets, evaluation=False, batch_num=batch_idx)
            loss = sum(losses.values())
        return {"loss": loss, **lazy_items(losses)}

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            images, targets = self.amd_prepare(batch, deferred=False)
            with torch.autocast("cuda", enabled=False):
                losses, prediction = self.model.train_step(images=images, targets=targets, evaluation=True, batch_num=batch_idx)
                loss = sum(losses.values())
        self.evaluation_step(prediction=prediction, targets=targets)
        return {"loss": loss.detach().item(), **{key: l.detach().item() for key, l in losses.items()}}

    # ---- data parallel: one process per GPU, gradient all-reduce over RCCL (nndetection_amd/ddp.py)
    def on_fit_start(self):
        import torch.distributed as dist
        self._amd_reducer = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not self._amd_under_lightning_ddp():
            from .ddp import GradAllReducer
            cfg = getattr(self, "trainer_cfg", None) or {}
            self._amd_reducer = GradAllReducer(self.model, **{k[len("amd_ddp_"):]: v for k, v in cfg.items() if k.startswith("amd_ddp_")})
        parent = getattr(super(), "on_fit_start", None)
        return parent() if parent is not None else None

    def _amd_under_lightning_ddp(self) -> bool:
        """True if somebody else already reduces the gradients: Lightning's own DDP strategy wraps the module in
        DistributedDataParallel (scripts/train.py:273-275 with accelerator='ddp'); a second all-reduce from our hooks would
        double the communication and swap `p.grad` behind DDP's back. `NNDET_AMD_DDP=1` forces ours, `=0` disables it."""
        sw = os.environ.get("NNDET_AMD_DDP")
        if sw is not None:
            return sw == "0"
        tr = getattr(self, "_trainer", None) or getattr(self, "trainer", None)
        for obj in (getattr(tr, "model", None), getattr(getattr(tr, "strategy", None), "model", None),
                    getattr(getattr(tr, "training_type_plugin", None), "model", None)):
            if isinstance(obj, torch.nn.parallel.DistributedDataParallel):
                return True
        name = type(getattr(tr, "strategy", None) or getattr(tr, "training_type_plugin", None)).__name__.lower()
        return "ddp" in name or "deepspeed" in name or "fsdp" in name

    def configure_optimizers(self):
        """The reference's `configure_optimizers` (retinaunet/base.py:300-336: SGD + nesterov, no weight decay on norms,
        LinearWarmupPolyLR per iteration) unchanged, then `amd_fuse_sgd` on every torch.optim.SGD it returned."""
        parent = getattr(super(), "configure_optimizers", None)
        res = parent() if parent is not None else configure_optimizer(self.model, getattr(self, "trainer_cfg", None), lean=True)
        opts = res[0] if isinstance(res, (tuple, list)) and res and isinstance(res[0], (tuple, list)) else [res]
        for o in opts:
            amd_fuse_sgd(o)
        return res

    def on_after_backward(self):
        red = getattr(self, "_amd_reducer", None)
        if red is not None:
            red.finish()
        parent = getattr(super(), "on_after_backward", None)
        return parent() if parent is not None else None


def amd_fuse_sgd(opt) -> bool:
    """Switch a `torch.optim.SGD` (as built by the reference) to torch's fused multi-tensor implementation when its parameters live
    on the GPU: one launch per parameter group instead of ~5 foreach passes, and -- under native AMP -- GradScaler hands a fused
    optimizer the scale / found-inf flag as device tensors instead of synchronising on `found_inf.item()` every step
    (`_step_supports_amp_scaling`). Same update rule (torch/optim/sgd.py); `NNDET_AMD_FUSED_OPT=0` leaves the optimizer alone."""
    if os.environ.get("NNDET_AMD_FUSED_OPT", "1") == "0" or not isinstance(opt, torch.optim.SGD):
        return False
    params = [p for g in opt.param_groups for p in g["params"]]
    if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params) or not hasattr(torch, "_fused_sgd_"):
        return False
    for g in opt.param_groups:
        if g.get("differentiable") or g.get("maximize"):
            return False
    for g in opt.param_groups:
        g["fused"], g["foreach"] = True, False
    opt.defaults["fused"], opt.defaults["foreach"] = True, False
    opt._step_supports_amp_scaling = True
    # torch._fused_sgd_ does not advance the parameters' version counters: invalidate everything cached from a parameter (packed /
    # cast weights, padded biases) by the step itself -- not only by the re-pack of the next training-mode forward
    if not getattr(opt, "_nndet_generation_hook", False):
        from .arch.conv import bump_param_generation
        opt.register_step_post_hook(bump_param_generation)
        opt._nndet_generation_hook = True
    # this edits a CONSTRUCTED optimizer (the reference builds it; we only see the instance): re-validate what torch.optim.SGD's
    # constructor would have checked for fused=True, also for groups added later
    if not getattr(opt, "_nndet_add_group_checked", False):
        import types
        opt.add_param_group = types.MethodType(_fused_add_param_group, opt)     # (a module-level function: nothing closes over the instance)
        opt._nndet_add_group_checked = True
    return True


def _fused_add_param_group(opt, group):
    """`add_param_group` of an optimizer that `amd_fuse_sgd` switched to fused=True: torch's own method first (its validation, its error
    types), then what torch.optim.SGD's constructor would have checked for a fused optimizer; a group that fails is taken out again."""
    n = len(opt.param_groups)
    res = type(opt).add_param_group(opt, group)
    for g in opt.param_groups[n:]:
        if not all(q.is_cuda and q.dtype == torch.float32 for q in g["params"]):
            del opt.param_groups[n:]
            raise ValueError("amd_fuse_sgd switched this optimizer to fused=True: new parameters must be fp32 tensors on the GPU")
    return res


def _autocast_dtype() -> torch.dtype:
    get = getattr(torch, "get_autocast_dtype", None)
    return get("cuda") if get is not None else torch.get_autocast_gpu_dtype()


class StandaloneRetinaUNetV001AMD(RetinaUNetAMDSteps, nn.Module):
    """The registered module without nnDetection / Lightning: same step bodies (`RetinaUNetAMDSteps`), same constructor arguments
    `(model_cfg, trainer_cfg, plan)` as `LightningBaseModule` (nndet/ptmodule/base_module.py:33-59), `self.model` built by the same
    `from_config_plan`. Used by `bench.py --via-plugin` and the GPU tests; a training loop drives it like Lightning drives the
    registered class: `training_step` -> `["loss"].backward()` -> `on_after_backward()` -> optimizer."""

    def __init__(self, model_cfg: dict, trainer_cfg: dict, plan: dict, **kwargs):
        super().__init__()
        self.model_cfg, self.trainer_cfg, self.plan = model_cfg, trainer_cfg, plan
        self.model = RetinaUNetV001.from_config_plan(copy.deepcopy(model_cfg), plan["architecture"], plan["anchors"])
        self.evaluated = []

    def evaluation_step(self, prediction: dict, targets: dict):
        self.evaluated.append((prediction, targets))


def register_with_nndet():
    """Register the HIP-backed module in nnDetection's MODULE_REGISTRY (requires nnDetection + Lightning to be importable).

    `RetinaUNetV001AMD` = `RetinaUNetAMDSteps` mixed in in front of the reference's `RetinaUNetV001`
    (nndet/ptmodule/retinaunet/v001.py:29-38), so Lightning hooks, evaluators, predictor / ensembler / sweep plumbing are inherited
    unchanged. It overrides
      * the component class attributes (retinaunet/base.py:74-85) -- the reference's own `from_config_plan` accepts them too;
      * `from_config_plan` (base.py:338-466): builds `nndetection_amd.core.retina.BaseRetinaNet`, i.e. also the detector core
        (batched ATSS assignment, fused post-processing) runs on the HIP kernels; same constructor calls, same state-dict keys;
      * `training_step` / `validation_step` (base.py:135-180) from the mixin: device-side target preparation instead of
        `self.pre_trafo`, activation dtype from the autocast state / `trainer_cfg.precision`, losses to the host without a sync;
      * `get_ensembler_cls("boxes", 3)`: the reference's BoxEnsemblerSelective with `postprocess_image` (the per-tile top-k / clip /
        small-box filter / model NMS, inference/ensembler/detection.py:166-217) as one fused HIP pass (inference/ensembler.py);
      * `on_fit_start` / `on_after_backward` from the mixin: the bucketed RCCL gradient all-reduce of nndetection_amd.ddp when the
        job runs as one process per GPU under torch.distributed and Lightning's own DDP wrapper is not active (the reference
        leaves multi-GPU to pl.Trainer, scripts/train.py:265-289).
    """
    from nndet.ptmodule import MODULE_REGISTRY
    from nndet.ptmodule.retinaunet.v001 import RetinaUNetV001 as _RefV001

    class RetinaUNetV001AMD(RetinaUNetAMDSteps, _RefV001):
        locals().update(_AMD_CLASS_ATTRS)

        @classmethod
        def from_config_plan(cls, model_cfg: dict, plan_arch: dict, plan_anchors: dict, log_num_anchors: str = None, **kwargs):
            return RetinaUNetV001.from_config_plan(model_cfg, plan_arch, plan_anchors, log_num_anchors, **kwargs)

        @staticmethod
        def get_ensembler_cls(key, dim: int):
            """retinaunet/base.py:677-695; the box ensembler's per-tile `postprocess_image` runs as one fused HIP pass."""
            base = _RefV001.get_ensembler_cls(key, dim)
            if key == "boxes" and dim == 3 and base is not None:
                from .inference.ensembler import amd_box_ensembler
                return amd_box_ensembler(base)
            return base

    if "RetinaUNetV001AMD" not in MODULE_REGISTRY.mapping:
        MODULE_REGISTRY.register(RetinaUNetV001AMD)
    import nndet.core.boxes.nms as _ref_nms
    from .core.boxes.nms import nms_gpu
    _ref_nms.nms_gpu = nms_gpu          # the reference resolves this name at call time (nms.py:74-78)
    return MODULE_REGISTRY["RetinaUNetV001AMD"]
