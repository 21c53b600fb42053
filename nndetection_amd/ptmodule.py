"""Plugin surface: `RetinaUNetV001` built from a plan exactly like the reference's
`RetinaUNetModule.from_config_plan` (nndet/ptmodule/retinaunet/base.py:338-675, v001.py:29-38), with every
class attribute pointing at the HIP-backed implementation. Also the optimizer / LR schedule of
`configure_optimizers` (base.py:300-336; nndet/training/optimizer/utils.py:31-72; learning_rate.py:127-184).

When nnDetection itself is importable, `register_with_nndet()` registers `RetinaUNetV001AMD` in its
MODULE_REGISTRY (the `additional_imports` route); the standalone `build_model` is what
bench.py and the tests use (nnDetection's Lightning stack is not installed on the benchmark box).
"""
import copy
from typing import Dict, List

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


def register_with_nndet():
    """Register the HIP-backed module in nnDetection's MODULE_REGISTRY (requires nnDetection + Lightning to be importable).

    `RetinaUNetV001AMD` subclasses the reference's `RetinaUNetV001` (nndet/ptmodule/retinaunet/v001.py:29-38), so Lightning
    hooks, evaluators, predictor / ensembler / sweep plumbing are inherited unchanged. It overrides
      * the component class attributes (retinaunet/base.py:74-85) -- the reference's own `from_config_plan` accepts them too;
      * `from_config_plan` (base.py:338-466): builds `nndetection_amd.core.retina.BaseRetinaNet`, i.e. also the detector core
        (batched ATSS assignment, fused post-processing) runs on the HIP kernels; same constructor calls, same state-dict keys;
      * `training_step` / `validation_step` (base.py:135-180): device-side target preparation instead of `self.pre_trafo`;
      * `on_fit_start` / `on_after_backward`: the bucketed RCCL gradient all-reduce of nndetection_amd.ddp when the job runs as
        one process per GPU under torch.distributed (the reference leaves multi-GPU to pl.Trainer, scripts/train.py:265-289).
    """
    from nndet.ptmodule import MODULE_REGISTRY
    from nndet.ptmodule.retinaunet.v001 import RetinaUNetV001 as _RefV001

    class RetinaUNetV001AMD(_RefV001):
        locals().update(_AMD_CLASS_ATTRS)

        @classmethod
        def from_config_plan(cls, model_cfg: dict, plan_arch: dict, plan_anchors: dict, log_num_anchors: str = None, **kwargs):
            return RetinaUNetV001.from_config_plan(model_cfg, plan_arch, plan_anchors, log_num_anchors, **kwargs)

        def _prepare(self, batch):
            from .core.targets import prepare_targets
            return prepare_targets(batch["data"], batch["target"], batch["instance_mapping"])

        def training_step(self, batch, batch_idx):
            images, targets = self._prepare(batch)
            losses, _ = self.model.train_step(images=images, targets=targets, evaluation=False, batch_num=batch_idx)
            loss = sum(losses.values())
            return {"loss": loss, **{key: l.detach().item() for key, l in losses.items()}}

        def validation_step(self, batch, batch_idx):
            with torch.no_grad():
                images, targets = self._prepare(batch)
                losses, prediction = self.model.train_step(images=images, targets=targets, evaluation=True, batch_num=batch_idx)
                loss = sum(losses.values())
            self.evaluation_step(prediction=prediction, targets=targets)
            return {"loss": loss.detach().item(), **{key: l.detach().item() for key, l in losses.items()}}

        # ---- data parallel: one process per GPU, gradient all-reduce over RCCL (nndetection_amd/ddp.py)
        def on_fit_start(self):
            import torch.distributed as dist
            self._amd_reducer = None
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from .ddp import GradAllReducer
                self._amd_reducer = GradAllReducer(self.model)
            parent = getattr(super(), "on_fit_start", None)
            return parent() if parent is not None else None

        def on_after_backward(self):
            red = getattr(self, "_amd_reducer", None)
            if red is not None:
                red.finish()
            parent = getattr(super(), "on_after_backward", None)
            return parent() if parent is not None else None

    if "RetinaUNetV001AMD" not in MODULE_REGISTRY.mapping:
        MODULE_REGISTRY.register(RetinaUNetV001AMD)
    import nndet.core.boxes.nms as _ref_nms
    from .core.boxes.nms import nms_gpu
    _ref_nms.nms_gpu = nms_gpu          # the reference resolves this name at call time (nms.py:74-78)
    return MODULE_REGISTRY["RetinaUNetV001AMD"]
