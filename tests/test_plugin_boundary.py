"""CPU: the plugin boundary B1 / a19 exercised against the UNMODIFIED reference package: `nndet.ptmodule` is
imported from /root/reference (pytorch_lightning stand-in + inert stubs for IO-only third-party packages, oracle/refimport.py),
`nndetection_amd.plugin` is imported the way `cfg.additional_imports` does it (nndet/utils/config.py:66-68) and the
registered module is built through the reference's own constructor path. Skipped where /root/reference does not exist
(the GPU box)."""
import copy

import pytest
import torch

from oracle.refimport import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="needs the reference checkout at /root/reference")


@pytest.fixture(scope="module")
def ref():
    from oracle.refimport import load_reference_ptmodule
    return load_reference_ptmodule()


def _plan(name="tiny"):
    from nndetection_amd.plans import get_plan, MODEL_CFG_V001, TRAINER_CFG_V001
    p = get_plan(name)
    plan = {"architecture": p["arch"], "anchors": p["anchors"], "patch_size": p["patch_size"], "batch_size": p["batch_size"]}
    trainer_cfg = dict(TRAINER_CFG_V001, swa_epochs=10)
    return copy.deepcopy(MODEL_CFG_V001), trainer_cfg, plan


def test_additional_import_registers_module_and_nms(ref):
    import importlib
    plugin = importlib.import_module("nndetection_amd.plugin")
    assert ref.MODULE_REGISTRY["RetinaUNetV001AMD"] is plugin.RetinaUNetV001AMD
    from nndet.ptmodule.retinaunet.v001 import RetinaUNetV001 as RefV001
    assert issubclass(plugin.RetinaUNetV001AMD, RefV001)
    import nndet.core.boxes.nms as ref_nms
    from nndetection_amd.core.boxes.nms import nms_gpu
    assert ref_nms.nms_gpu is nms_gpu                      # what nndet.core.boxes.nms.nms dispatches to (nms.py:74-78)
    with pytest.raises(TypeError):                         # registry contract: duplicate names are rejected (registry.py:28-31)
        ref.MODULE_REGISTRY.register(plugin.RetinaUNetV001AMD)


def test_registered_class_runs_the_reference_free_step_mixin(ref):
    """a19: the step bodies the GPU tests and `bench.py --via-plugin` exercise on the box (tests/test_plugin_gpu.py, where
    nnDetection does not exist) ARE the ones the registered class runs: `RetinaUNetAMDSteps` sits in front of the reference's
    RetinaUNetV001 in the MRO and provides training_step / validation_step / configure_optimizers / the two DDP hooks."""
    import nndetection_amd.plugin as plugin
    from nndetection_amd.ptmodule import RetinaUNetAMDSteps, StandaloneRetinaUNetV001AMD
    from nndet.ptmodule.retinaunet.v001 import RetinaUNetV001 as RefV001
    cls = plugin.RetinaUNetV001AMD
    mro = cls.__mro__
    assert mro.index(RetinaUNetAMDSteps) < mro.index(RefV001)
    for name in ("training_step", "validation_step", "configure_optimizers", "on_fit_start", "on_after_backward", "amd_compute_dtype"):
        assert getattr(cls, name) is getattr(RetinaUNetAMDSteps, name), name
        assert getattr(StandaloneRetinaUNetV001AMD, name) is getattr(RetinaUNetAMDSteps, name), name
    model_cfg, trainer_cfg, plan = _plan("tiny")
    amd = cls(model_cfg, dict(trainer_cfg, precision=16), plan)
    assert amd.amd_compute_dtype(torch.zeros(1)) == torch.float32            # CPU batch: untouched; the CUDA mapping is a GPU test
    (opt,), sched = amd.configure_optimizers()                              # the reference's optimizer, fused only on the GPU
    assert isinstance(opt, torch.optim.SGD) and not opt.param_groups[0].get("fused")


def test_registered_class_hands_out_the_fused_box_ensembler(ref):
    """f3 remainder: `get_ensembler_cls("boxes", 3)` (retinaunet/base.py:677-695, used by get_predictor :724)
    is the reference's BoxEnsemblerSelective with `postprocess_image` replaced by the fused HIP pass; the segmentation ensembler
    and everything else of the class are the reference's."""
    import nndetection_amd.plugin as plugin
    from nndetection_amd.inference.ensembler import AMDPostprocessMixin
    from nndet.inference.ensembler.detection import BoxEnsemblerSelective
    from nndet.inference.ensembler.segmentation import SegmentationEnsembler
    cls = plugin.RetinaUNetV001AMD.get_ensembler_cls("boxes", 3)
    assert issubclass(cls, BoxEnsemblerSelective) and issubclass(cls, AMDPostprocessMixin)
    assert cls.postprocess_image is AMDPostprocessMixin.postprocess_image
    assert cls.from_case.__func__ is BoxEnsemblerSelective.from_case.__func__ and cls.get_default_parameters() == BoxEnsemblerSelective.get_default_parameters()
    assert plugin.RetinaUNetV001AMD.get_ensembler_cls("seg", 3) is SegmentationEnsembler
    assert plugin.RetinaUNetV001AMD.get_ensembler_cls("boxes", 3) is cls           # one subclass, created once


def test_module_builds_with_reference_state_dict(ref):
    """Constructed like scripts/train.py:237 does (`MODULE_REGISTRY[cfg.module](model_cfg, trainer_cfg, plan)`): the 92
    parameter names / shapes equal those of the reference module, so reference checkpoints load strictly."""
    import nndetection_amd.plugin  # noqa: F401
    from nndet.ptmodule.retinaunet.v001 import RetinaUNetV001 as RefV001
    model_cfg, trainer_cfg, plan = _plan("luna160")
    amd = ref.MODULE_REGISTRY["RetinaUNetV001AMD"](copy.deepcopy(model_cfg), trainer_cfg, copy.deepcopy(plan))
    refm = RefV001(copy.deepcopy(model_cfg), trainer_cfg, copy.deepcopy(plan))
    sa, sr = amd.state_dict(), refm.state_dict()
    assert list(sa.keys()) == list(sr.keys()) and len(sa) == 92
    assert all(sa[k].shape == sr[k].shape and sa[k].dtype == sr[k].dtype for k in sa)
    amd.load_state_dict(sr, strict=True)
    from nndetection_amd.core.retina import BaseRetinaNet
    assert isinstance(amd.model, BaseRetinaNet)            # our from_config_plan: detector core on the HIP path too
    assert amd.max_epochs == trainer_cfg["max_num_epochs"] + trainer_cfg["swa_epochs"]


def test_reference_builder_accepts_our_components(ref):
    """B2: the reference's OWN `RetinaUNetModule.from_config_plan` (retinaunet/base.py:338-466, incl. its _build_* helpers)
    with only the class attributes swapped constructs our encoder / decoder / heads / segmenter / matcher / sampler."""
    import nndetection_amd.plugin  # noqa: F401
    from nndet.ptmodule.retinaunet.base import RetinaUNetModule
    from nndet.core.retina import BaseRetinaNet as RefNet
    from nndetection_amd import arch
    cls = ref.MODULE_REGISTRY["RetinaUNetV001AMD"]
    model_cfg, _, plan = _plan("tiny")
    net = RetinaUNetModule.from_config_plan.__func__(cls, model_cfg, copy.deepcopy(plan["architecture"]), copy.deepcopy(plan["anchors"]))
    assert isinstance(net, RefNet)
    assert isinstance(net.encoder, arch.Encoder) and isinstance(net.decoder, arch.UFPNModular)
    assert isinstance(net.head, arch.DetectionHeadHNMNative) and isinstance(net.segmenter, arch.DiCESegmenterFgBg)
    from nndetection_amd.ptmodule import build_model
    from nndetection_amd.plans import get_plan
    ours = build_model(get_plan("tiny"))
    assert list(net.state_dict().keys()) == list(ours.state_dict().keys())


def test_reference_configure_optimizers_on_our_module(ref):
    """a19: the reference's configure_optimizers (base.py:300-336) on our module: `get_params_no_wd_on_norm` finds our norm
    sub-modules (they ARE nn.InstanceNorm3d / nn.GroupNorm) -> 2 groups, no weight decay on exactly the norm parameters."""
    import nndetection_amd.plugin  # noqa: F401
    model_cfg, trainer_cfg, plan = _plan("tiny")
    amd = ref.MODULE_REGISTRY["RetinaUNetV001AMD"](model_cfg, trainer_cfg, plan)
    (opt,), sched = amd.configure_optimizers()
    assert isinstance(opt, torch.optim.SGD) and sched["interval"] == "step"
    n_norm = sum(p.numel() for m in amd.modules() if isinstance(m, (torch.nn.InstanceNorm3d, torch.nn.GroupNorm)) for p in m.parameters(recurse=False))
    no_wd = [g for g in opt.param_groups if g["weight_decay"] == 0.0]
    assert len(no_wd) == 1 and sum(p.numel() for p in no_wd[0]["params"]) == n_norm > 0
    assert sum(p.numel() for g in opt.param_groups for p in g["params"]) == sum(p.numel() for p in amd.parameters())


def test_ddp_hooks_present_and_inert_without_process_group(ref):
    """a20: the Lightning-facing hooks that drive nndetection_amd.ddp.GradAllReducer exist; without torch.distributed they are
    no-ops (single GPU)."""
    import nndetection_amd.plugin  # noqa: F401
    model_cfg, trainer_cfg, plan = _plan("tiny")
    amd = ref.MODULE_REGISTRY["RetinaUNetV001AMD"](model_cfg, trainer_cfg, plan)
    amd.on_fit_start()
    assert amd._amd_reducer is None
    amd.on_after_backward()
