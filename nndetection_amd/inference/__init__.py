"""GPU side of nnDetection's inference ensembling: the functions the reference's ensemblers plug in as
`model_nms_fn` / `ensemble_nms_fn` (nndet/inference/ensembler/detection.py:40-75,476-537)."""
from .detection import batched_nms_model, batched_weighted_nms_model, batched_wbc, wbc, batched_wbc_ensemble  # noqa: F401
from .ensembler import postprocess_image_fused, AMDPostprocessMixin, amd_box_ensembler  # noqa: F401
