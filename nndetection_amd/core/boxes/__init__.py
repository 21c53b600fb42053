"""Box ops of the hot path, same public names as `nndet.core.boxes` (nndet/core/boxes/__init__.py:1-21)."""
from .ops import box_iou, generalized_box_iou, giou_diag, remove_small_boxes, box_center  # noqa: F401
from .nms import nms, batched_nms  # noqa: F401
from .anchors import AnchorGenerator3DS, get_anchor_generator  # noqa: F401
from .matcher import ATSSMatcher, IoUMatcher  # noqa: F401
from .sampler import HardNegativeSamplerBatched  # noqa: F401
from .coder import BoxCoderND, decode_single, BBOX_XFORM_CLIP  # noqa: F401
from .postprocess import postprocess_batch, postprocess_batch_raw  # noqa: F401
from .clip import clip_boxes_to_image_  # noqa: F401
