"""Import hook for nnDetection's `additional_imports` (nndet/utils/config.py:66-68): importing this module registers
`RetinaUNetV001AMD` in MODULE_REGISTRY and points `nndet.core.boxes.nms.nms_gpu` at the HIP NMS."""
from .ptmodule import register_with_nndet

RetinaUNetV001AMD = register_with_nndet()
