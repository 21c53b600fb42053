"""Import the UNMODIFIED reference (MIC-DKFZ/nnDetection at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY. Used by `tests/golden/make_golden.py` (run in the build container,
where /root/reference exists) to pin the oracle restatements in `oracle/` against the real
reference. Nothing in the product, `bench.py`, `smoke()` or the `-m gpu` tests may call this:
/root/reference does not exist on the GPU box.

Recipe follows SURVEY.md Appendix C: a ~40-line stub layer (`oracle/_shims`) for third-party
packages that are not installed here (loguru, omegaconf, GitPython, four torchvision symbols)
plus the removed `torch._six` module.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("NNDET_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nndet"))


def load_reference():
    """Put the shims + reference on sys.path and return the imported `nndet` package."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    import torch
    if "torch._six" not in sys.modules:
        six = types.ModuleType("torch._six")
        six.string_classes = (str,)
        sys.modules["torch._six"] = six
        torch._six = six
    for p in (REFERENCE_ROOT, _SHIMS):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nndet  # noqa: F401
    return nndet
