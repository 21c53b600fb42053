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


class _Stub:
    """Placeholder for any attribute of a stubbed third-party module: callable, subclassable, iterable-empty."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Stub()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub()

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    __path__ = []          # behaves as a package: any submodule import is served by the finder below

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Stub,), {})


class _StubFinder:
    """Serves `import X[.y.z]` for the third-party packages of STUB_ROOTS that are not installed in this image and that the
    reference imports at module level along `nndet.ptmodule` (IO / evaluation / experiment tracking: none of them is on the
    hot path). Installed packages are never shadowed."""

    def __init__(self, roots):
        self.roots = set(roots)

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery
        if fullname.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


STUB_ROOTS = ("SimpleITK", "batchgenerators", "nnunet", "mlflow", "hydra", "nevergrad", "seaborn", "matplotlib",
              "torchmetrics", "medpy", "skimage")


def install_stub_finder():
    import importlib.util
    missing = [r for r in STUB_ROOTS if r not in sys.modules and importlib.util.find_spec(r) is None]
    if missing and not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder(missing))
    return missing


def load_reference_ptmodule():
    """`nndet.ptmodule` (MODULE_REGISTRY, RetinaUNetModule, RetinaUNetV001) of the unmodified reference: needs the
    pytorch_lightning stand-in of oracle/_shims plus inert stubs for the IO / evaluation third-party packages."""
    load_reference()
    install_stub_finder()
    import nndet.ptmodule
    return nndet.ptmodule


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
