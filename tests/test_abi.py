"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/nndet_amd.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from nndetection_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "nndet_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nndet_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    from nndetection_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nndet_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_identification(lib):
    assert lib.nndet_arch() == b"gfx950"
    assert b"nndetection_amd" in lib.nndet_version()


def test_struct_layout_matches_header():
    from nndetection_amd._lib import NndetConv
    assert ctypes.sizeof(NndetConv) == 4 * (13 + 9) + 8 + 4 + 4      # + in_affine pointer, in_relu, reserved


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU (no silent eager fallback)."""
    import torch
    from nndetection_amd._lib import NndetError
    from nndetection_amd.core.boxes import box_iou
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(NndetError):
        box_iou(torch.rand(3, 6), torch.rand(4, 6))


def test_oracle_not_imported_by_product():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import nndetection_amd.ptmodule, nndetection_amd.ddp; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; assert not bad, bad") % ROOT
    subprocess.check_call([sys.executable, "-c", code])
