"""ctypes wrapper of oracle/nms_ref.c (CPU oracle / cpu_baseline for large N). TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_nms.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_nms3d.restype = ctypes.c_int64
        _lib.oracle_nms3d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    return _lib


def nms(boxes, scores, thr: float) -> np.ndarray:
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    s = np.ascontiguousarray(scores, np.float32)
    n = b.shape[0]
    order = np.ascontiguousarray(np.argsort(-s, kind="stable"), np.int64)
    keep = np.empty(max(n, 1), np.int64)
    nk = _load().oracle_nms3d(b.ctypes.data, order.ctypes.data, n, float(thr), keep.ctypes.data)
    return keep[:nk].copy()
