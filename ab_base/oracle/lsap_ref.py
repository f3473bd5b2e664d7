"""Loader of oracle/lsap.c (TEST INFRASTRUCTURE; see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_lsap.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_lsap.restype = C.c_int
        _lib.oracle_lsap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def linear_sum_assignment(cost):
    """Same contract as scipy's on a float32 matrix: (row_ind, col_ind) int64, ValueError on
    NaN/-inf ("invalid numeric entries") or on an infeasible matrix."""
    a = np.ascontiguousarray(np.asarray(cost, dtype=np.float32))
    nr, nc = a.shape
    n = min(nr, nc)
    row = np.empty(n, dtype=np.int64)
    col = np.empty(n, dtype=np.int64)
    rc = _load().oracle_lsap(a.ctypes.data, nr, nc, row.ctypes.data, col.ctypes.data)
    if rc == -3:
        raise ValueError("matrix contains invalid numeric entries")
    if rc == -4:
        raise ValueError("cost matrix is infeasible")
    return row, col
