"""ctypes loader for oracle/spmm_ref.c (TEST INFRASTRUCTURE ONLY -- see the
header of gae_oracle.py for who may import this)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgae_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "spmm_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-shared", "-fPIC",
                               "-o", _SO, src])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def spmm_csr(indptr, indices, H, row_scale=None, col_scale=None, out=None):
    """``out``: preallocated [n, F] fp32 result (timed loops: keeps the page faults of a fresh array out)"""
    H = np.ascontiguousarray(H, dtype=np.float32)
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n, F = len(indptr) - 1, H.shape[1]
    M = np.empty((n, F), dtype=np.float32) if out is None else out
    rs = None if row_scale is None else np.ascontiguousarray(row_scale, dtype=np.float32).ravel()
    cs = None if col_scale is None else np.ascontiguousarray(col_scale, dtype=np.float32).ravel()
    lib().oracle_spmm_csr_f32(ctypes.c_int64(n), _p(indptr, ctypes.c_int32), _p(indices, ctypes.c_int32),
                              _p(H, ctypes.c_float), ctypes.c_int64(F), _p(M, ctypes.c_float),
                              ctypes.c_int64(F), ctypes.c_int64(F), _p(rs, ctypes.c_float),
                              _p(cs, ctypes.c_float))
    return M


def spmm_csr_acc64(indptr, indices, H):
    """un-normalised aggregate with a double accumulator (fp32 H, fp64 result)"""
    H = np.ascontiguousarray(H, dtype=np.float32)
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n, F = len(indptr) - 1, H.shape[1]
    M = np.empty((n, F), dtype=np.float64)
    lib().oracle_spmm_csr_f32_acc64(ctypes.c_int64(n), _p(indptr, ctypes.c_int32), _p(indices, ctypes.c_int32),
                                    _p(H, ctypes.c_float), ctypes.c_int64(F), _p(M, ctypes.c_double),
                                    ctypes.c_int64(F), ctypes.c_int64(F))
    return M


def linear(M, W, b, relu):
    M = np.ascontiguousarray(M, dtype=np.float32); W = np.ascontiguousarray(W, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    Y = np.empty((M.shape[0], W.shape[0]), dtype=np.float32)
    lib().oracle_linear_f32(ctypes.c_int64(M.shape[0]), _p(M, ctypes.c_float), ctypes.c_int64(M.shape[1]),
                            _p(W, ctypes.c_float), _p(b, ctypes.c_float), ctypes.c_int64(W.shape[0]),
                            ctypes.c_int(int(relu)), _p(Y, ctypes.c_float))
    return Y


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


def num_threads():
    return int(lib().oracle_num_threads())
