"""ctypes binding of oracle/_build/liboracle_vq.so (oracle/vq_argmin.c).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "_build", "liboracle_vq.so")
    src = os.path.join(_HERE, "vq_argmin.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_vq_argmin_f32.restype = None
        _LIB.oracle_vq_argmin_f32.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p]
    return _LIB


def vq_argmin_f32(z_flat, codebook, want_distances=False, want_quantized=False):
    """z_flat (N,D) f32, codebook (K,D) f32 -> dict(indices int64 (N,), distances, quantized, sq_err)."""
    z = np.ascontiguousarray(z_flat, dtype=np.float32)
    e = np.ascontiguousarray(codebook, dtype=np.float32)
    N, D = z.shape
    K = e.shape[0]
    idx = np.empty(N, dtype=np.int64)
    dist = np.empty((N, K), dtype=np.float32) if want_distances else None
    q = np.empty((N, D), dtype=np.float32) if want_quantized else None
    sq = ctypes.c_double(0.0)
    lib().oracle_vq_argmin_f32(z.ctypes.data, N, D, e.ctypes.data, K, idx.ctypes.data,
                               dist.ctypes.data if dist is not None else None,
                               q.ctypes.data if q is not None else None, ctypes.byref(sq))
    return dict(indices=idx, distances=dist, quantized=q, sq_err=sq.value)
