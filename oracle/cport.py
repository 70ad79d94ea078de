"""ctypes front-end of oracle/unproject_oracle.c (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmvhmr_oracle.so")
METHODS = {"softmax": 0, "sum": 1, "mean": 2, "max": 3}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "unproject_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libmvhmr_oracle.so"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        fp = ctypes.POINTER(ctypes.c_float)
        i, l = ctypes.c_int, ctypes.c_int64
        _lib.mvhmr_oracle_unproject_forward.argtypes = [fp, fp, fp, fp, i, i, i, i, i, l, i]
        _lib.mvhmr_oracle_unproject_forward.restype = i
        _lib.mvhmr_oracle_unproject_backward.argtypes = [fp, fp, fp, fp, fp, i, i, i, i, i, l, i]
        _lib.mvhmr_oracle_unproject_backward.restype = i
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward(features, proj, coords, method="softmax"):
    """features (B,V,C,Hf,Wf), proj (B,V,3,4), coords (B,X,Y,Z,3) -> (B,C,X,Y,Z); numpy fp32."""
    features, pf = _f32(features)
    proj, pp = _f32(proj)
    coords, pc = _f32(coords)
    B, V, C, Hf, Wf = features.shape
    vol = coords.shape[1:4]
    N = int(np.prod(vol))
    out = np.empty((B, C) + tuple(vol), dtype=np.float32)
    rc = lib().mvhmr_oracle_unproject_forward(pf, pp, pc, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                              B, V, C, Hf, Wf, N, METHODS[method])
    if rc != 0:
        raise ValueError("oracle forward rejected its arguments (rc=%d)" % rc)
    return out


def backward(grad_out, features, proj, coords, method="softmax"):
    grad_out, pg = _f32(grad_out)
    features, pf = _f32(features)
    proj, pp = _f32(proj)
    coords, pc = _f32(coords)
    B, V, C, Hf, Wf = features.shape
    N = int(np.prod(coords.shape[1:4]))
    gf = np.empty_like(features)
    rc = lib().mvhmr_oracle_unproject_backward(pg, pf, pp, pc, gf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                               B, V, C, Hf, Wf, N, METHODS[method])
    if rc != 0:
        raise ValueError("oracle backward rejected its arguments (rc=%d)" % rc)
    return gf
