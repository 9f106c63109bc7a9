"""ctypes binding of oracle/st_loops.c (scalar-loop ST restatement) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib(dtype):
    key = "f32" if np.dtype(dtype) == np.float32 else "f64"
    if key not in _LIBS:
        path = os.path.join(_HERE, "_build", f"st_loops_{key}.so")
        if not os.path.exists(path):
            build()
        _LIBS[key] = ctypes.CDLL(path)
    return _LIBS[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def st_read_fwd(img, where, crop):
    img = np.ascontiguousarray(img); where = np.ascontiguousarray(where, dtype=img.dtype)
    B, H, W = img.shape; h, w = crop
    out = np.empty((B, h, w), img.dtype)
    _lib(img.dtype).st_read_fwd(_p(img), _p(where), _p(out), B, H, W, h, w)
    return out


def st_read_bwd(img, where, dout, want_dimg=True):
    img = np.ascontiguousarray(img); where = np.ascontiguousarray(where, dtype=img.dtype)
    dout = np.ascontiguousarray(dout, dtype=img.dtype)
    B, H, W = img.shape; h, w = dout.shape[1:]
    dwhere = np.empty((B, 4), img.dtype)
    dimg = np.empty_like(img) if want_dimg else None
    _lib(img.dtype).st_read_bwd(_p(img), _p(where), _p(dout), _p(dwhere), _p(dimg), B, H, W, h, w)
    return dwhere, dimg


def st_write_fwd(glm, where, img_size):
    glm = np.ascontiguousarray(glm); where = np.ascontiguousarray(where, dtype=glm.dtype)
    B, h, w = glm.shape; H, W = img_size
    out = np.empty((B, H, W), glm.dtype)
    _lib(glm.dtype).st_write_fwd(_p(glm), _p(where), _p(out), B, H, W, h, w)
    return out


def st_write_bwd(glm, where, dout):
    glm = np.ascontiguousarray(glm); where = np.ascontiguousarray(where, dtype=glm.dtype)
    dout = np.ascontiguousarray(dout, dtype=glm.dtype)
    B, h, w = glm.shape; H, W = dout.shape[1:]
    dglm = np.empty_like(glm); dwhere = np.empty((B, 4), glm.dtype)
    _lib(glm.dtype).st_write_bwd(_p(glm), _p(where), _p(dout), _p(dglm), _p(dwhere), B, H, W, h, w)
    return dglm, dwhere
