"""ctypes binding of oracle/net_loops.c (scalar-loop restatement of affine / LSTM / Gaussian / num-steps / RMSProp) --
TEST INFRASTRUCTURE ONLY: never imported by the product."""
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
        path = os.path.join(_HERE, "_build", f"net_loops_{key}.so")
        if not os.path.exists(path):
            build()
        _LIBS[key] = ctypes.CDLL(path)
    return _LIBS[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _real(dtype):
    return ctypes.c_float if np.dtype(dtype) == np.float32 else ctypes.c_double


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def affine(x, w, b, elu):
    dt = x.dtype
    x, w, b = _c(x, dt), _c(w, dt), _c(b, dt)
    M, K = x.shape
    N = w.shape[1]
    y = np.empty((M, N), dt)
    _lib(dt).affine(_p(x), _p(w), _p(b), _p(y), M, K, N, int(bool(elu)))
    return y


def lstm_step(x, h, c, w, b, forget_bias=1.0):
    dt = x.dtype
    x, h, c, w, b = (_c(a, dt) for a in (x, h, c, w, b))
    M, I = x.shape
    H = h.shape[1]
    h2, c2 = np.empty((M, H), dt), np.empty((M, H), dt)
    _lib(dt).lstm_step(_p(x), _p(h), _p(c), _p(w), _p(b), _p(h2), _p(c2), M, I, H, _real(dt)(forget_bias))
    return h2, c2


def gauss_sample_kl(pre, eps, offset, prior_loc, prior_scale):
    dt = pre.dtype
    pre, eps = _c(pre, dt), _c(eps, dt)
    M, D = eps.shape
    loc, scale, sample, kl = np.empty((M, D), dt), np.empty((M, D), dt), np.empty((M, D), dt), np.empty((M,), dt)
    R = _real(dt)
    _lib(dt).gauss_sample_kl(_p(pre), _p(eps), R(offset), R(prior_loc), R(prior_scale), _p(loc), _p(scale), _p(sample),
                             _p(kl), M, D)
    return loc, scale, sample, kl


def numsteps_posterior(p):
    p = _c(p, np.float64)
    B, T = p.shape
    q = np.empty((B, T + 1), np.float64)
    _lib(np.float64).numsteps_posterior(_p(p), _p(q), B, T)
    return q


def geometric_prior(success_prob, T):
    pi = np.empty((T + 1,), np.float64)
    _lib(np.float64).geometric_prior(ctypes.c_double(success_prob), _p(pi), T)
    return pi


def tabular_kl(q, pi):
    q, pi = _c(q, np.float64), _c(pi, np.float64)
    B, T1 = q.shape
    kl = np.empty((B,), np.float64)
    _lib(np.float64).tabular_kl(_p(q), _p(pi), _p(kl), B, T1 - 1)
    return kl


def rmsprop_centered(p, g, ms, mg, mom, lr, decay, momentum, eps):
    """in place on p, ms, mg, mom"""
    dt = p.dtype
    R = _real(dt)
    _lib(dt).rmsprop_centered(_p(p), _p(_c(g, dt)), _p(ms), _p(mg), _p(mom), ctypes.c_long(p.size), R(lr), R(decay),
                              R(momentum), R(eps))
