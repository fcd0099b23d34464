"""ctypes front-end of oracle/c/oracle.c (plain-C oracle).  Test infrastructure only."""
import ctypes as C

import numpy as np

from . import build_c

_lib = None
_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_c.build())
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def gwc_volume(L, R, D, G):
    L, R = _a(L), _a(R)
    B, Cn, H, W = L.shape
    out = np.empty((B, G, D, H, W), np.float32)
    lib().ora_gwc_volume(L.ctypes, R.ctypes, out.ctypes, B, Cn, H, W, D, G)
    return out


def concat_volume(L, R, D, mask_left=True):
    L, R = _a(L), _a(R)
    B, Cn, H, W = L.shape
    out = np.empty((B, 2 * Cn, D, H, W), np.float32)
    lib().ora_concat_volume(L.ctypes, R.ctypes, out.ctypes, B, Cn, H, W, D, int(mask_left))
    return out


def conv3d(x, w, stride=1, pad=(1, 1, 1), dil=(1, 1, 1)):
    x, w = _a(x), _a(w)
    B, Ci, Di, Hi, Wi = x.shape
    Co, _, kd, kh, kw = w.shape
    o = lambda n, k, p, d: (n + 2 * p - d * (k - 1) - 1) // stride + 1
    out = np.empty((B, Co, o(Di, kd, pad[0], dil[0]), o(Hi, kh, pad[1], dil[1]), o(Wi, kw, pad[2], dil[2])), np.float32)
    lib().ora_conv3d(x.ctypes, w.ctypes, out.ctypes, B, Ci, Di, Hi, Wi, Co, kd, kh, kw, stride, *pad, *dil)
    return out


def deconv3d(x, w, k, stride=2, pad=1, opad=1):
    x, w = _a(x), _a(w)
    B, Ci, Di, Hi, Wi = x.shape
    Co = w.shape[1]
    o = lambda n: (n - 1) * stride - 2 * pad + k + opad
    out = np.empty((B, Co, o(Di), o(Hi), o(Wi)), np.float32)
    lib().ora_deconv3d(x.ctypes, w.ctypes, out.ctypes, B, Ci, Di, Hi, Wi, Co, k, stride, pad, opad)
    return out


def bn_act(y, mean, var, gamma, beta, eps=1e-5, res=None, act=0, slope=0.01):
    y = _a(y).copy()
    B, Cn = y.shape[:2]
    S = int(np.prod(y.shape[2:]))
    lib().ora_bn_act.argtypes = [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_float]
    r = None if res is None else _a(res)
    lib().ora_bn_act(y.ctypes.data, _a(mean).ctypes.data, _a(var).ctypes.data, _a(gamma).ctypes.data, _a(beta).ctypes.data,
                     eps, None if r is None else r.ctypes.data, B, Cn, S, act, slope)
    return y


def softargmin(prob):
    prob = _a(prob)
    B, D, H, W = prob.shape
    out = np.empty((B, H, W), np.float32)
    lib().ora_softargmin(prob.ctypes, out.ctypes, B, D, H, W)
    return out


def upsample_softargmin(cost, D, H, W, align=False):
    cost = _a(cost)
    if cost.ndim == 5:
        cost = cost[:, 0]
    B, Dl, Hl, Wl = cost.shape
    out = np.empty((B, H, W), np.float32)
    lib().ora_upsample_softargmin(cost.ctypes, out.ctypes, B, Dl, Hl, Wl, D, H, W, int(align))
    return out
