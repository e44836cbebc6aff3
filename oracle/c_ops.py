"""ORACLE (test infrastructure, not product): ctypes loader for the plain-C
restatement in ops_c.c (built by oracle/Makefile)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ops.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


_CT = {np.dtype(np.float32): ("f32", ctypes.c_float), np.dtype(np.float64): ("f64", ctypes.c_double)}


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def upfirdn2d_nhwc(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x: numpy [major, in_h, in_w, minor]; kernel: numpy [kh, kw]."""
    x = np.ascontiguousarray(x)
    kernel = np.ascontiguousarray(kernel, dtype=x.dtype)
    sfx, _ = _CT[x.dtype]
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    L = lib()
    out_h = L.oracle_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh)
    out_w = L.oracle_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw)
    out = np.empty((major, out_h, out_w, minor), dtype=x.dtype)
    fn = getattr(L, "oracle_upfirdn2d_" + sfx)
    fn.restype = None
    fn(_ptr(out), _ptr(x), _ptr(kernel), major, in_h, in_w, minor, kh, kw,
       up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    return out


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    x = np.ascontiguousarray(x)
    sfx, cty = _CT[x.dtype]
    use_bias = int(bias is not None and bias.size > 0)
    use_ref = int(ref is not None and ref.size > 0)
    b = np.ascontiguousarray(bias, dtype=x.dtype) if use_bias else np.zeros(1, x.dtype)
    r = np.ascontiguousarray(ref, dtype=x.dtype) if use_ref else np.zeros(1, x.dtype)
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    out = np.empty_like(x)
    fn = getattr(lib(), "oracle_fused_bias_act_" + sfx)
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long] + [ctypes.c_int] * 6 + [cty, cty]
    fn(_ptr(out), _ptr(x), _ptr(b), _ptr(r), x.size, step_b, max(b.size, 1), use_bias, use_ref,
       act, grad, float(np.float32(alpha)), float(np.float32(scale)))  # C float at the pybind boundary
    return out
