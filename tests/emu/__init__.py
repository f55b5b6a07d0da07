"""Host build of the device's per-surface step functions (TEST INFRASTRUCTURE ONLY, see emu_device.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.environ.get("ORT_EMU_LIB") or os.path.join(HERE, "libemu_device.so")     # (tests/test_sanitizers.py: an ASan + UBSan build)
_SRC = [os.path.join(HERE, "emu_device.cpp"), os.path.join(HERE, "stub", "hip", "hip_runtime.h"),
        os.path.join(ROOT, "opticalraytracing.jl_amd", "csrc", "ort_device.hpp")]
_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("ORT_EMU_LIB") and (not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in _SRC)):
            subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(HERE, "stub"),
                            "-o", LIB, _SRC[0]], check=True)
        _lib = C.CDLL(LIB)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def trace(pres, y, x, u, v, fast: bool, isys: int = 0):
    """(xv, yv, odd) of the device step functions run on the host: per-surface history [S][n] in the chosen policy and,
    for MATH_FAST, the per-ray `odd` flag (the kernel retraces the wave of such a ray with the reference sequence)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    R, t, n = f(np.atleast_2d(pres.R)[isys]), f(np.atleast_2d(pres.t)[isys]), f(np.atleast_2d(pres.n)[isys])
    K = None if pres.K is None else f(np.atleast_2d(pres.K)[isys])
    coef = None if pres.coef is None else f(pres.coef[isys])
    nc = 0 if coef is None else coef.shape[1]
    y, x, u, v = (f(a) for a in np.broadcast_arrays(y, x, u, v))
    N, S = y.size, R.size - 1
    xv = np.empty((S, N)); yv = np.empty((S, N)); odd = np.zeros(N, dtype=np.int32)
    arms = C.c_int(-1)
    lib().emu_trace(1 if fast else 0, R.size, _p(R), _p(t), _p(n), _p(K), _p(coef), nc, C.c_long(N), _p(y), _p(x), _p(u), _p(v),
                    _p(xv), _p(yv), odd.ctypes.data_as(C.POINTER(C.c_int)), C.byref(arms))
    global last_arms
    last_arms = arms.value            # the kernel build the host would launch for this system (ARMS_*: 0 basic .. 3 everything)
    return xv, yv, odd.astype(bool)


last_arms = -1


def trace_fast_with_retrace(pres, y, x, u, v, isys: int = 0):
    """What a MATH_FAST kernel leaves in memory, at ray granularity: the fast forms, except for `odd` rays, which get the
    reference sequence (the kernel retraces their whole wave; a superset of these rays)."""
    fx, fy, odd = trace(pres, y, x, u, v, True, isys)
    if odd.any():
        ix, iy, _ = trace(pres, np.asarray(y)[odd], np.asarray(x)[odd], np.asarray(u)[odd], np.asarray(v)[odd], False, isys)
        fx[:, odd], fy[:, odd] = ix, iy
    return fx, fy, odd


def fast_atan2(y, x):
    """ort::fast_atan2 (the FAST policy's theta), element-wise, on the host."""
    y = np.ascontiguousarray(y, dtype=np.float64); x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(y)
    L = lib()
    L.emu_fast_atan2.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
    L.emu_fast_atan2.restype = None
    L.emu_fast_atan2(y.size, _p(y), _p(x), _p(out))
    return out
