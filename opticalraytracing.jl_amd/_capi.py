"""ctypes binding of libort_hip.so (include/ort.h) — the only way the host layer computes.

There is no CPU fallback here by design: if the shared library is not built, or no gfx950
device is usable, every call raises.  The reference has no FFI (it is pure Julia); the
equivalent `ccall` stubs a maintainer would add are in INTEGRATION.md.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORT_HIP_LIB") or os.path.join(_HERE, "csrc", "libort_hip.so")

# flags / codes (include/ort.h)
ORT_DEVICE_PTRS = 1 << 0
ORT_INPUT_SLOPES = 1 << 1
ORT_RAYBASIS = 1 << 2
ORT_LAYOUT_INPUT = 1 << 3
ORT_CLIP = 1 << 4
ORT_FAST_MATH = 1 << 5
ORT_FT_LOOKBACK = 1 << 7
ORT_FT_FUSED = 1 << 10
ORT_NO_SMALL_PATH = 1 << 8
ORT_AIM_EDGE_AS_FOUND = 1 << 9
ORT_STATUS_STOPPED = 1 << 16
ORT_STATUS_VIGNETTED = 1 << 17
ORT_MAX_ROWS = 64
ORT_MAX_NCOEF = 12
ORT_EDOMAIN = -2
ORT_SURF_NAMES = ("spherical", "coma", "astigmatism", "sagittal", "distortion", "axial", "lateral", "petzval",
                  "medial", "tangential")     # ORT_SURF_* order of include/ort.h
ORT_INC_NAMES = ("ni", "nibar", "i", "ibar")


class OrtError(RuntimeError):
    """A non-zero return code from the C ABI (message from ort_last_error())."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[ort {code}] {msg}")
        self.code = code


class ort_bundle(C.Structure):
    _fields_ = [
        ("system", C.c_int32), ("stop", C.c_int32),
        ("U", C.c_double), ("V", C.c_double),
        ("a_stop", C.c_double), ("hprime", C.c_double),
        ("ybar", C.c_double), ("z0", C.c_double),
        ("yaxis_off", C.c_int64), ("xaxis_off", C.c_int64),
    ]


class ort_aim_in(C.Structure):
    _fields_ = [
        ("system", C.c_int32), ("stop", C.c_int32), ("layout_fwd", C.c_int32), ("layout_rev", C.c_int32),
        ("H", C.c_double), ("y_marg", C.c_double), ("a_stop", C.c_double),
        ("chief_y_end", C.c_double), ("chief_u_end", C.c_double), ("f", C.c_double), ("atol", C.c_double),
    ]


class ort_aim_out(C.Structure):
    _fields_ = [
        ("U", C.c_double), ("y1", C.c_double), ("y2", C.c_double), ("y_EP", C.c_double),
        ("hprime", C.c_double), ("EP_t", C.c_double), ("Ubar", C.c_double), ("XP_t", C.c_double),
        ("iters", C.c_int32), ("ok", C.c_int32),
    ]


class ort_fan_in(C.Structure):
    _fields_ = [("system", C.c_int32), ("layout_mode", C.c_int32), ("y_marg", C.c_double), ("XP_t", C.c_double),
                ("BFD", C.c_double)]


class ort_first_order(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "f", "EBFD", "EFFD", "N", "FOV", "EP_D", "EP_t", "XP_D", "XP_t", "H",
        "y_marg", "chief_y_end", "chief_u_end", "nu_end", "BFD", "PN",
        "W040", "W131", "W222", "W220", "W311", "W020", "W111", "W220P")] + [("stop", C.c_int32), ("k", C.c_int32)]


class ort_grid_out_f64(C.Structure):
    _fields_ = [
        ("xv", C.c_void_p), ("yv", C.c_void_p), ("ld", C.c_int64),
        ("xf", C.c_void_p), ("yf", C.c_void_p), ("xs", C.c_void_p), ("ys", C.c_void_p),
        ("status", C.c_void_p),
    ]


class ort_grid_out_f32(C.Structure):
    _fields_ = ort_grid_out_f64._fields_


_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_u = C.c_uint

# name -> (restype, argtypes); kept in one table so tests can check it against include/ort.h
SIGNATURES = {
    "ort_version": (_i, []),
    "ort_last_error": (C.c_char_p, []),
    "ort_ctx_create": (_i, [_i, _p, C.POINTER(_p)]),
    "ort_ctx_destroy": (_i, [_p]),
    "ort_ctx_set_stream": (_i, [_p, _p]),
    "ort_ctx_synchronize": (_i, [_p]),
    "ort_ctx_timer_start": (_i, [_p]),
    "ort_ctx_timer_stop": (_i, [_p, C.POINTER(C.c_float)]),
    "ort_ctx_device_info": (_i, [_p, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_l)]),
    "ort_device_malloc": (_i, [_p, C.c_size_t, C.POINTER(_p)]),
    "ort_device_free": (_i, [_p, _p]),
    "ort_device_upload": (_i, [_p, _p, _p, C.c_size_t]),
    "ort_device_download": (_i, [_p, _p, _p, C.c_size_t]),
    "ort_system_create": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, C.POINTER(_p)]),
    "ort_system_destroy": (_i, [_p]),
    "ort_system_set_apertures": (_i, [_p, _p]),
    "ort_system_rows": (_i, [_p]),
    "ort_system_count": (_i, [_p]),
    "ort_trace_skew_f64": (_i, [_p, _p, _i, _l, _p, _p, _p, _p, _p, _p, _l, _p, _u]),
    "ort_trace_skew_f32": (_i, [_p, _p, _i, _l, _p, _p, _p, _p, _p, _p, _l, _p, _u]),
    "ort_trace_grid_f64": (_i, [_p, _p, _i, C.POINTER(ort_bundle), _p, _l, _i, _i, C.POINTER(ort_grid_out_f64), _u]),
    "ort_trace_grid_f32": (_i, [_p, _p, _i, C.POINTER(ort_bundle), _p, _l, _i, _i, C.POINTER(ort_grid_out_f32), _u]),
    "ort_make_axes_f64": (_i, [_p, _i, _i, _i, _p, _p, _u]),
    "ort_full_trace_f64": (_i, [_p, _p, _i, C.POINTER(ort_bundle), _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _u]),
    "ort_full_trace_f32": (_i, [_p, _p, _i, C.POINTER(ort_bundle), _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _u]),
    "ort_wavegrad_f64": (_i, [_p, _i, _l, _p, _p, C.c_double, _p, _p, _p, _p, _u]),
    "ort_ctx_test_skew_tickets": (_i, [_p, _l]),
    "ort_ctx_test_fused_no_scan": (_i, [_p, _i]),
    "ort_aim_f64": (_i, [_p, _p, _p, _i, C.POINTER(ort_aim_in), C.POINTER(ort_aim_out), _u]),
    "ort_fan_f64": (_i, [_p, _p, _i, C.POINTER(ort_fan_in), _i, _i, _p, _p, _u]),
    "ort_first_order_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, C.c_double, C.POINTER(ort_first_order), _u]),
    "ort_aberrations_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, C.c_double, C.POINTER(ort_first_order), _p, _p, _u]),
    "ort_spot_batch_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, _p, _i, C.POINTER(ort_first_order), _p, _p, _u]),
    "ort_full_trace_batch_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, _p, _i, C.POINTER(ort_first_order),
                                      _p, _p, _p, _p, _p, _p, _u]),
    "ort_full_trace_layout_batch_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i,
                                             C.POINTER(ort_first_order), _p, _p, _p, _p, _p, _p, _u]),
    "ort_spot_batch_f32": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, _p, _i, C.POINTER(ort_first_order), _p, _p, _u]),
    "ort_trace_meridional_f64": (_i, [_p, _p, _i, _l, _p, _p, _p, _p, _p, _l, _u]),
    "ort_ctx_domain_error": (_i, [_p, C.POINTER(_l), C.POINTER(_i), C.POINTER(_l)]),
    "ort_trace_paraxial_f64": (_i, [_p, _i, _i, _p, _p, _p, _l, _p, _p, _p, _p, _l, _u]),
    "ort_abcd_f64": (_i, [_p, _i, _i, _p, _p, _p, _u]),
    "ort_abcd_transfer_f64": (_i, [_p, _p, _l, _p, _p, _p, _p, _u]),
    "ort_abcd_reverse_transfer_f64": (_i, [_p, _p, _l, _p, _p, _p, _p, _u]),
    "ort_comm_unique_id": (_i, [_p]),
    "ort_comm_create": (_i, [_p, _i, _i, _p, C.POINTER(_p)]),
    "ort_comm_destroy": (_i, [_p]),
    "ort_comm_size": (_i, [_p]),
    "ort_comm_rank": (_i, [_p]),
    "ort_comm_wait": (_i, [_p]),
    "ort_comm_wait_lag": (_i, [_p, _i]),
    "ort_comm_synchronize": (_i, [_p]),
    "ort_allgather_hits_packed_f64": (_i, [_p, _p, _l, _p]),
    "ort_allgather_hits_packed_f32": (_i, [_p, _p, _l, _p]),
    "ort_allgather_hits_f64": (_i, [_p, _p, _p, _l, _p, _p]),
    "ort_allgather_ragged_f64": (_i, [_p, _p, _l, _p, _l, _p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libort_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("ORT_HIP_LIB"):
        # a source-only checkout: compile the extension in-tree (hipcc cross-compiles gfx950 in seconds)
        try:
            from . import build as _build
            _build.build(verbose=False)
        except Exception as exc:  # fall through to the loud failure below
            sys.stderr.write(f"[ort] in-tree build of the HIP extension failed: {exc}\n")
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback.")
    # torch ships its own libamdhip64 with the same SONAME; when both live in one process the
    # HIP runtime must be loaded once.  Importing torch first makes the dynamic loader reuse
    # torch's copy for this library (torch is only plumbing: device memory and collectives).
    if os.environ.get("ORT_NO_TORCH", "0") != "1" and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional for the library itself
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("ORT_HIP_LIB"):      # an older build loaded for an A/B run: it may lack newer entry points
                continue
            raise ImportError(f"{LIB_PATH} lacks {name}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise OrtError(rc, load().ort_last_error().decode("utf-8", "replace"))


def ptr(a) -> Optional[int]:
    """Address of a numpy array / torch tensor / int / None as a void*."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(f"cannot take the address of {type(a)}")


def f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        out = out.reshape(shape)
    return out


class Context:
    """ort_ctx: one per (host thread, GPU)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.ort_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ort_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.ort_ctx_synchronize(self.h))

    def set_stream(self, stream: int):
        check(self.lib.ort_ctx_set_stream(self.h, C.c_void_p(stream)))

    def timer_start(self):
        check(self.lib.ort_ctx_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.lib.ort_ctx_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mhz, mem = C.c_int(), C.c_int(), C.c_int64()
        check(self.lib.ort_ctx_device_info(self.h, name, 256, C.byref(cus), C.byref(mhz), C.byref(mem)))
        return {"name": name.value.decode(), "cus": cus.value, "clock_mhz": mhz.value, "mem_bytes": mem.value}


class DeviceSystem:
    """ort_system: device-resident batch of prescriptions (R, t, n, K, coef)."""

    def __init__(self, ctx: Context, R, t, n, K=None, coef=None):
        R = np.atleast_2d(f64(R))
        t = np.atleast_2d(f64(t))
        n = np.atleast_2d(f64(n))
        nsys, rows = R.shape
        if t.shape != R.shape or n.shape != R.shape:
            raise ValueError("R, t, n must have the same shape [nsys][rows]")
        Kp = None
        if K is not None:
            K = np.atleast_2d(f64(K))
            if K.shape != R.shape:
                raise ValueError("K must match R")
            Kp = K
        ncoef = 0
        cp = None
        if coef is not None:
            coef = f64(coef)
            if coef.ndim == 2:
                coef = coef[None, :, :]
            if coef.shape[:2] != (nsys, rows):
                raise ValueError("coef must be [nsys][rows][ncoef]")
            ncoef = coef.shape[2]
            cp = np.ascontiguousarray(coef)
        self.ctx = ctx
        self.nsys, self.rows, self.ncoef = nsys, rows, ncoef
        self._keep = (R, t, n, Kp, cp)
        h = C.c_void_p()
        check(ctx.lib.ort_system_create(ctx.h, nsys, rows, ptr(R), ptr(t), ptr(n), ptr(Kp), ptr(cp), ncoef, C.byref(h)))
        self.h = h

    def set_apertures(self, a) -> None:
        """Clear semi-diameters [nsys][rows-1] for the real-ray trace (extension; None clears)."""
        if a is None:
            check(self.ctx.lib.ort_system_set_apertures(self.h, None))
            return
        a = np.ascontiguousarray(np.broadcast_to(f64(a), (self.nsys, self.rows - 1)))
        check(self.ctx.lib.ort_system_set_apertures(self.h, ptr(a)))

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.ort_system_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_bundles(specs) -> "C.Array[ort_bundle]":
    """specs: iterable of dicts with the ort_bundle fields."""
    specs = list(specs)
    arr = (ort_bundle * len(specs))()
    for i, s in enumerate(specs):
        b = arr[i]
        b.system = int(s.get("system", 0))
        b.stop = int(s.get("stop", 0))
        b.U = float(s.get("U", 0.0))
        b.V = float(s.get("V", 0.0))
        b.a_stop = float(s.get("a_stop", np.inf))
        b.hprime = float(s.get("hprime", 0.0))
        b.ybar = float(s.get("ybar", 0.0))
        b.z0 = float(s.get("z0", 1.0))
        b.yaxis_off = int(s["yaxis_off"])
        b.xaxis_off = int(s["xaxis_off"])
    return arr
