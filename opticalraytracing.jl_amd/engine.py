"""HipEngine — numpy-facing wrapper of the C ABI (include/ort.h).

The host layer (api.py) talks to an *engine*: an object with the methods below.  The only
engine the package ships is this one, and it runs every trace on the GPU through
libort_hip.so; it raises when the library or the device is missing.  (Tests build a second
engine around the CPU oracle to check host logic without a GPU; that class lives under
tests/, not here.)

Array conventions: per-surface histories are [rows_or_S, nrays] C-contiguous (surface-major,
the `ld` layout of the ABI); per-ray inputs are 1-D.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from ._capi import Context, DeviceSystem, check, f64, ptr


@dataclass
class Prescription:
    """Columns of the reference `surfaces` matrix for nsys systems (+ K, polynomial coefs)."""
    R: np.ndarray            # [nsys, rows]
    t: np.ndarray
    n: np.ndarray
    K: Optional[np.ndarray] = None      # [nsys, rows]
    coef: Optional[np.ndarray] = None   # [nsys, rows, ncoef]
    apertures: Optional[np.ndarray] = None   # [nsys, rows-1] clear semi-diameters (extension, see ort.h)

    def __post_init__(self):
        self.R = np.atleast_2d(f64(self.R))
        self.t = np.atleast_2d(f64(self.t))
        self.n = np.atleast_2d(f64(self.n))
        if self.K is not None:
            self.K = np.atleast_2d(f64(self.K))
        if self.coef is not None:
            c = f64(self.coef)
            self.coef = c[None] if c.ndim == 2 else c
        if self.apertures is not None:
            self.apertures = np.ascontiguousarray(np.broadcast_to(f64(self.apertures), (self.R.shape[0], self.R.shape[1] - 1)))

    @property
    def rows(self) -> int:
        return self.R.shape[1]

    @property
    def nsys(self) -> int:
        return self.R.shape[0]

    def key(self) -> bytes:
        parts = [self.R.tobytes(), self.t.tobytes(), self.n.tobytes(),
                 b"" if self.K is None else self.K.tobytes(),
                 b"" if self.coef is None else self.coef.tobytes(),
                 b"" if self.apertures is None else self.apertures.tobytes(),
                 str(self.R.shape).encode()]
        return b"|".join(parts)

    @classmethod
    def from_matrix(cls, surfaces, K=None, coef=None) -> "Prescription":
        M = np.asarray(surfaces, dtype=np.float64)
        return cls(M[:, 0].copy(), M[:, 1].copy(), M[:, 2].copy(), K, coef)


class HipEngine:
    name = "hip"

    def __init__(self, device: int = 0, stream: Optional[int] = None, fast_math: bool = False, fused_full_trace: bool = False):
        """fast_math: ORT_FAST_MATH on every call of this engine.  fused_full_trace: ORT_FT_FUSED on every call (full_trace's second
        pass inside the trace launch for launches of two or more bundles; bit-identical results, include/ort.h)."""
        self.ctx = Context(device, stream)
        self.base_flags = (_capi.ORT_FAST_MATH if fast_math else 0) | (_capi.ORT_FT_FUSED if fused_full_trace else 0)
        self._systems = OrderedDict()
        self.last_domain_error = None
        self.cache_size = 32

    # ---- systems ---------------------------------------------------------------------
    def system(self, pres: Prescription) -> DeviceSystem:
        """Device tables of `pres`, from an LRU cache.  An evicted entry is only dropped from the cache, never
        destroyed here: it is freed (DeviceSystem.__del__) when the last caller still holding the object lets go,
        so keep the returned OBJECT — not just its raw handle `.h` — for as long as the handle is in use."""
        k = pres.key()
        s = self._systems.get(k)
        if s is not None:
            self._systems.move_to_end(k)
            return s
        s = DeviceSystem(self.ctx, pres.R, pres.t, pres.n, pres.K, pres.coef)
        if pres.apertures is not None:
            s.set_apertures(pres.apertures)
        self._systems[k] = s
        while len(self._systems) > self.cache_size:
            self._systems.popitem(last=False)
        return s

    # ---- skew: raytrace(surfaces, y, x, U, V, Vector{RealRay})  PupilSampling.jl:34-65 --
    def skew(self, pres: Prescription, y, x, U, V, isys: int = 0, slopes: bool = False,
             want_status: bool = False):
        y, x, U, V = (np.atleast_1d(f64(a)) for a in (y, x, U, V))
        y, x, U, V = np.broadcast_arrays(y, x, U, V)
        y, x, U, V = (np.ascontiguousarray(a) for a in (y, x, U, V))
        N = y.size
        S = pres.rows - 1
        xv = np.empty((S, N)); yv = np.empty((S, N))
        st = np.empty(N, dtype=np.int32) if want_status else None
        flags = self.base_flags | (_capi.ORT_INPUT_SLOPES if slopes else 0)
        sysd = self.system(pres)
        check(self.ctx.lib.ort_trace_skew_f64(self.ctx.h, sysd.h, isys, N, ptr(y), ptr(x), ptr(U), ptr(V),
                                              ptr(xv), ptr(yv), N, ptr(st), flags))
        return (xv, yv, st) if want_status else (xv, yv)

    def skew_f32(self, pres: Prescription, y, x, u, v, isys: int = 0, nrays: Optional[int] = None):
        """The Float32 build of `skew` (`ort_trace_skew_f32`, BASELINE config 5's arithmetic; the reference itself is
        Float64-only, Q21) over an explicit ray list with SLOPES u = tan U, v = tan V given.  Returns xv, yv [S][N] float32
        and status.  nrays < N traces a prefix of the list into the same [S][N] buffers (ld = N)."""
        y, x, u, v = np.broadcast_arrays(*(np.atleast_1d(np.asarray(a, dtype=np.float32)) for a in (y, x, u, v)))
        y, x, u, v = (np.ascontiguousarray(a) for a in (y, x, u, v))
        N = y.size
        S = pres.rows - 1
        xv = np.full((S, N), np.nan, dtype=np.float32); yv = np.full((S, N), np.nan, dtype=np.float32)
        st = np.zeros(N, dtype=np.int32)
        sysd = self.system(pres)
        check(self.ctx.lib.ort_trace_skew_f32(self.ctx.h, sysd.h, isys, N if nrays is None else int(nrays), ptr(y), ptr(x), ptr(u), ptr(v),
                                              ptr(xv), ptr(yv), N, ptr(st), self.base_flags | _capi.ORT_INPUT_SLOPES))
        return xv, yv, st

    # ---- grid bundles: PupilSampling.jl:121-138 -----------------------------------------
    def grid(self, pres: Prescription, bundles: Sequence[dict], axes, ny: int, nx: int,
             history: bool = True, summary: bool = True, raybasis: bool = False):
        axes = f64(axes).ravel()
        nb = len(bundles)
        N = nb * ny * nx
        S = pres.rows - 1
        out = _capi.ort_grid_out_f64()
        res = {}
        if history:
            res["xv"] = np.empty((S, N)); res["yv"] = np.empty((S, N))
            out.xv, out.yv, out.ld = ptr(res["xv"]), ptr(res["yv"]), N
        if summary:
            have_stop = all(int(b.get("stop", 0)) > 0 for b in bundles)
            res["xf"] = np.empty(N); res["yf"] = np.empty(N)
            res["status"] = np.empty(N, dtype=np.int32)
            out.xf, out.yf, out.status = ptr(res["xf"]), ptr(res["yf"]), ptr(res["status"])
            if have_stop:
                res["xs"] = np.empty(N); res["ys"] = np.empty(N)
                out.xs, out.ys = ptr(res["xs"]), ptr(res["ys"])
        barr = _capi.make_bundles(bundles)
        flags = self.base_flags | (_capi.ORT_RAYBASIS if raybasis else 0)
        sysd = self.system(pres)
        check(self.ctx.lib.ort_trace_grid_f64(self.ctx.h, sysd.h, nb, barr, ptr(axes), axes.size, ny, nx,
                                              C.byref(out), flags))
        return res

    # ---- full_trace grid stage: PupilSampling.jl:121-146,169-173 -------------------------
    def full_trace_grid(self, pres: Prescription, bundles: Sequence[dict], axes, ny: int, nx: int,
                        raybasis: bool = False, stats_only: bool = False, dtype=np.float64,
                        lookback: bool = False, fused: bool = False) -> List[dict]:
        """dtype = np.float32 traces the grid in binary32 (`ort_full_trace_f32`; statistics stay binary64).
        lookback = True takes the ORT_FT_LOOKBACK route (the trace kernel writes the first half at its final place);
        fused = True the ORT_FT_FUSED route (the second pass inside the trace launch)."""
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("full_trace_grid: dtype must be float64 or float32")
        fn = self.ctx.lib.ort_full_trace_f64 if dtype == np.float64 else self.ctx.lib.ort_full_trace_f32
        axes = np.ascontiguousarray(f64(axes).ravel(), dtype=dtype)
        nb = len(bundles)
        cap = 2 * ny * nx
        count = np.zeros(nb, dtype=np.int64); rms = np.zeros(nb)
        barr = _capi.make_bundles(bundles)
        flags = self.base_flags | (_capi.ORT_RAYBASIS if raybasis else 0) | (_capi.ORT_FT_LOOKBACK if lookback else 0) | (_capi.ORT_FT_FUSED if fused else 0)
        sysd = self.system(pres)
        if stats_only:      # one pass, nothing ray-sized leaves (or is even written on) the device
            check(fn(self.ctx.h, sysd.h, nb, barr, ptr(axes), axes.size, ny, nx,
                     None, None, None, None, ptr(count), ptr(rms), flags))
            return [{"rms": float(rms[b]), "count": int(count[b])} for b in range(nb)]
        ex, ey, rho, th = (np.empty((nb, cap), dtype=dtype) for _ in range(4))
        check(fn(self.ctx.h, sysd.h, nb, barr, ptr(axes), axes.size, ny, nx,
                 ptr(ex), ptr(ey), ptr(rho), ptr(th), ptr(count), ptr(rms), flags))
        out = []
        for b in range(nb):
            c = int(count[b])
            out.append({"ex": ex[b, :c].copy(), "ey": ey[b, :c].copy(), "rho": rho[b, :c].copy(),
                        "theta": th[b, :c].copy(), "rms": float(rms[b]), "count": c})
        return out

    # ---- wavegrad(eps, lambda): PupilSampling.jl:165-167 ------------------------------------
    def wavegrad(self, ex, ey, count, nu, lam: float, gx=None, gy=None):
        """`ort_wavegrad_f64`: (ex nu / lambda, ey nu / lambda) of nb full_trace slabs [nb][cap].  numpy arrays go through
        host buffers; torch CUDA tensors (ex, ey, count, nu and the outputs) stay on the device — nothing is copied."""
        on_dev = hasattr(ex, "data_ptr")
        nb, cap = ex.shape
        if on_dev:
            import torch
            gx = torch.empty_like(ex) if gx is None else gx
            gy = torch.empty_like(ey) if gy is None else gy
            flags = self.base_flags | _capi.ORT_DEVICE_PTRS
        else:
            ex, ey = f64(ex), f64(ey)
            count = np.ascontiguousarray(count, dtype=np.int64); nu = np.ascontiguousarray(np.broadcast_to(f64(nu), (nb,)))
            gx = np.empty_like(ex); gy = np.empty_like(ey)
            flags = self.base_flags
        check(self.ctx.lib.ort_wavegrad_f64(self.ctx.h, int(nb), int(cap), ptr(count), ptr(nu), float(lam), ptr(ex), ptr(ey),
                                            ptr(gx), ptr(gy), flags))
        return gx, gy

    # ---- meridional: raytrace(surfaces, y, U, RealRay)  RayTracing.jl:145-169 -----------
    def meridional(self, pres: Prescription, y, U, layout_mode: bool = False, isys: int = 0):
        y, U = (np.atleast_1d(f64(a)) for a in (y, U))
        y, U = np.broadcast_arrays(y, U)
        y, U = np.ascontiguousarray(y), np.ascontiguousarray(U)
        N = y.size
        rows = pres.rows
        yo = np.empty((rows, N)); Uo = np.empty((rows, N)); ts = np.empty((rows, N))
        flags = self.base_flags | (_capi.ORT_LAYOUT_INPUT if layout_mode else 0)
        sysd = self.system(pres)
        rc = self.ctx.lib.ort_trace_meridional_f64(self.ctx.h, sysd.h, isys, N, ptr(y), ptr(U),
                                                   ptr(yo), ptr(Uo), ptr(ts), N, flags)
        # ORT_EDOMAIN: Base.asin would have thrown for some ray (RayTracing.jl:162); the arrays are complete (those
        # rays NaN): the message is kept for the host mirror, which raises the reference's DomainError
        self.last_domain_error = None
        if rc == _capi.ORT_EDOMAIN:
            self.last_domain_error = self.ctx.lib.ort_last_error().decode("utf-8", "replace")
        else:
            check(rc)
        return yo, Uo, ts

    # ---- batched aiming: RayTracing.jl:223-296 + PupilSampling.jl:67-83,94-103 ------------
    def aim(self, fwd: Prescription, rev: Prescription, specs: Sequence[dict], edge_as_found: bool = False) -> List[dict]:
        """ort_aim_f64.  edge_as_found: leave the edge-ray searches where the FD-Newton ends (diagnostic: the default
        ends them inside the stop's edge, a rule fitted to the reference's published Tessar figure, include/ort.h)."""
        n = len(specs)
        ain = (_capi.ort_aim_in * n)()
        for i, sp in enumerate(specs):
            a = ain[i]
            a.system, a.stop = int(sp["system"]), int(sp["stop"])
            a.layout_fwd, a.layout_rev = int(bool(sp.get("layout_fwd", 0))), int(bool(sp.get("layout_rev", 0)))
            a.H, a.y_marg, a.a_stop = float(sp["H"]), float(sp["y_marg"]), float(sp["a_stop"])
            a.chief_y_end, a.chief_u_end = float(sp["chief_y_end"]), float(sp["chief_u_end"])
            a.f, a.atol = float(sp["f"]), float(sp.get("atol", 1.4901161193847656e-08))
        aout = (_capi.ort_aim_out * n)()
        sf, sr = self.system(fwd), self.system(rev)      # both objects held across the call (see system())
        check(self.ctx.lib.ort_aim_f64(self.ctx.h, sf.h, sr.h, n, ain, aout,
                                       self.base_flags | (_capi.ORT_AIM_EDGE_AS_FOUND if edge_as_found else 0)))
        return [dict(U=o.U, y1=o.y1, y2=o.y2, y_EP=o.y_EP, hprime=o.hprime, EP_t=o.EP_t, Ubar=o.Ubar, XP_t=o.XP_t,
                     iters=o.iters, ok=bool(o.ok)) for o in aout]

    # ---- meridional fans: TSA (SeidelAberrations.jl:116-137) / caustic ray set (MakieExtension.jl:364-381) ----
    def fan(self, pres: Prescription, specs: Sequence[dict], k_rays: int, descending: bool = False):
        """ort_fan_f64: for every spec {system, layout_mode, y_marg, XP_t, BFD} the k_rays rays
        y = range(y_marg / k, y_marg, k) (reversed when `descending`), U = 0, in ONE launch.
        Returns (y_XP, eps), each [len(specs), k_rays]."""
        n = len(specs)
        fin = (_capi.ort_fan_in * n)()
        for i, sp in enumerate(specs):
            f = fin[i]
            f.system, f.layout_mode = int(sp.get("system", 0)), int(bool(sp.get("layout_mode", 0)))
            f.y_marg, f.XP_t, f.BFD = float(sp["y_marg"]), float(sp["XP_t"]), float(sp["BFD"])
        y_xp = np.empty((n, k_rays)); eps = np.empty((n, k_rays))
        sysd = self.system(pres)
        check(self.ctx.lib.ort_fan_f64(self.ctx.h, sysd.h, n, fin, int(k_rays), 1 if descending else 0,
                                       ptr(y_xp), ptr(eps), self.base_flags))
        return y_xp, eps

    # ---- batched first-order solve + Seidel sums: RayTracing.jl:302-323, SeidelAberrations.jl:6-53 --
    def first_order(self, R, t, n, a, hprime, dn=None, lam: float = 587.5618e-6) -> List[dict]:
        R, t, n = (np.atleast_2d(f64(v)) for v in (R, t, n))
        nsys, rows = R.shape
        a = np.ascontiguousarray(np.broadcast_to(f64(a), (nsys, rows - 1)))
        hp = np.ascontiguousarray(np.broadcast_to(f64(hprime), (nsys,)))
        dnp = None if dn is None else np.ascontiguousarray(np.broadcast_to(f64(dn), (nsys, rows)))
        out = (_capi.ort_first_order * nsys)()
        check(self.ctx.lib.ort_first_order_f64(self.ctx.h, nsys, rows, ptr(R), ptr(t), ptr(n), ptr(a), ptr(dnp), ptr(hp),
                                               float(lam), out, self.base_flags))
        names = [f[0] for f in _capi.ort_first_order._fields_]
        return [{k: getattr(o, k) for k in names} for o in out]

    # ---- per-surface Seidel contributions + incidences: SeidelAberrations.jl:25-34, RayTracing.jl:338-353 --
    def aberrations(self, R, t, n, a, hprime, dn=None, lam: float = 587.5618e-6) -> dict:
        """ort_aberrations_f64 over nsys prescriptions: dict with the first-order / Seidel-sum fields as
        [nsys] arrays, the ten per-surface vectors (_capi.ORT_SURF_NAMES) and the four incidence columns
        (_capi.ORT_INC_NAMES) as [nsys][rows-1] arrays."""
        R, t, n = (np.atleast_2d(f64(v)) for v in (R, t, n))
        nsys, rows = R.shape
        a = np.ascontiguousarray(np.broadcast_to(f64(a), (nsys, rows - 1)))
        hp = np.ascontiguousarray(np.broadcast_to(f64(hprime), (nsys,)))
        dnp = None if dn is None else np.ascontiguousarray(np.broadcast_to(f64(dn), (nsys, rows)))
        out = (_capi.ort_first_order * nsys)()
        surf = np.empty((len(_capi.ORT_SURF_NAMES), nsys, rows - 1)); inc = np.empty((4, nsys, rows - 1))
        check(self.ctx.lib.ort_aberrations_f64(self.ctx.h, nsys, rows, ptr(R), ptr(t), ptr(n), ptr(a), ptr(dnp), ptr(hp),
                                               float(lam), out, ptr(surf), ptr(inc), self.base_flags))
        names = [f[0] for f in _capi.ort_first_order._fields_]
        res = {k: np.array([getattr(o, k) for o in out]) for k in names}
        res.update({k: surf[j] for j, k in enumerate(_capi.ORT_SURF_NAMES)})
        res.update({k: inc[j] for j, k in enumerate(_capi.ORT_INC_NAMES)})
        return res

    # ---- paraxial: raytrace(lens, y, ω, a; clip)  RayTracing.jl:127-143 ------------------
    def paraxial(self, tau, phi, y, w, a=None, clip: bool = False):
        tau = np.atleast_2d(f64(tau)); phi = np.atleast_2d(f64(phi))
        nlens, k = tau.shape
        y, w = (np.atleast_1d(f64(v)) for v in (y, w))
        y, w = np.broadcast_arrays(y, w)
        y, w = np.ascontiguousarray(y), np.ascontiguousarray(w)
        N = y.size
        if N % nlens:
            raise ValueError("ray count must be a multiple of the lens count")
        ap = None
        if a is not None:
            ap = np.atleast_2d(f64(a))
            if ap.shape != tau.shape:
                raise ValueError("aperture vector must have one entry per lens row")
        rt_y = np.empty((k + 1, N)); rt_w = np.empty((k + 1, N))
        flags = self.base_flags | (_capi.ORT_CLIP if clip else 0)
        check(self.ctx.lib.ort_trace_paraxial_f64(self.ctx.h, nlens, k, ptr(tau), ptr(phi), ptr(ap),
                                                  N // nlens, ptr(y), ptr(w), ptr(rt_y), ptr(rt_w), N, flags))
        return rt_y, rt_w

    # ---- ABCD: TransferMatrix.jl:1-17 -----------------------------------------------------
    def abcd(self, tau, phi) -> np.ndarray:
        tau = np.atleast_2d(f64(tau)); phi = np.atleast_2d(f64(phi))
        nlens, k = tau.shape
        M = np.empty((nlens, 2, 2))
        check(self.ctx.lib.ort_abcd_f64(self.ctx.h, nlens, k, ptr(tau), ptr(phi), ptr(M), self.base_flags))
        return M

    def abcd_transfer(self, M, v, tau, tau_p, reverse: bool = False) -> np.ndarray:
        M = f64(M, (2, 2))
        v = np.atleast_2d(f64(v))
        nv = v.shape[0]
        tau = np.ascontiguousarray(np.broadcast_to(f64(tau), (nv,)))
        tau_p = np.ascontiguousarray(np.broadcast_to(f64(tau_p), (nv,)))
        out = np.empty((nv, 2))
        if reverse:
            check(self.ctx.lib.ort_abcd_reverse_transfer_f64(self.ctx.h, ptr(M), nv, ptr(v), ptr(tau_p), ptr(tau),
                                                             ptr(out), self.base_flags))
        else:
            check(self.ctx.lib.ort_abcd_transfer_f64(self.ctx.h, ptr(M), nv, ptr(v), ptr(tau), ptr(tau_p),
                                                     ptr(out), self.base_flags))
        return out


_default: Optional[HipEngine] = None


def default_engine() -> HipEngine:
    """The process-wide GPU engine (device 0).  Raises when no gfx950 device is usable."""
    global _default
    if _default is None:
        _default = HipEngine(0)
    return _default


def set_default_engine(engine) -> None:
    global _default
    _default = engine
