"""ctypes front end of the CPU oracle (oracle/ort_oracle.c) + OracleEngine.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this.  `OracleEngine` offers the same
methods as the product's HipEngine so that host logic (api.py) can be exercised on a machine
without a GPU and so that GPU results can be compared call for call.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORT_ORACLE_LIB: another build of the same sources (tests/test_sanitizers.py loads an ASan + UBSan one)
LIB = os.environ.get("ORT_ORACLE_LIB") or os.path.join(HERE, "libort_oracle.so")
_SRC = [os.path.join(HERE, f) for f in ("ort_oracle.c", "ort_oracle_skew.inc", "ort_oracle.h", "Makefile")]

_lib = None


def build(force: bool = False) -> str:
    if os.environ.get("ORT_ORACLE_LIB"):
        return LIB
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in _SRC)
    if stale:
        subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_d, _i, _l = C.c_double, C.c_int, C.c_int64


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB)
    L.orc_trace_skew.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _d, _d, _d, _d, _dp, _dp]
    L.orc_trace_skew_slopes.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _d, _d, _d, _d, _dp, _dp, C.POINTER(_i)]
    L.orc_status.argtypes = [_i, _dp, _dp]; L.orc_status.restype = _i
    L.orc_skew_margins.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _l, _dp, _dp, _dp, _dp, _dp]
    L.orc_trace_skew_batch.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _l, _dp, _dp, _dp, _dp, _dp, _dp, _l, _ip, _i]
    L.orc_trace_skew_grid.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _i, _dp, _i, _dp, _d, _d, _dp, _dp, _l, _ip, _i]
    L.orc_trace_skew_grid.restype = _l
    L.orc_trace_skew_grid_f32.argtypes = [_i, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _i, _fp, C.c_float, C.c_float,
                                          _fp, _fp, _l, _ip, _i]
    L.orc_trace_skew_grid_f32.restype = _l
    L.orc_trace_skew_batch_f32.argtypes = [_i, _fp, _fp, _fp, _fp, _fp, _i, _l, _fp, _fp, _fp, _fp, _fp, _fp, _l, _ip, _i]
    L.orc_trace_meridional.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _i, _d, _d, _dp, _dp, _dp]
    L.orc_trace_meridional.restype = _i
    L.orc_lens_from_surfaces.argtypes = [_i, _dp, _dp, _dp, _dp, _dp]; L.orc_lens_from_surfaces.restype = _i
    L.orc_trace_paraxial.argtypes = [_i, _dp, _dp, _d, _d, _dp, _i, _dp, _dp]
    L.orc_abcd.argtypes = [_i, _dp, _dp, _dp]
    L.orc_extend.argtypes = [_dp, _d, _d, _dp]
    L.orc_transfer.argtypes = [_dp, _dp, _d, _d, _dp]
    L.orc_reverse_transfer.argtypes = [_dp, _dp, _d, _d, _dp]
    L.orc_full_trace_grid.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _i, _i, _dp, _i, _dp, _d, _d, _i, _d, _d,
                                      _i, _d, _d, _dp, _dp, _dp, _dp, _dp, C.POINTER(_l)]
    L.orc_full_trace_grid.restype = _l
    L.orc_sigma.argtypes = [_l, _dp, _dp]; L.orc_sigma.restype = _d
    L.orc_linrange.argtypes = [_d, _d, _i, _i]; L.orc_linrange.restype = _d
    L.orc_solve_aberrations.argtypes = [_i, _dp, _dp, _dp, _dp, _dp, _d, _d, C.POINTER(orc_system_t), _dp, _dp,
                                        _dp, _dp, _dp, _dp]
    L.orc_solve_aberrations.restype = _i
    _lib = L
    return L


class orc_system_t(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("f", "EBFD", "EFFD", "N", "FOV", "EP_D", "EP_t", "XP_D", "XP_t", "H", "PN",
                                          "W040", "W131", "W222", "W220", "W311", "W020", "W111", "W220P", "W220M",
                                          "W220T")] + [("stop", C.c_int), ("k", C.c_int)]


SURF_NAMES = ("spherical", "coma", "astigmatism", "sagittal", "distortion", "axial", "lateral", "petzval", "medial",
              "tangential")
INC_NAMES = ("ni", "nibar", "i", "ibar")


def solve_aberrations(surfaces, a, hprime: float, lam: float = 587.5618e-6, dn=None) -> dict:
    """orc_solve_aberrations: solve(surfaces, a, h′) + aberrations(...) + incidences(...) of the reference
    for ONE prescription (rows x 3 matrix [R t n]).  Returns the scalars, the ten per-surface vectors
    (SURF_NAMES), the four incidence columns (INC_NAMES) and the paraxial marginal / chief y, nu."""
    M = np.asarray(surfaces, dtype=np.float64)
    rows = M.shape[0]
    R, t, n = (np.ascontiguousarray(M[:, j]) for j in range(3))
    a = np.ascontiguousarray(a, dtype=np.float64)
    dnp = None if dn is None else np.ascontiguousarray(dn, dtype=np.float64)
    S = rows - 1
    out = orc_system_t()
    surf = np.empty((len(SURF_NAMES), S)); inc = np.empty((4, S))
    my, mw, cy, cw = (np.empty(rows + 1) for _ in range(4))
    rc = lib().orc_solve_aberrations(rows, _p(R), _p(t), _p(n), _p(a), _p(dnp), float(hprime), float(lam),
                                     C.byref(out), _p(surf), _p(inc), _p(my), _p(mw), _p(cy), _p(cw))
    if rc != 0:
        raise ValueError("orc_solve_aberrations: Lens() keeps the last row (finite, non-zero last thickness)")
    res = {k: getattr(out, k) for k, _ in orc_system_t._fields_}
    res.update({k: surf[j] for j, k in enumerate(SURF_NAMES)})
    res.update({k: inc[j] for j, k in enumerate(INC_NAMES)})
    res.update(marginal_y=my, marginal_nu=mw, chief_y=cy, chief_nu=cw)
    return res


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def linrange(a: float, b: float, n: int) -> np.ndarray:
    L = lib()
    return np.array([L.orc_linrange(a, b, n, i) for i in range(n)])


class _Sys:
    """One system's columns pulled out of a product-side Prescription-like object."""

    def __init__(self, pres, isys: int = 0):
        self.R = _f(np.atleast_2d(pres.R)[isys])
        self.t = _f(np.atleast_2d(pres.t)[isys])
        self.n = _f(np.atleast_2d(pres.n)[isys])
        K = getattr(pres, "K", None)
        self.K = None if K is None else _f(np.atleast_2d(K)[isys])
        coef = getattr(pres, "coef", None)
        if coef is None:
            self.coef, self.ncoef = None, 0
        else:
            c = np.asarray(coef, dtype=np.float64)
            c = c[isys] if c.ndim == 3 else c
            self.coef, self.ncoef = np.ascontiguousarray(c), c.shape[1]
        self.rows = len(self.R)

    def args(self):
        return (self.rows, _p(self.R), _p(self.t), _p(self.n), _p(self.K), _p(self.coef), self.ncoef)


class OracleEngine:
    """The engine interface of opticalraytracing.jl_amd/engine.py, answered by the C oracle."""
    name = "oracle"

    def __init__(self, nthreads: int = 1):
        self.L = lib()
        self.nthreads = nthreads
        self.last_domain_error = None

    def skew(self, pres, y, x, U, V, isys: int = 0, slopes: bool = False, want_status: bool = False):
        s = _Sys(pres, isys)
        y, x, U, V = np.broadcast_arrays(*(np.atleast_1d(_f(a)) for a in (y, x, U, V)))
        y, x, U, V = (np.ascontiguousarray(a) for a in (y, x, U, V))
        N, S = y.size, s.rows - 1
        xv = np.empty((S, N)); yv = np.empty((S, N)); st = np.empty(N, dtype=np.int32)
        if slopes:
            bx = np.empty(S); by = np.empty(S)
            for r in range(N):
                self.L.orc_trace_skew_slopes(*s.args(), y[r], x[r], U[r], V[r], _p(bx), _p(by), None)
                xv[:, r], yv[:, r] = bx, by
                st[r] = self.L.orc_status(S, _p(bx), _p(by))
        else:
            self.L.orc_trace_skew_batch(*s.args(), N, _p(y), _p(x), _p(U), _p(V), _p(xv), _p(yv), N,
                                        st.ctypes.data_as(_ip), self.nthreads)
        return (xv, yv, st) if want_status else (xv, yv)

    def skew_f32(self, pres, y, x, u, v, isys: int = 0):
        """The Float32 build of the same loop over an explicit ray list, SLOPES in (orc_trace_skew_batch_f32): the
        prescription and the rays are rounded to binary32 once, as ort_system_create / the caller do.  Returns xv, yv
        [S][N] float32 and status."""
        s = _Sys(pres, isys)
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        fp = lambda a: None if a is None else a.ctypes.data_as(_fp)
        R, t, n, K, coef = f(s.R), f(s.t), f(s.n), f(s.K), f(s.coef)
        y, x, u, v = np.broadcast_arrays(*(np.atleast_1d(np.asarray(a, dtype=np.float32)) for a in (y, x, u, v)))
        y, x, u, v = (np.ascontiguousarray(a) for a in (y, x, u, v))
        N, S = y.size, s.rows - 1
        xv = np.empty((S, N), dtype=np.float32); yv = np.empty((S, N), dtype=np.float32); st = np.empty(N, dtype=np.int32)
        self.L.orc_trace_skew_batch_f32(s.rows, fp(R), fp(t), fp(n), fp(K), fp(coef), s.ncoef, N, fp(y), fp(x), fp(u), fp(v),
                                        fp(xv), fp(yv), N, st.ctypes.data_as(_ip), self.nthreads)
        return xv, yv, st

    def skew_margins(self, pres, y, x, u, v, isys: int = 0) -> np.ndarray:
        """[nrays, 4] conditioning probe of orc_skew_margins (slopes in): min normalised sag discriminant,
        min refraction discriminant, min tilt radicand, far-cap hit count."""
        s = _Sys(pres, isys)
        y, x, u, v = np.broadcast_arrays(*(np.atleast_1d(_f(a)) for a in (y, x, u, v)))
        y, x, u, v = (np.ascontiguousarray(a) for a in (y, x, u, v))
        m = np.empty((y.size, 4))
        self.L.orc_skew_margins(*s.args(), y.size, _p(y), _p(x), _p(u), _p(v), _p(m))
        return m

    def grid(self, pres, bundles: Sequence[dict], axes, ny: int, nx: int, history=True, summary=True,
             raybasis: bool = False):
        axes = _f(axes).ravel()
        nb = len(bundles)
        rpb = ny * nx
        N = nb * rpb
        S = np.atleast_2d(pres.R).shape[1] - 1
        xv = np.empty((S, N)); yv = np.empty((S, N)); st = np.empty(N, dtype=np.int32)
        for b, bd in enumerate(bundles):
            s = _Sys(pres, int(bd.get("system", 0)))
            ya = np.ascontiguousarray(axes[bd["yaxis_off"]:bd["yaxis_off"] + ny])
            xa = np.ascontiguousarray(axes[bd["xaxis_off"]:bd["xaxis_off"] + nx])
            bx = np.empty((S, rpb)); by = np.empty((S, rpb)); bs = np.empty(rpb, dtype=np.int32)
            if raybasis:
                yy = np.repeat(ya, nx); xx = np.tile(xa, ny)
                UU = (bd["ybar"] - yy) / bd["z0"]; VV = -xx / bd["z0"]
                self.L.orc_trace_skew_batch(*s.args(), rpb, _p(yy), _p(xx), _p(_f(UU)), _p(_f(VV)), _p(bx), _p(by),
                                            rpb, bs.ctypes.data_as(_ip), self.nthreads)
            else:
                self.L.orc_trace_skew_grid(*s.args(), ny, _p(ya), nx, _p(xa), float(bd.get("U", 0.0)),
                                           float(bd.get("V", 0.0)), _p(bx), _p(by), rpb,
                                           bs.ctypes.data_as(_ip), self.nthreads)
            xv[:, b * rpb:(b + 1) * rpb] = bx; yv[:, b * rpb:(b + 1) * rpb] = by
            stop = int(bd.get("stop", 0))
            if stop > 0:
                ri = np.hypot(bx[stop - 1], by[stop - 1])
                bs = np.where(ri > bd.get("a_stop", np.inf), bs | (1 << 16), bs).astype(np.int32)
            st[b * rpb:(b + 1) * rpb] = bs
        res = {}
        if history:
            res["xv"], res["yv"] = xv, yv
        if summary:
            res["xf"], res["yf"], res["status"] = xv[-1].copy(), yv[-1].copy(), st
            if all(int(b.get("stop", 0)) > 0 for b in bundles):
                xs = np.empty(N); ys = np.empty(N)
                for b, bd in enumerate(bundles):
                    sl = slice(b * rpb, (b + 1) * rpb)
                    xs[sl] = xv[bd["stop"] - 1, sl]; ys[sl] = yv[bd["stop"] - 1, sl]
                res["xs"], res["ys"] = xs, ys
        return res

    def full_trace_grid(self, pres, bundles: Sequence[dict], axes, ny: int, nx: int,
                        raybasis: bool = False) -> List[dict]:
        axes = _f(axes).ravel()
        out = []
        for bd in bundles:
            s = _Sys(pres, int(bd.get("system", 0)))
            ya = np.ascontiguousarray(axes[bd["yaxis_off"]:bd["yaxis_off"] + ny])
            xa = np.ascontiguousarray(axes[bd["xaxis_off"]:bd["xaxis_off"] + nx])
            cap = 2 * ny * nx
            ex = np.empty(cap); ey = np.empty(cap); rho = np.empty(cap); th = np.empty(cap)
            rms = C.c_double(); traced = C.c_int64()
            cnt = self.L.orc_full_trace_grid(*s.args(), ny, _p(ya), nx, _p(xa), float(bd.get("U", 0.0)),
                                             float(bd.get("V", 0.0)), 1 if raybasis else 0,
                                             float(bd.get("ybar", 0.0)), float(bd.get("z0", 1.0)),
                                             int(bd["stop"]), float(bd["a_stop"]), float(bd.get("hprime", 0.0)),
                                             _p(ex), _p(ey), _p(rho), _p(th), C.byref(rms), C.byref(traced))
            out.append({"ex": ex[:cnt].copy(), "ey": ey[:cnt].copy(), "rho": rho[:cnt].copy(),
                        "theta": th[:cnt].copy(), "rms": rms.value, "count": int(cnt), "traced": traced.value})
        return out

    def meridional(self, pres, y, U, layout_mode: bool = False, isys: int = 0):
        s = _Sys(pres, isys)
        y, U = np.broadcast_arrays(np.atleast_1d(_f(y)), np.atleast_1d(_f(U)))
        N, rows = y.size, s.rows
        yo = np.empty((rows, N)); Uo = np.empty((rows, N)); ts = np.empty((rows, N))
        by = np.empty(rows); bU = np.empty(rows); bt = np.empty(rows)
        self.last_domain_error = None
        for r in range(N):
            dom = self.L.orc_trace_meridional(*s.args(), 1 if layout_mode else 0, float(y[r]), float(U[r]),
                                              _p(by), _p(bU), _p(bt))
            if dom and self.last_domain_error is None:
                self.last_domain_error = f"DomainError: asin(x) is not defined for |x| > 1: ray {r} at surface row {dom}"
            yo[:, r], Uo[:, r], ts[:, r] = by, bU, bt
        return yo, Uo, ts

    def fan(self, pres, specs: Sequence[dict], k_rays: int, descending: bool = False):
        """The ray loop of TSA (src/SeidelAberrations.jl:121-133): y = range(y_m / k, y_m, k), U = 0,
        raytrace(surfaces, y, 0.0, RealRay) (orc_trace_meridional), then
        y_XP = ray.y[end] + tan(ray.u[end]) XP_t (:131) and eps = ray.y[end] + tan(ray.u[end]) (BFD - sag(ray)),
        sag(ray) = ray.z[end-1] - ray.z[end] = -ts[end] (:130,132; RayTracing.jl:91,105-115); the last ray is the real
        marginal ray, whose sag is taken against the paraxial vertex (:125-127)."""
        import math
        n = len(specs)
        y_xp = np.empty((n, k_rays)); eps = np.empty((n, k_rays))
        for q, sp in enumerate(specs):
            ym = float(sp["y_marg"])
            ys = linrange(ym, ym / k_rays, k_rays) if descending else linrange(ym / k_rays, ym, k_rays)
            yo, Uo, ts = self.meridional(pres, ys, np.zeros(k_rays), bool(sp.get("layout_mode", 0)), int(sp.get("system", 0)))
            t_last = float(np.atleast_2d(pres.t)[int(sp.get("system", 0))][-1])
            for i in range(k_rays):
                tu = math.tan(Uo[-1, i])
                y_xp[q, i] = yo[-1, i] + tu * sp["XP_t"]
                sag = -ts[-1, i]                                  # sag(ray) = ray.z[end-1] - ray.z[end]  (RayTracing.jl:91)
                if not descending and i == k_rays - 1:            # TSA's marginal ray: sag(real, paraxial) = real.z[end-1] -
                    sag = sag + t_last                            # paraxial.z[end-1] = s_last, no last thickness (:93-95, :125-127);
                                                                  # the caustic set re-traces it like the others (MakieExtension.jl:369-371)
                eps[q, i] = yo[-1, i] + tu * (sp["BFD"] - sag)
        return y_xp, eps

    def paraxial(self, tau, phi, y, w, a=None, clip: bool = False):
        tau = np.atleast_2d(_f(tau)); phi = np.atleast_2d(_f(phi))
        nlens, k = tau.shape
        y, w = np.broadcast_arrays(np.atleast_1d(_f(y)), np.atleast_1d(_f(w)))
        N = y.size
        rpl = N // nlens
        ap = None if a is None else np.atleast_2d(_f(a))
        rt_y = np.empty((k + 1, N)); rt_w = np.empty((k + 1, N))
        by = np.empty(k + 1); bw = np.empty(k + 1)
        for r in range(N):
            l = r // rpl
            self.L.orc_trace_paraxial(k, _p(np.ascontiguousarray(tau[l])), _p(np.ascontiguousarray(phi[l])),
                                      float(y[r]), float(w[r]),
                                      None if ap is None else _p(np.ascontiguousarray(ap[l])),
                                      1 if clip else 0, _p(by), _p(bw))
            rt_y[:, r], rt_w[:, r] = by, bw
        return rt_y, rt_w

    def abcd(self, tau, phi) -> np.ndarray:
        tau = np.atleast_2d(_f(tau)); phi = np.atleast_2d(_f(phi))
        nlens, k = tau.shape
        M = np.empty((nlens, 2, 2))
        for l in range(nlens):
            m = np.empty(4)
            self.L.orc_abcd(k, _p(np.ascontiguousarray(tau[l])), _p(np.ascontiguousarray(phi[l])), _p(m))
            M[l] = m.reshape(2, 2)
        return M

    def abcd_transfer(self, M, v, tau, tau_p, reverse: bool = False) -> np.ndarray:
        M = _f(M).reshape(4)
        v = np.atleast_2d(_f(v))
        nv = v.shape[0]
        tau = np.broadcast_to(_f(tau), (nv,)); tau_p = np.broadcast_to(_f(tau_p), (nv,))
        out = np.empty((nv, 2))
        o = np.empty(2)
        for j in range(nv):
            vj = np.ascontiguousarray(v[j])
            if reverse:
                self.L.orc_reverse_transfer(_p(M), _p(vj), float(tau_p[j]), float(tau[j]), _p(o))
            else:
                self.L.orc_transfer(_p(M), _p(vj), float(tau[j]), float(tau_p[j]), _p(o))
            out[j] = o
        return out
