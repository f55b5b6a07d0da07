"""Vectorised drivers for many-system runs (BASELINE configs 4-5): the per-system host logic of
api.py (`solve` -> `full_trace`) restated over arrays so that 10^4 perturbed instances cost a
handful of launches and no per-instance Python:

    first-order solve + Seidel sums   ort_first_order_f64   one thread per instance
    real-ray aiming                   ort_aim_f64           four lanes per (instance, field)
    pupil grid trace + spot stats     ort_full_trace_f64    statistics-only mode (16 B per bundle out)

Reference lines: solve src/RayTracing.jl:302-335, aberrations src/SeidelAberrations.jl:6-53,
full_trace src/PupilSampling.jl:85-147.  Plain spherical prescriptions [ninst][rows][3].
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np

from . import _capi
from .api import DomainError, EPS, LAMBDA, SPOT_RAYS, _eng
from .engine import Prescription

_FO_FIELDS = [f[0] for f in _capi.ort_first_order._fields_]


def first_order_arrays(engine, mats: np.ndarray, a, hprime, dn=None, lam: float = LAMBDA) -> Dict[str, np.ndarray]:
    """ort_first_order_f64 over [ninst][rows][3] -> dict of [ninst] arrays."""
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    ninst, rows, _ = mats.shape
    R = np.ascontiguousarray(mats[:, :, 0]); t = np.ascontiguousarray(mats[:, :, 1]); n = np.ascontiguousarray(mats[:, :, 2])
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    hp = np.ascontiguousarray(np.broadcast_to(np.asarray(hprime, dtype=np.float64), (ninst,)))
    dnp = None if dn is None else np.ascontiguousarray(np.broadcast_to(np.asarray(dn, dtype=np.float64), (ninst, rows)))
    out = (_capi.ort_first_order * ninst)()
    _capi.check(engine.ctx.lib.ort_first_order_f64(engine.ctx.h, ninst, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n),
                                                   _capi.ptr(a), _capi.ptr(dnp), _capi.ptr(hp), float(lam), out,
                                                   engine.base_flags))
    dt = np.dtype([(k, np.float64) for k in _FO_FIELDS[:-2]] + [("stop", np.int32), ("k", np.int32)])
    arr = np.frombuffer(out, dtype=dt, count=ninst)
    return {k: np.array(arr[k]) for k in _FO_FIELDS}


def tolerance_run(mats: np.ndarray, a, hprime, fields: Sequence[float] = (0.0,), k_rays: int = SPOT_RAYS,
                  dn=None, engine=None) -> Dict[str, np.ndarray]:
    """Seidel + spot-size Monte-Carlo over perturbed instances: for every instance the first-order
    properties and third-order sums, and for every (instance, field) the RMS spot radius of
    `full_trace(system, H, k_rays)` (stop-filtered, mirrored, about the centroid) and its ray count.
    Returns arrays: first-order keys [ninst], `rms`, `count` [ninst][nfields]."""
    import os
    import time
    _t = [time.perf_counter()]
    _trace = os.environ.get("ORT_TRACE_PHASES") == "1"

    def _mark(name):
        if _trace:
            eng.ctx.synchronize()
            now = time.perf_counter()
            print(f"[tolerance_run] {name}: {(now - _t[0]) * 1e3:.2f} ms", flush=True)
            _t[0] = now

    eng = _eng(engine)
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    ninst, rows, _ = mats.shape
    fields = np.abs(np.asarray(fields, dtype=np.float64))
    if not np.all(fields <= 1.0):
        raise DomainError("Domain: |H| ≤ 1.0")
    nf = len(fields)
    a_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    fo = first_order_arrays(eng, mats, a_arr, hprime, dn)
    _mark("first_order")
    if not np.all(fo["k"] == rows - 1):
        raise ValueError("tolerance_run expects prescriptions whose last thickness is 0 (image space)")
    R, t, n = mats[:, :, 0], mats[:, :, 1].copy(), mats[:, :, 2]
    t[:, 0] = np.where(np.isfinite(t[:, 0]), t[:, 0], 0.0)                      # Lens() mutation (Q19)
    BFD = fo["BFD"]
    # forward / reversed prescriptions (RayTracing.jl:267-277; K, p are zero here)
    fwd = Prescription(R, t, n)
    rev_R = -np.concatenate([np.full((ninst, 1), math.inf), R[:, :0:-1]], axis=1)
    rev_t = t[:, ::-1].copy(); rev_t[:, 0] = BFD
    rev_n = n[:, ::-1].copy()
    rev = Prescription(rev_R, rev_t, rev_n, np.zeros_like(rev_R))
    # aiming: four lanes per (instance, field)
    na = ninst * nf
    ain = (_capi.ort_aim_in * na)()
    dt_in = np.dtype([("system", np.int32), ("stop", np.int32), ("layout_fwd", np.int32), ("layout_rev", np.int32),
                      ("H", np.float64), ("y_marg", np.float64), ("a_stop", np.float64), ("chief_y_end", np.float64),
                      ("chief_u_end", np.float64), ("f", np.float64), ("atol", np.float64)])
    spec = np.frombuffer(ain, dtype=dt_in, count=na)
    inst = np.repeat(np.arange(ninst, dtype=np.int32), nf)
    stop = fo["stop"][inst]
    spec["system"] = inst; spec["stop"] = stop; spec["layout_fwd"] = 0; spec["layout_rev"] = 1
    spec["H"] = np.tile(fields, ninst); spec["y_marg"] = fo["y_marg"][inst]
    a_stop = a_arr[inst, stop - 1]
    spec["a_stop"] = a_stop; spec["chief_y_end"] = fo["chief_y_end"][inst]; spec["chief_u_end"] = fo["chief_u_end"][inst]
    spec["f"] = fo["f"][inst]; spec["atol"] = EPS
    aout = (_capi.ort_aim_out * na)()
    _mark("aim specs (numpy)")
    sf, sr = eng.system(fwd), eng.system(rev)
    _mark("upload forward + reversed tables")
    _capi.check(eng.ctx.lib.ort_aim_f64(eng.ctx.h, sf.h, sr.h, na, ain, aout, eng.base_flags))
    _mark("aim kernel call")
    dt_out = np.dtype([(k, np.float64) for k in ("U", "y1", "y2", "y_EP", "hprime", "EP_t", "Ubar")] +
                      [("iters", np.int32), ("ok", np.int32)])
    aim = np.frombuffer(aout, dtype=dt_out, count=na)
    if not np.all(aim["ok"] == 1):
        raise RuntimeError(f"ray aiming did not converge for {int(np.sum(aim['ok'] != 1))} (instance, field) pairs")
    # extended prescriptions (PupilSampling.jl:111-114) and the pupil grids (:121-122)
    ext = Prescription(np.concatenate([R, np.full((ninst, 1), math.inf)], axis=1),
                       np.concatenate([t[:, :-1], BFD[:, None], np.zeros((ninst, 1))], axis=1),
                       np.concatenate([n, np.ones((ninst, 1))], axis=1))
    k2 = k_rays // 2
    barr = (_capi.ort_bundle * na)()
    dt_b = np.dtype([("system", np.int32), ("stop", np.int32), ("U", np.float64), ("V", np.float64), ("a_stop", np.float64),
                     ("hprime", np.float64), ("ybar", np.float64), ("z0", np.float64), ("yaxis_off", np.int64),
                     ("xaxis_off", np.int64)])
    bd = np.frombuffer(barr, dtype=dt_b, count=na)
    off = np.arange(na, dtype=np.int64) * (k_rays + k2)
    bd["system"] = inst; bd["stop"] = stop; bd["U"] = aim["U"]; bd["V"] = 0.0; bd["a_stop"] = np.abs(a_stop)
    bd["hprime"] = aim["hprime"]; bd["ybar"] = 0.0; bd["z0"] = 1.0; bd["yaxis_off"] = off; bd["xaxis_off"] = off + k_rays
    ends = np.ascontiguousarray(np.stack([aim["y1"], aim["y2"], np.zeros(na), aim["y_EP"]], axis=1))
    lib, h = eng.ctx.lib, eng.ctx.h
    se = eng.system(ext)
    _mark("bundles (numpy) + upload extended tables")
    try:
        import torch
        dev = torch.device("cuda", eng.ctx.device)
    except Exception:          # no torch: host buffers (axes cross PCIe twice)
        torch = None
    if torch is not None:
        # device-resident run: axes are generated on the GPU from 32 B of end points per bundle and
        # only (count, rms) come back — 16 B per bundle
        d_ends = torch.from_numpy(ends).to(dev)
        d_axes = torch.empty(na * (k_rays + k2), dtype=torch.float64, device=dev)
        d_count = torch.empty(na, dtype=torch.int64, device=dev)
        d_rms = torch.empty(na, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        fl = eng.base_flags | _capi.ORT_DEVICE_PTRS
        _capi.check(lib.ort_make_axes_f64(h, na, k_rays, k2, d_ends.data_ptr(), d_axes.data_ptr(), fl))
        _capi.check(lib.ort_full_trace_f64(h, se.h, na, barr, d_axes.data_ptr(), d_axes.numel(), k_rays, k2,
                                           None, None, None, None, d_count.data_ptr(), d_rms.data_ptr(), fl))
        eng.ctx.synchronize()
        count, rms = d_count.cpu().numpy(), d_rms.cpu().numpy()
    else:
        axes = np.empty(na * (k_rays + k2))
        _capi.check(lib.ort_make_axes_f64(h, na, k_rays, k2, _capi.ptr(ends), _capi.ptr(axes), eng.base_flags))
        count = np.zeros(na, dtype=np.int64); rms = np.zeros(na)
        _capi.check(lib.ort_full_trace_f64(h, se.h, na, barr, _capi.ptr(axes), axes.size, k_rays, k2,
                                           None, None, None, None, _capi.ptr(count), _capi.ptr(rms), eng.base_flags))
    _mark("axes + trace + statistics")
    out = dict(fo)
    out["rms"] = rms.reshape(ninst, nf)
    out["count"] = count.reshape(ninst, nf)
    out["U"] = np.array(aim["U"]).reshape(ninst, nf)
    out["aim_iters"] = np.array(aim["iters"]).reshape(ninst, nf)
    return out


def spot_batch(mats: np.ndarray, a, hprime, fields: Sequence[float] = (0.0,), k_rays: int = SPOT_RAYS,
               engine=None, dtype=np.float64) -> Dict[str, np.ndarray]:
    """`tolerance_run` as ONE C call (`ort_spot_batch_f64`): solve -> aiming -> pupil axes -> grid trace
    -> spot statistics chained on the context's stream with every intermediate (first-order results,
    forward / reversed / extended tables, aiming requests, bundles, axes) built and kept on the GPU.
    In: 3 x [ninst][rows] prescriptions + semi-diameters; out: first-order struct per instance and
    16 B (count, rms) per (instance, field).  dtype = np.float32 traces the pupil grid in binary32
    (BASELINE config 5); solve, aiming and the statistics stay binary64."""
    eng = _eng(engine)
    fn = eng.ctx.lib.ort_spot_batch_f64 if np.dtype(dtype) == np.float64 else eng.ctx.lib.ort_spot_batch_f32
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    ninst, rows, _ = mats.shape
    fields = np.ascontiguousarray(np.abs(np.asarray(fields, dtype=np.float64)))
    nf = len(fields)
    R = np.ascontiguousarray(mats[:, :, 0]); t = np.ascontiguousarray(mats[:, :, 1]); n = np.ascontiguousarray(mats[:, :, 2])
    a_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    hp = np.ascontiguousarray(np.broadcast_to(np.asarray(hprime, dtype=np.float64), (ninst,)))
    fo = (_capi.ort_first_order * ninst)()
    count = np.zeros(ninst * nf, dtype=np.int64); rms = np.zeros(ninst * nf)
    rc = fn(eng.ctx.h, ninst, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n), _capi.ptr(a_arr),
                                        _capi.ptr(hp), nf, _capi.ptr(fields), int(k_rays), fo, _capi.ptr(count),
                                        _capi.ptr(rms), eng.base_flags)
    if rc == _capi.ORT_EDOMAIN:
        raise DomainError(eng.ctx.lib.ort_last_error().decode('utf-8', 'replace'))
    _capi.check(rc)
    dt = np.dtype([(k, np.float64) for k in _FO_FIELDS[:-2]] + [("stop", np.int32), ("k", np.int32)])
    arr = np.frombuffer(fo, dtype=dt, count=ninst)
    out = {k: np.array(arr[k]) for k in _FO_FIELDS}
    out["rms"] = rms.reshape(ninst, nf)
    out["count"] = count.reshape(ninst, nf)
    return out


def full_trace_systems(mats: np.ndarray, a, hprime, fields: Sequence[float] = (0.0,), k_rays: int = SPOT_RAYS,
                       engine=None, coef=None):
    """`[full_trace(solve(M, a, h′), H, k_rays) for M in mats, H in fields]` as ONE C call: the spot pipeline of
    `spot_batch` with the error vectors returned.  mats : [ninst][rows][3] = [R t n] (`ort_full_trace_batch_f64`)
    or [ninst][rows][4] = [R t n K] and / or coef : [ninst][rows][ncoef] power-series coefficients of p — the
    reference's Layout(R, t, n, K, p) (`ort_full_trace_layout_batch_f64`).
    Returns (first-order dict of [ninst] arrays, list over (instance, field) of dicts with ex, ey, rho,
    theta, rms, count, H — the fields of RealRayError, src/Types.jl:184-192)."""
    eng = _eng(engine)
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    if mats.ndim == 2:
        mats = mats[None]
    ninst, rows, ncol = mats.shape
    fields = np.ascontiguousarray(np.abs(np.asarray(fields, dtype=np.float64)))
    nf = len(fields)
    R = np.ascontiguousarray(mats[:, :, 0]); t = np.ascontiguousarray(mats[:, :, 1]); n = np.ascontiguousarray(mats[:, :, 2])
    K = np.ascontiguousarray(mats[:, :, 3]) if ncol >= 4 else None
    if coef is not None:
        coef = np.ascontiguousarray(np.broadcast_to(np.asarray(coef, dtype=np.float64), (ninst, rows, np.shape(coef)[-1])))
    a_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    hp = np.ascontiguousarray(np.broadcast_to(np.asarray(hprime, dtype=np.float64), (ninst,)))
    na = ninst * nf
    cap = 2 * k_rays * (k_rays // 2)
    fo = (_capi.ort_first_order * ninst)()
    ex, ey, rho, th = (np.empty((na, cap)) for _ in range(4))
    count = np.zeros(na, dtype=np.int64); rms = np.zeros(na)
    lib, h = eng.ctx.lib, eng.ctx.h
    if K is None and coef is None:
        rc = lib.ort_full_trace_batch_f64(h, ninst, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n),
                                          _capi.ptr(a_arr), _capi.ptr(hp), nf, _capi.ptr(fields), int(k_rays), fo,
                                          _capi.ptr(ex), _capi.ptr(ey), _capi.ptr(rho), _capi.ptr(th),
                                          _capi.ptr(count), _capi.ptr(rms), eng.base_flags)
    else:
        rc = lib.ort_full_trace_layout_batch_f64(h, ninst, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n), _capi.ptr(K),
                                                 _capi.ptr(coef), 0 if coef is None else coef.shape[2],
                                                 _capi.ptr(a_arr), _capi.ptr(hp), nf, _capi.ptr(fields), int(k_rays), fo,
                                                 _capi.ptr(ex), _capi.ptr(ey), _capi.ptr(rho), _capi.ptr(th),
                                                 _capi.ptr(count), _capi.ptr(rms), eng.base_flags)
    if rc == _capi.ORT_EDOMAIN:
        raise DomainError(lib.ort_last_error().decode('utf-8', 'replace'))
    _capi.check(rc)
    dt = np.dtype([(k, np.float64) for k in _FO_FIELDS[:-2]] + [("stop", np.int32), ("k", np.int32)])
    arr = np.frombuffer(fo, dtype=dt, count=ninst)
    first = {k: np.array(arr[k]) for k in _FO_FIELDS}
    out = []
    for b in range(na):
        c = int(count[b])
        out.append({"ex": ex[b, :c].copy(), "ey": ey[b, :c].copy(), "rho": rho[b, :c].copy(), "theta": th[b, :c].copy(),
                    "rms": float(rms[b]), "count": c, "H": float(fields[b % nf])})
    return first, out


def image_hits(mats: np.ndarray, a, hprime, fields: Sequence[float], k: int, engine=None, shard=None,
               dtype=np.float64):
    """Image-plane hit points of every (instance, field) bundle over the FULL square pupil k x k
    (BASELINE config 4: zoom / wavelength sweeps): first-order solve and aiming on the device, pupil
    boxes from the aimed marginal and chief rays, one summary-mode trace writing only (x_f, y_f,
    status) — 20 B per ray.  Returns torch tensors on the engine's GPU: xf, yf [nb, k, k], status
    [nb, k, k] (bit 16 = rejected by the stop filter), in bundle order (instance-major, field-minor).

    shard = (rank, world): trace only this rank's contiguous slab of bundles (dist.shard_bounds) —
    concatenating the ranks' outputs in rank order (dist.allgather_hits / ort_allgather_hits_f64)
    reproduces the single-GPU result.  dtype = np.float32: Float32 trace and hits (BASELINE config 5:
    8 B per ray out); solve, aiming and the pupil axes are still computed in Float64."""
    import torch
    from . import dist as odist
    eng = _eng(engine)
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    ninst, rows, _ = mats.shape
    fields = np.abs(np.asarray(fields, dtype=np.float64))
    if not np.all(fields <= 1.0):
        raise DomainError("Domain: |H| ≤ 1.0")
    nf = len(fields)
    a_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    hp_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(hprime, dtype=np.float64), (ninst,)))
    na = ninst * nf
    lo, hi = (0, na) if shard is None else odist.shard_bounds(na, shard[1])[shard[0]]
    if hi <= lo:
        raise ValueError("image_hits: this shard holds no bundle")
    # a rank solves, aims and uploads only the instances its slab of bundles touches
    i0, i1 = lo // nf, (hi - 1) // nf + 1
    mats, a_arr, hp_arr = mats[i0:i1], a_arr[i0:i1], hp_arr[i0:i1]
    lo, hi, ninst = lo - i0 * nf, hi - i0 * nf, i1 - i0
    fo = first_order_arrays(eng, mats, a_arr, hp_arr)
    R, t, n = mats[:, :, 0], mats[:, :, 1].copy(), mats[:, :, 2]
    t[:, 0] = np.where(np.isfinite(t[:, 0]), t[:, 0], 0.0)
    BFD = fo["BFD"]
    fwd = Prescription(R, t, n)
    rev_R = -np.concatenate([np.full((ninst, 1), math.inf), R[:, :0:-1]], axis=1)
    rev_t = t[:, ::-1].copy(); rev_t[:, 0] = BFD
    rev = Prescription(rev_R, rev_t, n[:, ::-1].copy(), np.zeros_like(rev_R))
    nb = hi - lo
    inst = np.repeat(np.arange(ninst, dtype=np.int32), nf)[lo:hi]
    Hs = np.tile(fields, ninst)[lo:hi]
    stop = fo["stop"][inst]
    ain = (_capi.ort_aim_in * nb)()
    dt_in = np.dtype([("system", np.int32), ("stop", np.int32), ("layout_fwd", np.int32), ("layout_rev", np.int32),
                      ("H", np.float64), ("y_marg", np.float64), ("a_stop", np.float64), ("chief_y_end", np.float64),
                      ("chief_u_end", np.float64), ("f", np.float64), ("atol", np.float64)])
    spec = np.frombuffer(ain, dtype=dt_in, count=nb)
    a_stop = a_arr[inst, stop - 1]
    spec["system"] = inst; spec["stop"] = stop; spec["layout_fwd"] = 0; spec["layout_rev"] = 1; spec["H"] = Hs
    spec["y_marg"] = fo["y_marg"][inst]; spec["a_stop"] = a_stop; spec["chief_y_end"] = fo["chief_y_end"][inst]
    spec["chief_u_end"] = fo["chief_u_end"][inst]; spec["f"] = fo["f"][inst]; spec["atol"] = EPS
    aout = (_capi.ort_aim_out * nb)()
    sf, sr = eng.system(fwd), eng.system(rev)                    # objects held across the call (HipEngine.system)
    _capi.check(eng.ctx.lib.ort_aim_f64(eng.ctx.h, sf.h, sr.h, nb, ain, aout, eng.base_flags))
    dt_out = np.dtype([(kk, np.float64) for kk in ("U", "y1", "y2", "y_EP", "hprime", "EP_t", "Ubar")] +
                      [("iters", np.int32), ("ok", np.int32)])
    aim = np.frombuffer(aout, dtype=dt_out, count=nb)
    if not np.all(aim["ok"] == 1):
        raise RuntimeError("ray aiming did not converge")
    ext = Prescription(np.concatenate([R, np.full((ninst, 1), math.inf)], axis=1),
                       np.concatenate([t[:, :-1], BFD[:, None], np.zeros((ninst, 1))], axis=1),
                       np.concatenate([n, np.ones((ninst, 1))], axis=1))
    barr = (_capi.ort_bundle * nb)()
    dt_b = np.dtype([("system", np.int32), ("stop", np.int32), ("U", np.float64), ("V", np.float64), ("a_stop", np.float64),
                     ("hprime", np.float64), ("ybar", np.float64), ("z0", np.float64), ("yaxis_off", np.int64),
                     ("xaxis_off", np.int64)])
    bd = np.frombuffer(barr, dtype=dt_b, count=nb)
    off = np.arange(nb, dtype=np.int64) * (2 * k)
    bd["system"] = inst; bd["stop"] = stop; bd["U"] = aim["U"]; bd["V"] = 0.0; bd["a_stop"] = np.abs(a_stop)
    bd["hprime"] = aim["hprime"]; bd["ybar"] = 0.0; bd["z0"] = 1.0; bd["yaxis_off"] = off; bd["xaxis_off"] = off + k
    # full square pupil: y from the aimed upper to lower edge ray, x symmetric about the axis
    ends = np.ascontiguousarray(np.stack([aim["y1"], aim["y2"], -aim["y_EP"], aim["y_EP"]], axis=1))
    dev = torch.device("cuda", eng.ctx.device)
    d_ends = torch.from_numpy(ends).to(dev)
    d_axes = torch.empty(nb * 2 * k, dtype=torch.float64, device=dev)
    f32 = np.dtype(dtype) == np.float32
    tdt = torch.float32 if f32 else torch.float64
    xf = torch.empty((nb, k, k), dtype=tdt, device=dev); yf = torch.empty_like(xf)
    st = torch.empty((nb, k, k), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    fl = eng.base_flags | _capi.ORT_DEVICE_PTRS
    lib, h = eng.ctx.lib, eng.ctx.h
    _capi.check(lib.ort_make_axes_f64(h, nb, k, k, d_ends.data_ptr(), d_axes.data_ptr(), fl))
    import ctypes as C
    if f32:
        eng.ctx.synchronize()
        d_axes = d_axes.to(torch.float32)
        torch.cuda.synchronize(dev)
        out = _capi.ort_grid_out_f32()
        trace = lib.ort_trace_grid_f32
    else:
        out = _capi.ort_grid_out_f64()
        trace = lib.ort_trace_grid_f64
    out.xf, out.yf, out.status = xf.data_ptr(), yf.data_ptr(), st.data_ptr()
    se = eng.system(ext)
    _capi.check(trace(h, se.h, nb, barr, d_axes.data_ptr(), d_axes.numel(), k, k, C.byref(out), fl))
    eng.ctx.synchronize()
    return xf, yf, st
