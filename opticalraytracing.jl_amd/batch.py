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
_FO_DTYPE = np.dtype([(k, np.float64) for k in _FO_FIELDS[:-2]] + [("stop", np.int32), ("k", np.int32)])


def split_columns(mats):
    """[ninst][rows][3] = [R t n] -> the three contiguous [ninst][rows] arrays the C ABI takes (include/ort.h: R, t, n :
    [nsys][rows]).  numpy takes ~0.5 ms to pull the columns of 10^4 x 12 x 3 apart; a caller that runs many batches (a tolerance
    loop, bench.py) does it once and passes the tuple (R, t, n) wherever this module takes `mats`."""
    if isinstance(mats, (tuple, list)) and len(mats) == 3 and all(np.ndim(m) == 2 for m in mats):
        return tuple(np.ascontiguousarray(m, dtype=np.float64) for m in mats)
    mats = np.asarray(mats, dtype=np.float64)
    if mats.ndim == 2:
        mats = mats[None]
    cols = np.ascontiguousarray(mats[:, :, :3].transpose(2, 0, 1))
    return cols[0], cols[1], cols[2]


def first_order_arrays(engine, mats: np.ndarray, a, hprime, dn=None, lam: float = LAMBDA) -> Dict[str, np.ndarray]:
    """ort_first_order_f64 over [ninst][rows][3] (or the pre-split (R, t, n)) -> dict of [ninst] arrays."""
    R, t, n = split_columns(mats)
    ninst, rows = R.shape
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    hp = np.ascontiguousarray(np.broadcast_to(np.asarray(hprime, dtype=np.float64), (ninst,)))
    dnp = None if dn is None else np.ascontiguousarray(np.broadcast_to(np.asarray(dn, dtype=np.float64), (ninst, rows)))
    out = (_capi.ort_first_order * ninst)()
    _capi.check(engine.ctx.lib.ort_first_order_f64(engine.ctx.h, ninst, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n),
                                                   _capi.ptr(a), _capi.ptr(dnp), _capi.ptr(hp), float(lam), out,
                                                   engine.base_flags))
    arr = np.frombuffer(out, dtype=_FO_DTYPE, count=ninst)
    return {k: arr[k] for k in _FO_FIELDS}                        # views into the struct array the call filled


_AIM_OUT = np.dtype([(k, np.float64) for k in ("U", "y1", "y2", "y_EP", "hprime", "EP_t", "Ubar", "XP_t")] +
                    [("iters", np.int32), ("ok", np.int32)])
_AIM_IN = np.dtype([("system", np.int32), ("stop", np.int32), ("layout_fwd", np.int32), ("layout_rev", np.int32),
                    ("H", np.float64), ("y_marg", np.float64), ("a_stop", np.float64), ("chief_y_end", np.float64),
                    ("chief_u_end", np.float64), ("f", np.float64), ("atol", np.float64)])


def aim_instances(mats: np.ndarray, a, hprime, fields: Sequence[float] = (0.0,), engine=None, fo=None) -> Dict[str, np.ndarray]:
    """Real-ray aiming of every (instance, field) pair in one launch of the four-lane aiming kernel
    (`ort_aim_f64`: trace_marginal_ray / trace_chief_ray / trace_edge_rays, src/RayTracing.jl:223-296,
    src/PupilSampling.jl:67-103) for plain spherical prescriptions [ninst][rows][3].  Returns the fields of
    ort_aim_out as [ninst][nfields] arrays (U, y1, y2, y_EP, hprime, EP_t, Ubar, XP_t, iters, ok)."""
    eng = _eng(engine)
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    ninst, rows, _ = mats.shape
    fields = np.abs(np.asarray(fields, dtype=np.float64))
    if not np.all(fields <= 1.0):
        raise DomainError("Domain: |H| ≤ 1.0")
    nf = len(fields)
    a_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
    if fo is None:
        fo = first_order_arrays(eng, mats, a_arr, hprime)
    R, t, n = mats[:, :, 0], mats[:, :, 1].copy(), mats[:, :, 2]
    t[:, 0] = np.where(np.isfinite(t[:, 0]), t[:, 0], 0.0)                      # Lens() mutation (Q19)
    fwd = Prescription(R, t, n)
    rev_R = -np.concatenate([np.full((ninst, 1), math.inf), R[:, :0:-1]], axis=1)   # RayTracing.jl:267-277
    rev_t = t[:, ::-1].copy(); rev_t[:, 0] = fo["BFD"]
    rev = Prescription(rev_R, rev_t, n[:, ::-1].copy(), np.zeros_like(rev_R))
    na = ninst * nf
    ain = (_capi.ort_aim_in * na)()
    spec = np.frombuffer(ain, dtype=_AIM_IN, count=na)
    inst = np.repeat(np.arange(ninst, dtype=np.int32), nf)
    stop = fo["stop"][inst]
    spec["system"] = inst; spec["stop"] = stop; spec["layout_fwd"] = 0; spec["layout_rev"] = 1
    spec["H"] = np.tile(fields, ninst); spec["y_marg"] = fo["y_marg"][inst]
    spec["a_stop"] = a_arr[inst, stop - 1]; spec["chief_y_end"] = fo["chief_y_end"][inst]
    spec["chief_u_end"] = fo["chief_u_end"][inst]; spec["f"] = fo["f"][inst]; spec["atol"] = EPS
    aout = (_capi.ort_aim_out * na)()
    sf, sr = eng.system(fwd), eng.system(rev)                    # objects held across the call (HipEngine.system)
    _capi.check(eng.ctx.lib.ort_aim_f64(eng.ctx.h, sf.h, sr.h, na, ain, aout, eng.base_flags))
    arr = np.frombuffer(aout, dtype=_AIM_OUT, count=na)
    return {k: np.array(arr[k]).reshape(ninst, nf) for k in _AIM_OUT.names}


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
    # aiming: four lanes per (instance, field), forward / reversed prescriptions built by aim_instances
    na = ninst * nf
    inst = np.repeat(np.arange(ninst, dtype=np.int32), nf)
    stop = fo["stop"][inst]
    a_stop = a_arr[inst, stop - 1]
    aim = {key: v.reshape(-1) for key, v in aim_instances(mats, a_arr, hprime, fields, engine=eng, fo=fo).items()}
    _mark("aiming (tables + kernel)")
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
    R, t, n = split_columns(mats)
    ninst, rows = R.shape
    fields = np.ascontiguousarray(np.abs(np.asarray(fields, dtype=np.float64)))
    nf = len(fields)
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
    arr = np.frombuffer(fo, dtype=_FO_DTYPE, count=ninst)   # views into the one struct array the call filled (no per-field copies)
    out = {k: arr[k] for k in _FO_FIELDS}
    out["rms"] = rms.reshape(ninst, nf)
    out["count"] = count.reshape(ninst, nf)
    return out


def full_trace_systems(mats: np.ndarray, a, hprime, fields: Sequence[float] = (0.0,), k_rays: int = SPOT_RAYS,
                       engine=None, coef=None, flags: int = 0):
    """`[full_trace(solve(M, a, h′), H, k_rays) for M in mats, H in fields]` as ONE C call: the spot pipeline of
    `spot_batch` with the error vectors returned.  mats : [ninst][rows][3] = [R t n] (`ort_full_trace_batch_f64`)
    or [ninst][rows][4] = [R t n K] and / or coef : [ninst][rows][ncoef] power-series coefficients of p — the
    reference's Layout(R, t, n, K, p) (`ort_full_trace_layout_batch_f64`).
    Returns (first-order dict of [ninst] arrays, list over (instance, field) of dicts with ex, ey, rho,
    theta, rms, count, H — the fields of RealRayError, src/Types.jl:184-192).  flags: further ORT_* flags of the call
    (`_capi.ORT_FT_FUSED`, `_capi.ORT_FT_LOOKBACK`: the route of the full_trace stage; same results)."""
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
                                          _capi.ptr(count), _capi.ptr(rms), eng.base_flags | int(flags))
    else:
        rc = lib.ort_full_trace_layout_batch_f64(h, ninst, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n), _capi.ptr(K),
                                                 _capi.ptr(coef), 0 if coef is None else coef.shape[2],
                                                 _capi.ptr(a_arr), _capi.ptr(hp), nf, _capi.ptr(fields), int(k_rays), fo,
                                                 _capi.ptr(ex), _capi.ptr(ey), _capi.ptr(rho), _capi.ptr(th),
                                                 _capi.ptr(count), _capi.ptr(rms), eng.base_flags | int(flags))
    if rc == _capi.ORT_EDOMAIN:
        raise DomainError(lib.ort_last_error().decode('utf-8', 'replace'))
    _capi.check(rc)
    arr = np.frombuffer(fo, dtype=_FO_DTYPE, count=ninst)
    first = {k: arr[k] for k in _FO_FIELDS}
    out = []
    for b in range(na):
        c = int(count[b])
        out.append({"ex": ex[b, :c].copy(), "ey": ey[b, :c].copy(), "rho": rho[b, :c].copy(), "theta": th[b, :c].copy(),
                    "rms": float(rms[b]), "count": c, "H": float(fields[b % nf])})
    return first, out


_BUNDLE = np.dtype([("system", np.int32), ("stop", np.int32), ("U", np.float64), ("V", np.float64), ("a_stop", np.float64),
                    ("hprime", np.float64), ("ybar", np.float64), ("z0", np.float64), ("yaxis_off", np.int64),
                    ("xaxis_off", np.int64)])


class ImageHitsPlan:
    """Everything `image_hits` does before its trace, kept on the device for repeated launches (bench.py's
    multi-GPU step; a zoom / tolerance loop that re-traces): first-order solve and aiming of the instances this
    shard touches, pupil axes, bundle descriptors.

    shard = (rank, world) splits the work into contiguous rank-ordered slabs (dist.shard_bounds) of
      unit="bundle": (instance, field) bundles — equal slabs whenever the bundle count divides;
      unit="row":    pupil ROWS of the flattened (instance, field, row) list — slabs within one row of equal for
                     any world size; a slab may start and end inside a bundle (<= 3 launches per trace).
    Concatenating the ranks' hits in rank order reproduces the single-GPU order either way
    (src/PupilSampling.jl:123,134-137: y outer, x inner, bundles in call order)."""

    def __init__(self, mats: np.ndarray, a, hprime, fields: Sequence[float], k: int, engine=None, shard=None,
                 dtype=np.float64, unit: str = "bundle"):
        import torch
        from . import dist as odist
        eng = _eng(engine)
        mats = np.ascontiguousarray(mats, dtype=np.float64)
        ninst, rows, _ = mats.shape
        fields = np.abs(np.asarray(fields, dtype=np.float64))
        if not np.all(fields <= 1.0):
            raise DomainError("Domain: |H| ≤ 1.0")
        if unit not in ("bundle", "row"):
            raise ValueError("unit must be 'bundle' or 'row'")
        nf = len(fields)
        a_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ninst, rows - 1)))
        hp_arr = np.ascontiguousarray(np.broadcast_to(np.asarray(hprime, dtype=np.float64), (ninst,)))
        na = ninst * nf
        per = k if unit == "row" else 1                            # shard units per bundle
        ulo, uhi = (0, na * per) if shard is None else odist.shard_bounds(na * per, shard[1])[shard[0]]
        if uhi <= ulo:
            raise ValueError("image_hits: this shard holds no work")
        lo, hi = ulo // per, (uhi - 1) // per + 1                  # bundles touched
        # a rank solves, aims and uploads only the instances its slab touches
        i0, i1 = lo // nf, (hi - 1) // nf + 1
        mats, a_arr, hp_arr = mats[i0:i1], a_arr[i0:i1], hp_arr[i0:i1]
        lo, hi, ninst = lo - i0 * nf, hi - i0 * nf, i1 - i0
        fo = first_order_arrays(eng, mats, a_arr, hp_arr)
        aim_all = aim_instances(mats, a_arr, hp_arr, fields, engine=eng, fo=fo)
        nb = hi - lo
        inst = np.repeat(np.arange(ninst, dtype=np.int32), nf)[lo:hi]
        aim = {key: v.reshape(-1)[lo:hi] for key, v in aim_all.items()}
        if not np.all(aim["ok"] == 1):
            raise RuntimeError("ray aiming did not converge")
        stop = fo["stop"][inst]
        a_stop = a_arr[inst, stop - 1]
        R, t, n = mats[:, :, 0], mats[:, :, 1].copy(), mats[:, :, 2]
        t[:, 0] = np.where(np.isfinite(t[:, 0]), t[:, 0], 0.0)
        BFD = fo["BFD"]
        ext = Prescription(np.concatenate([R, np.full((ninst, 1), math.inf)], axis=1),
                           np.concatenate([t[:, :-1], BFD[:, None], np.zeros((ninst, 1))], axis=1),
                           np.concatenate([n, np.ones((ninst, 1))], axis=1))
        off = np.arange(nb, dtype=np.int64) * (2 * k)

        def bundles(sel, row_start):
            arr = (_capi.ort_bundle * len(sel))()
            bd = np.frombuffer(arr, dtype=_BUNDLE, count=len(sel))
            bd["system"] = inst[sel]; bd["stop"] = stop[sel]; bd["U"] = aim["U"][sel]; bd["V"] = 0.0
            bd["a_stop"] = np.abs(a_stop[sel]); bd["hprime"] = aim["hprime"][sel]; bd["ybar"] = 0.0; bd["z0"] = 1.0
            bd["yaxis_off"] = off[sel] + row_start; bd["xaxis_off"] = off[sel] + k
            return arr

        # segments: (bundle array, number of bundles, rows per bundle, first ray of the segment inside the shard)
        segs, pos = [], 0
        idx = np.arange(nb)
        if unit == "bundle":
            segs.append((bundles(idx, 0), nb, k, 0))
        else:
            base = (i0 * nf + lo) * k                              # first row unit of local bundle 0
            for b0, nbs, rs, nrows in odist.row_segments(ulo - base, uhi - base, k):
                segs.append((bundles(idx[b0:b0 + nbs], rs), nbs, nrows, pos))
                pos += nbs * nrows * k
        self.n_rays = (uhi - ulo) * (k if unit == "row" else k * k)
        self.k, self.nb, self.eng, self.S = k, nb, eng, rows
        self.units = (ulo, uhi)
        # full square pupil: y from the aimed upper to lower edge ray, x symmetric about the axis
        ends = np.ascontiguousarray(np.stack([aim["y1"], aim["y2"], -aim["y_EP"], aim["y_EP"]], axis=1))
        dev = torch.device("cuda", eng.ctx.device)
        self.dev = dev
        d_ends = torch.from_numpy(ends).to(dev)
        d_axes = torch.empty(nb * 2 * k, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        self.fl = eng.base_flags | _capi.ORT_DEVICE_PTRS
        lib, h = eng.ctx.lib, eng.ctx.h
        _capi.check(lib.ort_make_axes_f64(h, nb, k, k, d_ends.data_ptr(), d_axes.data_ptr(), self.fl))
        eng.ctx.synchronize()
        self.f32 = np.dtype(dtype) == np.float32
        self.tdt = torch.float32 if self.f32 else torch.float64
        self.d_axes = d_axes.to(torch.float32) if self.f32 else d_axes
        torch.cuda.synchronize(dev)
        self.sys = eng.system(ext)                                 # the object, held for the life of the plan
        self.ext, self.inst, self.aim_U = ext, inst, aim["U"]      # extended prescriptions, local instance and aimed field angle per bundle
        self.segs = segs

    def new_hits(self):
        """A packed [2, n_rays] slab (x row, y row): what ort_allgather_hits_packed_* sends as ONE message."""
        import torch
        return torch.empty((2, self.n_rays), dtype=self.tdt, device=self.dev)

    def trace(self, hits, status=None) -> None:
        """Launch the summary-mode trace(s) of this shard: x_f into hits[0], y_f into hits[1] (asynchronous on the
        engine's stream); status: optional int32 [n_rays]."""
        import ctypes as C
        lib, h = self.eng.ctx.lib, self.eng.ctx.h
        es = 4 if self.f32 else 8
        for barr, nbs, ny, pos in self.segs:
            out = _capi.ort_grid_out_f32() if self.f32 else _capi.ort_grid_out_f64()
            out.xf, out.yf = hits[0].data_ptr() + pos * es, hits[1].data_ptr() + pos * es
            out.status = None if status is None else status.data_ptr() + pos * 4
            fn = lib.ort_trace_grid_f32 if self.f32 else lib.ort_trace_grid_f64
            _capi.check(fn(h, self.sys.h, nbs, barr, self.d_axes.data_ptr(), self.d_axes.numel(), ny, self.k, C.byref(out), self.fl))


def image_hits(mats: np.ndarray, a, hprime, fields: Sequence[float], k: int, engine=None, shard=None,
               dtype=np.float64):
    """Image-plane hit points of every (instance, field) bundle over the FULL square pupil k x k
    (BASELINE config 4: zoom / wavelength sweeps): first-order solve and aiming on the device, pupil
    boxes from the aimed marginal and chief rays, one summary-mode trace writing only (x_f, y_f,
    status) — 20 B per ray.  Returns torch tensors on the engine's GPU: xf, yf [nb, k, k], status
    [nb, k, k] (bit 16 = rejected by the stop filter), in bundle order (instance-major, field-minor).

    shard = (rank, world): trace only this rank's contiguous slab of bundles (dist.shard_bounds) —
    concatenating the ranks' outputs in rank order (dist.allgather_hits / ort_allgather_hits_packed_f64)
    reproduces the single-GPU result.  dtype = np.float32: Float32 trace and hits (BASELINE config 5:
    8 B per ray out); solve, aiming and the pupil axes are still computed in Float64.
    (ImageHitsPlan is the same thing split into a reusable setup and its launches, with row-level sharding.)"""
    import torch
    plan = ImageHitsPlan(mats, a, hprime, fields, k, engine=engine, shard=shard, dtype=dtype)
    hits = plan.new_hits()
    st = torch.empty(plan.n_rays, dtype=torch.int32, device=plan.dev)
    plan.trace(hits, st)
    plan.eng.ctx.synchronize()
    return hits[0].view(plan.nb, k, k), hits[1].view(plan.nb, k, k), st.view(plan.nb, k, k)
