#!/usr/bin/env python3
"""GPU box: where does the FAST arithmetic policy deviate from the oracle?  Random prescriptions x random skew
rays (the generator of tests/test_gpu_parity.py::test_random_systems_property, more cases), every ray classified
by the oracle's conditioning probe (oracle/ort_oracle_skew.inc, skew_margins): distance to the miss / TIR /
equator boundaries and far-cap hits.  Prints, per decade of the smallest margin, the ray count, the status flips
and the worst coordinate deviation; far-cap rays separately."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd.engine import Prescription
from oracle.cpu import OracleEngine
from tests import common as cm
from tests.test_gpu_parity import _random_system

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2024
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(seed)
orc = OracleEngine()
fast = ort.HipEngine(0, fast_math=True)
edges = [0.0] + [10.0 ** e for e in range(-16, 1)]
cnt = np.zeros(len(edges), dtype=np.int64); flips = cnt.copy(); worst = np.zeros(len(edges)); over10 = cnt.copy()
far_n = far_flip = 0; far_worst = 0.0
tot = 0
for case in range(ncase):
    rows = int(rng.integers(2, 15))
    aspheric = case % 3 == 0
    R, t, n, K, coef = _random_system(rng, rows, aspheric)
    if case % 5 == 4:                       # strongly curved rows and wide bundles: far-cap hits, TIR and misses
        fin = np.isfinite(R); R[fin] = np.sign(R[fin]) * rng.uniform(6.5, 30.0, int(fin.sum()))
    pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
    m = 2000
    w = 14.0 if case % 5 == 4 else 6.0
    y = rng.uniform(-w, w, m); x = rng.uniform(-w, w, m)
    u = np.tan(rng.uniform(-0.2, 0.2, m)); v = np.tan(rng.uniform(-0.2, 0.2, m))
    ox, oy, os_ = orc.skew(pres, y, x, u, v, slopes=True, want_status=True)
    fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
    mg = orc.skew_margins(pres, y, x, u, v)
    cond = np.min(np.abs(mg[:, :3]), axis=1)
    err = np.maximum(cm.rel_err(fx, ox, 1.0).max(axis=0), cm.rel_err(fy, oy, 1.0).max(axis=0))
    flip = fs != os_
    far = mg[:, 3] > 0
    far_n += int(far.sum()); far_flip += int((flip & far).sum())
    if far.any():
        far_worst = max(far_worst, float(np.where(np.isfinite(err[far]), err[far], 0).max()))
    odd = (flip | (err > 1e-10)) & (cond > 1e-6) & ~far
    for j in np.nonzero(odd)[0][:6]:
        print(f"  [unattributed] case {case} rows {rows} aspheric {aspheric} ray {j}: status fast {fs[j]} oracle {os_[j]} err {err[j]:.2e} "
              f"margins {mg[j]} R {np.array2string(R, precision=1)} K {np.array2string(K, precision=2) if aspheric else None}"
              f" n {np.array2string(n, precision=3)} poly rows {np.nonzero(coef.any(axis=1))[0] if aspheric else None}")
        print(f"     oracle x {np.array2string(ox[:, j], precision=4)} y {np.array2string(oy[:, j], precision=4)}")
        print(f"     fast   x {np.array2string(fx[:, j], precision=4)} y {np.array2string(fy[:, j], precision=4)}")
    k = np.searchsorted(edges, cond, side="right") - 1
    for j in range(len(edges)):
        sel = (k == j) & ~far
        cnt[j] += int(sel.sum()); flips[j] += int((flip & sel).sum())
        if sel.any():
            e = err[sel]; e = e[np.isfinite(e)]
            if e.size: worst[j] = max(worst[j], float(e.max()))
            over10[j] += int((err[sel] > 1e-10).sum())
    tot += m
print(f"seed {seed}: {ncase} systems, {tot} rays; far-cap rays {far_n}: status flips {far_flip}, worst deviation {far_worst:.2e}")
print("smallest margin >=   rays      flips   >1e-10   worst rel. deviation")
for j, e in enumerate(edges):
    print(f"  {e:8.0e}      {cnt[j]:9d} {flips[j]:7d} {over10[j]:8d}   {worst[j]:.2e}")
