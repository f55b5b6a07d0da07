#!/usr/bin/env python3
"""GPU box: where does the FAST arithmetic policy deviate from the oracle?  Random prescriptions x random skew
rays (the generators of tests/test_gpu_parity.py: the standard +-6 mm / +-0.1 rad bundles and, every fifth case,
strongly curved rows under +-14 mm / +-0.2 rad bundles), every ray classified by the oracle itself:
  * margin  = its smallest normalised distance to a miss / TIR / equator boundary (oracle skew_margins),
  * sens    = the oracle's own response (max relative change of any coordinate) to a 1e-13 relative perturbation
              of the launch coordinates and slopes: the conditioning of the ray's path.
Prints, per decade of the margin, ray count, status flips, rays beyond 1e-10, worst deviation, and the worst
ratio deviation / (1e-12 + sens); then the unattributed rays, if any."""
import os, sys
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
Z = lambda: np.zeros(len(edges))
cnt, flips, over10, worst, wratio = Z(), Z(), Z(), Z(), Z()
far_n = far_bad = odd_n = 0
tot = 0
shown = 0


def devi(ax, ay, bx, by):
    s = np.maximum(1.0, np.maximum(np.nanmax(np.abs(bx), axis=0, initial=0.0), np.nanmax(np.abs(by), axis=0, initial=0.0)))
    d = np.maximum(np.nanmax(np.abs(ax - bx), axis=0, initial=0.0), np.nanmax(np.abs(ay - by), axis=0, initial=0.0)) / s
    pat = (np.isnan(ax) != np.isnan(bx)).any(axis=0) | (np.isnan(ay) != np.isnan(by)).any(axis=0)
    return np.where(pat, np.inf, d)


for case in range(ncase):
    rows = int(rng.integers(2, 15))
    aspheric = (True, "even", False)[case % 3]
    R, t, n, K, coef = _random_system(rng, rows, aspheric)
    wide = case % 5 == 4
    if wide:                                # strongly curved rows and wide bundles: far-cap hits, TIR and misses
        fin = np.isfinite(R); R[fin] = np.sign(R[fin]) * rng.uniform(6.5, 30.0, int(fin.sum()))
    pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
    m = 2000
    w = 14.0 if wide else 6.0
    a = 0.2 if wide else 0.1
    y = rng.uniform(-w, w, m); x = rng.uniform(-w, w, m)
    u = np.tan(rng.uniform(-a, a, m)); v = np.tan(rng.uniform(-a, a, m))
    ox, oy, os_ = orc.skew(pres, y, x, u, v, slopes=True, want_status=True)
    d = 1e-13
    px, py, ps = orc.skew(pres, y * (1 + d), x * (1 - d), u * (1 + d), v * (1 - d), slopes=True, want_status=True)
    sens = devi(px, py, ox, oy)
    fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
    mg = orc.skew_margins(pres, y, x, u, v)
    cond = np.min(np.abs(mg[:, :3]), axis=1)
    err = devi(fx, fy, ox, oy)
    flip = fs != os_
    far = mg[:, 3] > 0
    far_n += int(far.sum()); far_bad += int((far & (flip | (err > 1e-10 + 100 * sens))).sum())
    ratio = err / (1e-12 + sens)
    unattr = (flip & (cond > 1e-9)) | (~flip & (cond > 1e-6) & (err > 1e-10) & ~(ratio < 100))
    for j in np.nonzero(unattr)[0][:4]:
        if shown < 12:
            shown += 1
            print(f"  [unattributed] case {case} rows {rows} wide {wide} aspheric {aspheric} ray {j}: status fast {fs[j]} oracle {os_[j]} "
                  f"(perturbed oracle {ps[j]}) err {err[j]:.2e} sens {sens[j]:.2e} margins {mg[j]}")
            print(f"     R {np.array2string(R, precision=1)} n {np.array2string(n, precision=3)}")
            print(f"     oracle x {np.array2string(ox[:, j], precision=5)} y {np.array2string(oy[:, j], precision=5)}")
            print(f"     fast   x {np.array2string(fx[:, j], precision=5)} y {np.array2string(fy[:, j], precision=5)}")
    odd_n += int(unattr.sum())
    k = np.searchsorted(edges, cond, side="right") - 1
    for j in range(len(edges)):
        sel = k == j
        if not sel.any():
            continue
        cnt[j] += sel.sum(); flips[j] += (flip & sel).sum(); over10[j] += (err[sel] > 1e-10).sum()
        e = err[sel & ~flip]; e = e[np.isfinite(e)]
        if e.size:
            worst[j] = max(worst[j], float(e.max()))
        r = ratio[sel & ~flip]; r = r[np.isfinite(r)]
        if r.size:
            wratio[j] = max(wratio[j], float(r.max()))
    tot += m
print(f"seed {seed}: {ncase} systems, {tot} rays; far-cap rays {far_n}, of them flipped or beyond 1e-10 + 100 sens: {far_bad}; "
      f"unattributed rays: {odd_n}")
print("smallest margin >=   rays    flips  >1e-10   worst deviation   worst deviation / (1e-12 + sens)")
for j, e in enumerate(edges):
    if cnt[j]:
        print(f"  {e:8.0e}      {int(cnt[j]):9d} {int(flips[j]):6d} {int(over10[j]):7d}   {worst[j]:.2e}          {wratio[j]:.2f}")
