"""One-off soak on the GPU box: N random prescriptions x M random skew rays, IEEE policy vs the CPU oracle,
bit for bit (status and coordinates; polynomial rows to 1e-11), different seed from the test suite; also
extreme inputs (huge / tiny radii, grazing rays).  python scripts/soak_parity.py [nsys] [seed]"""
import math, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import Prescription
from oracle.cpu import OracleEngine
from tests import common as cm
from tests.test_gpu_parity import _random_system

nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
hip = ort.HipEngine(0); fast = ort.HipEngine(0, fast_math=True); orc = OracleEngine(nthreads=8)
bad = worst_fast = 0; nfr = ntot = 0
for case in range(nsys):
    rows = int(rng.integers(2, 20))
    aspheric = (True, "even", False)[case % 3]
    R, t, n, K, coef = _random_system(rng, rows, aspheric)
    if case % 7 == 0:                       # extremes: very strong and very weak curvatures
        R[1:] = np.where(rng.random(rows - 1) < 0.3, R[1:] * 1e4, R[1:])
        R[1:] = np.where(rng.random(rows - 1) < 0.2, np.sign(R[1:]) * rng.uniform(6.5, 9.0, rows - 1), R[1:])
    pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
    m = 800
    y = rng.uniform(-6, 6, m); x = rng.uniform(-6, 6, m)
    u = np.tan(rng.uniform(-0.25, 0.25, m)); v = np.tan(rng.uniform(-0.25, 0.25, m))
    ox, oy, os_ = orc.skew(pres, y, x, u, v, slopes=True, want_status=True)
    gx, gy, gs = hip.skew(pres, y, x, u, v, slopes=True, want_status=True)
    ok = np.array_equal(gs, os_)
    if aspheric:
        fin = np.isfinite(ox) & np.isfinite(gx)
        ok = ok and np.array_equal(np.isnan(gx), np.isnan(ox)) and cm.rel_err(gx, ox, 1.0).max() <= 1e-11 and cm.rel_err(gy, oy, 1.0).max() <= 1e-11
    else:
        ok = ok and np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True)
    if not ok:
        bad += 1
        print("MISMATCH case", case, "rows", rows, "aspheric", aspheric, "status diffs", int((gs != os_).sum()), flush=True)
    fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
    sane = (os_ == rows) & (np.nanmax(np.abs(ox), axis=0) < 1e3) & (np.nanmax(np.abs(oy), axis=0) < 1e3)
    err = np.maximum(cm.rel_err(fx, ox, 1.0).max(axis=0), cm.rel_err(fy, oy, 1.0).max(axis=0))
    nfr += int(((fs != os_) | (sane & (err > 1e-9))).sum()); ntot += m
    if sane.any():
        med = float(np.median(err[sane]))
        if med > 1e-9:
            j = int(np.argmax(np.where(sane, err, 0)))
            print("FAST OFF case", case, "rows", rows, "aspheric", aspheric, "median", med, "nsane", int(sane.sum()),
                  "\n R", R.tolist(), "\n t", t.tolist(), "\n n", n.tolist(), "\n K", K.tolist(),
                  "\n ray", float(y[j]), float(x[j]), float(u[j]), float(v[j]),
                  "\n ox", ox[:, j].tolist(), "\n fx", fx[:, j].tolist(), flush=True)
        worst_fast = max(worst_fast, med)
    if case % 250 == 0:
        print(f"case {case}: mismatching systems so far {bad}, fast fringe {nfr}/{ntot}", flush=True)
print(f"DONE {nsys} systems x 800 rays, seed {seed}: IEEE mismatching systems {bad}; FAST fringe (status flip or > 1e-9) {nfr}/{ntot} = {nfr / ntot:.2e}; "
      f"worst median FAST deviation {worst_fast:.2e}")
sys.exit(1 if bad else 0)
