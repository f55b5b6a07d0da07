"""GPU box: one case of scripts/soak_parity.py again (same RNG stream), the rays whose deviation (reference-sequence policy on the
device against the oracle) is not explained by their conditioning, surface by surface.   python scripts/debug_asph_case.py seed maxrows case"""
import math, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import Prescription
from oracle.cpu import OracleEngine
from tests.test_gpu_parity import _random_system, _deviation
from tests import emu

seed, maxrows, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
hip = ort.HipEngine(0); orc = OracleEngine(nthreads=8)
for case in range(want + 1):
    rows = int(rng.integers(2, maxrows))
    aspheric = (True, "even", False)[case % 3]
    R, t, n, K, coef = _random_system(rng, rows, aspheric)
    if case % 7 == 0:
        R[1:] = np.where(rng.random(rows - 1) < 0.3, R[1:] * 1e4, R[1:])
        R[1:] = np.where(rng.random(rows - 1) < 0.2, np.sign(R[1:]) * rng.uniform(6.5, 9.0, rows - 1), R[1:])
    m = 800
    y = rng.uniform(-6, 6, m); x = rng.uniform(-6, 6, m)
    u = np.tan(rng.uniform(-0.25, 0.25, m)); v = np.tan(rng.uniform(-0.25, 0.25, m))
pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
ox, oy, os_ = orc.skew(pres, y, x, u, v, slopes=True, want_status=True)
gx, gy, gs = hip.skew(pres, y, x, u, v, slopes=True, want_status=True)
ex, ey, _ = emu.trace(pres, y, x, u, v, False)
d = 1e-13
px, py = orc.skew(pres, y * (1 + d), x * (1 - d), u * (1 + d), v * (1 - d), slopes=True)
sens = _deviation(px, py, ox, oy); dev = _deviation(gx, gy, ox, oy)
bad = np.nonzero(dev > np.maximum(1e-10, 100 * sens))[0]
print("case", want, "rows", rows, "aspheric", aspheric, "bad rays", bad.tolist(), "emulation == device:", bool(np.array_equal(ex, gx, equal_nan=True) and np.array_equal(ey, gy, equal_nan=True)))
print("row kinds: R finite", np.isfinite(R).astype(int).tolist(), "\nK", np.round(K, 3).tolist(), "\npoly rows", (np.abs(coef).sum(axis=1) > 0).astype(int).tolist())
for j in bad[:2]:
    print(f"ray {j}: launch y {y[j]:.6f} x {x[j]:.6f} u {u[j]:.6f} v {v[j]:.6f}; status {os_[j]}; conditioning {sens[j]:.2e}, deviation {dev[j]:.2e}")
    for i in range(rows - 1):
        print(f"  surf {i + 1:2d} R {R[i + 1]:10.3f} K {K[i + 1]:6.3f} poly {int(np.abs(coef[i + 1]).sum() > 0)}  oracle x {ox[i, j]: .12e} y {oy[i, j]: .12e}   device-oracle dx {gx[i, j] - ox[i, j]: .2e} dy {gy[i, j] - oy[i, j]: .2e}   perturbed-oracle dx {px[i, j] - ox[i, j]: .2e}")
