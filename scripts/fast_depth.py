"""GPU box: how the FAST policy's deviation from the oracle grows with the DEPTH of a prescription — the Cooke relay chains of
tests/test_gpu_parity.py::_cooke_relay at 24 / 40 / 63 / 64 rows (spheres, planes, conics, even and odd aspheres), many rays:
worst and percentile deviations, rays past 1e-10 (north_star's bar) and the oracle's own conditioning of those rays (largest
relative response to a 1e-13 perturbation of the launch data).   python scripts/fast_depth.py [rays_per_case]"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import Prescription
from oracle.cpu import OracleEngine
from tests.test_gpu_parity import _cooke_relay, _deviation

m = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
fast = ort.HipEngine(0, fast_math=True); hip = ort.HipEngine(0); orc = OracleEngine(nthreads=8)
for rows, units, pad in ((24, 3, 1), (32, 4, 2), (40, 5, 3), (47, 6, 3), (54, 7, 3), (57, 7, 6), (63, 8, 5), (64, 8, 6)):
    for variant in ("sph", "even", "mixed"):
        M, coef = _cooke_relay(units, variant, pad)
        pres = Prescription(M[:, 0], M[:, 1], M[:, 2], None if variant == "sph" else M[:, 3], None if variant == "sph" else coef[None])
        rng = np.random.default_rng(7 * rows + len(variant))
        w = np.where(np.arange(m) % 2 == 0, 1.0, 3.2)
        y = rng.uniform(-5, 5, m) * w; x = rng.uniform(-5, 5, m) * w
        u = np.tan(rng.uniform(-0.01, 0.01, m) * w); v = np.tan(rng.uniform(-0.01, 0.01, m) * w)
        ox, oy, os_ = orc.skew(pres, y, x, u, v, slopes=True, want_status=True)
        d = 1e-13
        px, py = orc.skew(pres, y * (1 + d), x * (1 - d), u * (1 + d), v * (1 - d), slopes=True)
        sens = _deviation(px, py, ox, oy)
        fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
        gx, gy, gs = hip.skew(pres, y, x, u, v, slopes=True, want_status=True)
        err = _deviation(fx, fy, ox, oy)
        ok = np.isfinite(err)
        past = err > 1e-10
        fail = err > np.maximum(1e-10, 100.0 * sens)              # the bar of tests/test_gpu_parity.py::_fast_attribution
        ieee_equal = bool(np.array_equal(gs, os_) and (variant != "sph" or (np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True))))
        print(f"{rows} rows {variant:5s}: rays {m}, status flips {int((fs != os_).sum())}, reference-sequence policy {'bit-identical' if variant == 'sph' and ieee_equal else ('status identical' if ieee_equal else 'MISMATCH')}; "
              f"FAST deviation median {np.median(err[ok]):.1e}, 99.9 % {np.quantile(err[ok], 0.999):.1e}, worst {err[ok].max():.1e}; past 1e-10: {int(past.sum())}, past max(1e-10, 100 x conditioning): {int(fail.sum())} "
              f"(their conditioning: {', '.join(f'{s:.1e}' for s in np.sort(sens[past])[-3:]) if past.any() else '-'}; median conditioning of all rays {np.median(sens[ok]):.1e})", flush=True)
