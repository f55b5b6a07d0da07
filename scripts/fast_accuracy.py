"""Worst relative deviation of the FAST policy from the IEEE policy (== the oracle, bit for bit) on the
config-2 and config-3 systems at 256 x 256 per bundle.  python scripts/fast_accuracy.py"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import api, workloads

ie = ort.HipEngine(fast_math=False); fa = ort.HipEngine(fast_math=True)
for name, cfg in (("config2", workloads.config2), ("config3", workloads.config3)):
    pres, bundles, axes = cfg(api, 256, engine=ie)
    a = ie.grid(pres, bundles, axes, 256, 256, summary=False)
    b = fa.grid(pres, bundles, axes, 256, 256, summary=False)
    worst = 0.0
    for key in ("xv", "yv"):
        x, y = a[key], b[key]
        ok = ~np.isnan(x) & ~np.isnan(y)
        assert np.array_equal(np.isnan(x), np.isnan(y)) or (np.isnan(x) ^ np.isnan(y)).sum() < 10
        worst = max(worst, float(np.max(np.abs(x[ok] - y[ok]) / np.maximum(np.abs(x[ok]), 1.0))))
    print(f"{name}: max |fast - ieee| / max(|ieee|, 1 mm) = {worst:.3e}", flush=True)
