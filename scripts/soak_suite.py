"""One-off soak: the randomized GPU parity tests of tests/test_gpu_parity.py re-run with shifted RNG seeds
(meridional / paraxial kernels, grid + full_trace pipelines, skew property test).  python scripts/soak_suite.py [n]"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from oracle.cpu import OracleEngine
from tests import test_gpu_parity as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hip, orc = ort.HipEngine(0), OracleEngine(nthreads=8)
orig = np.random.default_rng
fails = 0
for off in range(1, n + 1):
    np.random.default_rng = lambda seed=None, _o=off: orig(None if seed is None else seed + 1000 * _o)
    for name in ("test_random_systems_meridional_and_paraxial", "test_random_bundles_grid_and_full_trace",
                 "test_random_systems_property"):
        try:
            getattr(T, name)(hip, orc)
            print(f"seed shift {off}: {name} ok", flush=True)
        except AssertionError as e:
            fails += 1
            print(f"seed shift {off}: {name} FAILED {str(e)[:300]}", flush=True)
np.random.default_rng = orig
print("DONE, failures:", fails)
sys.exit(1 if fails else 0)
