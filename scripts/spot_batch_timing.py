"""Config 5 (10^4 instances x 2 fields, k_rays=32): staged `tolerance_run` vs the single-call
device-resident `spot_batch`.  Run on the GPU box: python scripts/spot_batch_timing.py"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads
from tests import common as cm

eng = ort.HipEngine(fast_math=True)
mats = workloads.config5(None, ninst=10000)
for k in ([int(a) for a in sys.argv[1:]] or [32, 64, 256]):
    f32 = lambda *a, **kw: batch.spot_batch(*a, dtype=np.float32, **kw)
    for name, fn in (("tolerance_run", batch.tolerance_run), ("spot_batch", batch.spot_batch), ("spot_batch_f32", f32)):
        for rep in range(4):
            t0 = time.perf_counter()
            r = fn(mats, cm.DG_A, cm.DG_H, fields=(0.0, 1.0), k_rays=k, engine=eng)
            dt = time.perf_counter() - t0
            print(f"k_rays {k} {name} rep {rep}: {dt * 1e3:.2f} ms  mean rms {r['rms'].mean():.9e}  rays {int(r['count'].sum())}", flush=True)
