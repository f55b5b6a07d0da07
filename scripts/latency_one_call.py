"""Config 1 (the reference's own case: full_trace(system, H, 64) on the Cooke triplet) through the
one-call pipelines: ort_full_trace_batch_f64 (error vectors back) and ort_spot_batch_f64 (RMS only),
solve + aiming + axes + trace + statistics included.  python scripts/latency_one_call.py"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch
from tests import common as cm

eng = ort.HipEngine(fast_math=True)
mats = cm.cooke()[None]
for name, fn in (("full_trace_systems", lambda: batch.full_trace_systems(mats, cm.COOKE_A, cm.COOKE_H, (1.0,), 64, engine=eng)),
                 ("spot_batch", lambda: batch.spot_batch(mats, cm.COOKE_A, cm.COOKE_H, (1.0,), 64, engine=eng))):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{name}: median {np.median(ts):.3f} ms  min {ts.min():.3f} ms  p90 {np.percentile(ts, 90):.3f} ms", flush=True)
