"""GPU box: where tolerance_run spends its wall time (10^4 instances x 2 fields x 256x128)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads, _capi
import opticalraytracing_jl_amd.batch as B
eng = ort.HipEngine(0, fast_math=True)
mats = workloads.config5(None, ninst=10000)
B.tolerance_run(mats[:64], workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=256, engine=eng)
lib = eng.ctx.lib
marks = []
def wrap(name):
    f = getattr(lib, name)
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); lib.ort_ctx_synchronize(eng.ctx.h); marks.append((name, time.perf_counter() - t0)); return r
    return g
class L:  # proxy recording the time spent inside each C call
    def __getattr__(self, n):
        return wrap(n) if n.startswith("ort_") and n not in ("ort_last_error", "ort_ctx_synchronize") else getattr(lib, n)
eng.ctx.lib = L()
t0 = time.perf_counter()
res = B.tolerance_run(mats, workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=256, engine=eng)
tot = time.perf_counter() - t0
eng.ctx.lib = lib
acc = {}
for n, t in marks: acc[n] = acc.get(n, 0) + t
print(f"total {tot*1e3:.1f} ms")
for n, t in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {n:28s} {t*1e3:8.1f} ms")
print(f"  host python/numpy            {(tot - sum(acc.values()))*1e3:8.1f} ms")
