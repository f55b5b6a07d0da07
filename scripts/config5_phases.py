import os, sys, time
sys.path.insert(0, "/root/repo")
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads
eng = ort.HipEngine(0, fast_math=True)
mats = workloads.config5(None, ninst=10000)
batch.tolerance_run(mats[:64], workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=256, engine=eng)
os.environ["ORT_TRACE_PHASES"] = "1"
for rep in range(2):
    t0 = time.perf_counter()
    batch.tolerance_run(mats, workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=256, engine=eng)
    print("total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
