"""GPU box: BASELINE config 5 at scale — 10^4 perturbed Double-Gauss instances, Seidel sums + spot RMS."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads
eng = ort.HipEngine(0, fast_math=True)
for ninst, k in ((10000, 64), (10000, 256)):
    mats = workloads.config5(None, ninst=ninst)
    batch.tolerance_run(mats[:64], workloads.DG_A, workloads.DG_H, fields=(0.0,), k_rays=k, engine=eng)
    t0 = time.perf_counter()
    res = batch.tolerance_run(mats, workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=k, engine=eng)
    dt = time.perf_counter() - t0
    rays = ninst * 2 * k * (k // 2)
    print(f"{ninst} instances x 2 fields x {k}x{k//2} rays = {rays:.3e} rays ({rays*12:.3e} intersections): {dt*1e3:.1f} ms wall "
          f"-> {rays*12/dt:.3e} intersections/s end to end;  RMS on axis mean {res['rms'][:,0].mean():.5f} std {res['rms'][:,0].std():.5f}, "
          f"W040 mean {res['W040'].mean():.4f} std {res['W040'].std():.4f}")
