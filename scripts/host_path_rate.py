"""PCIe-inclusive rate of the HOST-buffer entry point (what a `ccall` with Julia arrays sees), config 2:
pageable numpy outputs vs page-locked outputs (torch pin_memory, or hipHostRegister by the caller).
Run on the GPU box: python scripts/host_path_rate.py [pupil]"""
import ctypes as C
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import torch
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import _capi, api, workloads

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = ort.HipEngine(fast_math=True)
ort.set_default_engine(eng)
pres, bundles, axes = workloads.config2(api, k, engine=eng)
nb = len(bundles); N = nb * k * k; S = pres.rows - 1
sysd = eng.system(pres); barr = _capi.make_bundles(bundles)
for name, alloc in (("pageable", lambda: np.empty((S, N))),
                    ("pinned", lambda: torch.empty((S, N), dtype=torch.float64).pin_memory().numpy())):
    xv, yv = alloc(), alloc()
    out = _capi.ort_grid_out_f64(); out.xv, out.yv, out.ld = xv.ctypes.data, yv.ctypes.data, N
    for rep in range(3):
        t0 = time.perf_counter()
        _capi.check(eng.ctx.lib.ort_trace_grid_f64(eng.ctx.h, sysd.h, nb, barr, axes.ctypes.data, axes.size, k, k,
                                                   C.byref(out), eng.base_flags))
        dt = time.perf_counter() - t0
        print(f"{name} rep {rep}: {dt * 1e3:.1f} ms  {N * S / dt:.3e} intersections/s  {16.0 * N * S / dt / 1e9:.1f} GB/s to host",
              flush=True)
    del xv, yv
