"""GPU box: explicit ray lists (the batch form of raytrace(surfaces, y, x, U, V, Vector{RealRay})) through
device pointers: 9.4 M rays, S = 12, history out; angles (tan on the device) vs slopes; f64 and f32."""
import ctypes as C, math, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import _capi, workloads, Prescription
eng = ort.default_engine(); dev = torch.device("cuda:0")
M = np.vstack([workloads.double_gauss(0), [math.inf, 0.0, 1.0]]); M[-2, 1] = 57.8
pres = Prescription.from_matrix(M); sysd = eng.system(pres)
N, S = 9437184, 12
g = torch.Generator(device=dev); g.manual_seed(1)
for dt, fn in ((torch.float64, eng.ctx.lib.ort_trace_skew_f64), (torch.float32, eng.ctx.lib.ort_trace_skew_f32)):
    y = (torch.rand(N, dtype=dt, device=dev, generator=g) - 0.5) * 28; x = (torch.rand(N, dtype=dt, device=dev, generator=g) - 0.5) * 28
    U = (torch.rand(N, dtype=dt, device=dev, generator=g) - 0.5) * 0.3; V = (torch.rand(N, dtype=dt, device=dev, generator=g) - 0.5) * 0.3
    xv = torch.empty((S, N), dtype=dt, device=dev); yv = torch.empty_like(xv)
    torch.cuda.synchronize()
    for name, fl in (("angles ieee", 0), ("slopes ieee", _capi.ORT_INPUT_SLOPES), ("angles fast", _capi.ORT_FAST_MATH),
                     ("slopes fast", _capi.ORT_FAST_MATH | _capi.ORT_INPUT_SLOPES)):
        def step():
            _capi.check(fn(eng.ctx.h, sysd.h, 0, N, y.data_ptr(), x.data_ptr(), U.data_ptr(), V.data_ptr(),
                           xv.data_ptr(), yv.data_ptr(), N, None, fl | _capi.ORT_DEVICE_PTRS))
        for _ in range(3): step()
        eng.ctx.synchronize(); eng.ctx.timer_start()
        for _ in range(10): step()
        ms = eng.ctx.timer_stop() / 10
        w = 8 if dt == torch.float64 else 4
        print(f"{str(dt):14s} {name:12s} {ms:.4f} ms  {N*S/(ms*1e-3):.3e} intersections/s  {(2*w*N*S + 4*w*N)/ms/1e6:.0f} GB/s (algorithmic)")
