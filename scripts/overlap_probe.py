#!/usr/bin/env python3
"""GPU box: can an HBM-bound pass run BESIDE the FP64-issue-bound config-3 trace kernel (the question behind overlapping the
full_trace placement of one bundle group with the trace of the next)?  Stream A: config 3 statistics-only full_trace calls
(the trace kernel: 112 VGPRs, 4 waves per SIMD, 64 VGPRs per SIMD left free); stream B: device-to-device copies of 1 GiB (a
1 : 1 read : write stream of few-VGPR waves).  Times each alone, then both launched together."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import _capi, api, workloads

dev = torch.device("cuda", 0)
sA = torch.cuda.Stream(dev); sB = torch.cuda.Stream(dev, priority=-1)
eng = ort.HipEngine(0, stream=sA.cuda_stream, fast_math=True)
lib, h = eng.ctx.lib, eng.ctx.h
k = 2048
pres, bundles, axes = workloads.config3(api, k, engine=eng)
nb = len(bundles)
sysd = eng.system(pres); barr = _capi.make_bundles(bundles)
d_axes = torch.from_numpy(axes).to(dev)
cnt = torch.empty(nb, dtype=torch.int64, device=dev); rms = torch.empty(nb, dtype=torch.float64, device=dev)
fl = eng.base_flags | _capi.ORT_DEVICE_PTRS
def trace():
    _capi.check(lib.ort_full_trace_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, None, None, None, None,
                                       cnt.data_ptr(), rms.data_ptr(), fl))
src = torch.empty(1 << 27, dtype=torch.float64, device=dev).normal_(); dst = torch.empty_like(src)
def copy():
    with torch.cuda.stream(sB):
        dst.copy_(src, non_blocking=True)
NA, NB = 20, 40
def run(a, b):
    torch.cuda.synchronize()
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    if b: eb0.record(sB)
    if a: ea0.record(sA)
    for i in range(max(NA if a else 0, NB if b else 0)):
        if b and i < NB: copy()
        if a and i < NA: trace()
    if a: ea1.record(sA)
    if b: eb1.record(sB)
    torch.cuda.synchronize()
    return (ea0.elapsed_time(ea1) / NA if a else None, eb0.elapsed_time(eb1) / NB if b else None)
for _ in range(2): trace(); copy()
ta, _ = run(True, False); _, tb = run(False, True)
tca, tcb = run(True, True)
gb = 2 * src.numel() * 8 / 1e9
print(f"alone:    trace call {ta:.3f} ms; 1 GiB copy {tb:.3f} ms = {gb / tb:.2f} TB/s (read + write)")
print(f"together: trace call {tca:.3f} ms ({tca / ta:.2f} x); copy {tcb:.3f} ms = {gb / tcb:.2f} TB/s ({tb / tcb:.2f} of its rate alone)")
print(f"serial time of one trace + one copy {ta + tb:.3f} ms; together, per pair of them {max(tca * NA, tcb * NB) / min(NA, NB) if False else (tca * NA + 0) / NA:.3f} ms (trace stream) / {tcb:.3f} ms (copy stream)")
