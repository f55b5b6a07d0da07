#!/usr/bin/env python3
"""GPU box: does the history kernel's rate depend on WHERE its output arrays lie?  BASELINE config 2 (the headline launch) into
K separately allocated pairs of [S][N] Float64 arrays of ONE process, each timed over 300 launches behind a 1-s pre-roll, twice
(interleaved): the same kernel, the same bytes, only the addresses differ.   python scripts/history_placement_probe.py [K = 8]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opticalraytracing_jl_amd as ort                      # noqa: E402
from opticalraytracing_jl_amd import _capi, api, workloads   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
eng = ort.HipEngine(0, fast_math=True)
lib, h = eng.ctx.lib, eng.ctx.h
pres, bundles, axes = workloads.config2(api, 1024, engine=ort.default_engine())
nb, k = len(bundles), 1024
N, S = nb * k * k, pres.rows - 1
sysd = eng.system(pres); barr = _capi.make_bundles(bundles)
d_axes = torch.from_numpy(axes).to(dev)
fl = eng.base_flags | _capi.ORT_DEVICE_PTRS
pairs = [(torch.empty((S, N), dtype=torch.float64, device=dev), torch.empty((S, N), dtype=torch.float64, device=dev)) for _ in range(K)]


def step(xv, yv):
    out = _capi.ort_grid_out_f64(); out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
    return lambda: _capi.check(lib.ort_trace_grid_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, C.byref(out), fl))


f0 = step(*pairs[0])
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    for _ in range(100):
        f0()
    eng.ctx.synchronize()
for rep in range(2):
    for i, (xv, yv) in enumerate(pairs):
        f = step(xv, yv)
        for _ in range(20):
            f()
        eng.ctx.synchronize()
        eng.ctx.timer_start()
        for _ in range(300):
            f()
        ms = eng.ctx.timer_stop() / 300
        print(f"pair {i} rep {rep}: {ms:.4f} ms = {16.0 * N * S / (ms * 1e-3) / 1e9:7.1f} GB/s = {16.0 * N * S / (ms * 1e-3) / 8e12:.3f} of the HBM spec   x = {xv.data_ptr():#x}", flush=True)
