#!/usr/bin/env python3
"""GPU box: the headline launch with a PADDED leading dimension (the ABI's ld >= N): ld = N + pad for a few pads, into each of K
pairs of arrays of one process, three interleaved repetitions.   python scripts/history_ld_probe.py [K = 4]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opticalraytracing_jl_amd as ort                      # noqa: E402
from opticalraytracing_jl_amd import _capi, api, workloads   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PADS = [0, 32, 96, 512, 8192, 8224]
dev = torch.device("cuda", 0)
eng = ort.HipEngine(0, fast_math=True)
lib, h = eng.ctx.lib, eng.ctx.h
pres, bundles, axes = workloads.config2(api, 1024, engine=ort.default_engine())
nb, k = len(bundles), 1024
N, S = nb * k * k, pres.rows - 1
sysd = eng.system(pres); barr = _capi.make_bundles(bundles)
d_axes = torch.from_numpy(axes).to(dev)
fl = eng.base_flags | _capi.ORT_DEVICE_PTRS
LDMAX = N + max(PADS)
pairs = [(torch.empty((S, LDMAX), dtype=torch.float64, device=dev), torch.empty((S, LDMAX), dtype=torch.float64, device=dev)) for _ in range(K)]


def step(xv, yv, ld):
    out = _capi.ort_grid_out_f64(); out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), ld
    return lambda: _capi.check(lib.ort_trace_grid_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, C.byref(out), fl))


f0 = step(*pairs[0], N)
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    for _ in range(100):
        f0()
    eng.ctx.synchronize()
for rep in range(3):
    for i, (xv, yv) in enumerate(pairs):
        row = []
        for pad in PADS:
            f = step(xv, yv, N + pad)
            for _ in range(20):
                f()
            eng.ctx.synchronize()
            eng.ctx.timer_start()
            for _ in range(200):
                f()
            ms = eng.ctx.timer_stop() / 200
            row.append(16.0 * N * S / (ms * 1e-3) / 8e12)
        print(f"rep {rep} pair {i}: " + "  ".join(f"pad {p}: {r:.3f}" for p, r in zip(PADS, row)), flush=True)
