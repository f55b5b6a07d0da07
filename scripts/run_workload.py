#!/usr/bin/env python3
"""One workload through the C ABI, a fixed number of times — the thing rocprofv3 wraps (scripts/pmc_sq.sh) and the
thing interleaved A/B timings call.  Prints one JSON line.

  python scripts/run_workload.py config3 --mode stats|full|lookback [--policy fast|ieee] [--pupil 2048] [--reps 5]
  python scripts/run_workload.py config2 --mode summary|history|stats|full
  python scripts/run_workload.py config1 --mode full|stats [--field 1.0]      (one-call pipeline, host buffers)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["config1", "config2", "config3"])
    ap.add_argument("--mode", default="stats")
    ap.add_argument("--policy", default="fast", choices=["fast", "ieee"])
    ap.add_argument("--pupil", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--field", type=float, default=1.0)
    ap.add_argument("--lib", default=None, help="another build of libort_hip.so (A/B)")
    ap.add_argument("--raybasis", action="store_true", help="config2 / config3: the finite-conjugate launch rule (per-ray angles, "
                    "src/PupilSampling.jl:124-127) with the object 900 mm in front of the first surface")
    a = ap.parse_args()
    if a.lib:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.environ["ORT_HIP_LIB"] = a.lib if os.path.isabs(a.lib) else os.path.join(root, a.lib)
    import torch
    import opticalraytracing_jl_amd as ort
    from opticalraytracing_jl_amd import _capi, api, batch, workloads
    eng = ort.HipEngine(fast_math=(a.policy == "fast"))
    ort.set_default_engine(eng)
    lib, h = eng.ctx.lib, eng.ctx.h
    res = {"workload": a.workload, "mode": a.mode, "policy": a.policy, "reps": a.reps, "raybasis": bool(a.raybasis)}

    def timed(fn):
        fn(); eng.ctx.synchronize()
        eng.ctx.timer_start()
        for _ in range(a.reps):
            fn()
        return eng.ctx.timer_stop() / a.reps

    if a.workload == "config1":
        from tests import common as cm
        mats = cm.cooke()[None]
        if a.mode == "full":
            fn = lambda: batch.full_trace_systems(mats, cm.COOKE_A, cm.COOKE_H, (a.field,), 64, engine=eng)
        else:
            fn = lambda: batch.spot_batch(mats, cm.COOKE_A, cm.COOKE_H, (a.field,), 64, engine=eng)
        for _ in range(3):
            fn()
        ts = []
        for _ in range(max(a.reps, 20)):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        res.update(wall_us_median=float(np.median(ts)), wall_us_min=float(ts.min()), wall_us_p90=float(np.percentile(ts, 90)))
        if a.mode == "full":
            res.update(rms=r[1][0]["rms"], count=r[1][0]["count"])
        else:
            res.update(rms=float(r["rms"][0, 0]), count=int(r["count"][0, 0]))
        print(json.dumps(res), flush=True)
        return

    k = a.pupil or (2048 if a.workload == "config3" else 1024)
    pres, bundles, axes = (workloads.config3 if a.workload == "config3" else workloads.config2)(api, k, engine=eng)
    dev = torch.device("cuda", 0)
    nb, rpb = len(bundles), k * k
    N, S = nb * rpb, pres.rows - 1
    if a.raybasis:
        for bd in bundles:
            bd["ybar"], bd["z0"] = -40.0 * bd["U"] / 0.3, -900.0
    sysd = eng.system(pres); barr = _capi.make_bundles(bundles)
    d_axes = torch.from_numpy(axes).to(dev)
    fl = eng.base_flags | _capi.ORT_DEVICE_PTRS | (_capi.ORT_RAYBASIS if a.raybasis else 0)
    res.update(rays=N, intersections=N * S, pupil=k)
    if a.mode in ("summary", "history"):
        out = _capi.ort_grid_out_f64()
        if a.mode == "history":
            xv = torch.empty((S, N), dtype=torch.float64, device=dev); yv = torch.empty_like(xv)
            out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
        else:
            bufs = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(4)]
            st = torch.empty(N, dtype=torch.int32, device=dev)
            out.xf, out.yf, out.xs, out.ys = (b.data_ptr() for b in bufs)
            out.status = st.data_ptr()
        ms = timed(lambda: _capi.check(lib.ort_trace_grid_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, C.byref(out), fl)))
    else:
        cnt = torch.empty(nb, dtype=torch.int64, device=dev); rms = torch.empty(nb, dtype=torch.float64, device=dev)
        if a.mode == "stats":
            ptrs = (None, None, None, None)
        else:
            vec = [torch.empty((nb, 2 * rpb), dtype=torch.float64, device=dev) for _ in range(4)]
            ptrs = tuple(v.data_ptr() for v in vec)
        ffl = fl | (_capi.ORT_FT_LOOKBACK if a.mode == "lookback" else 0)
        ms = timed(lambda: _capi.check(lib.ort_full_trace_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, *ptrs,
                                                              cnt.data_ptr(), rms.data_ptr(), ffl)))
        res.update(survivors=int(cnt.sum().item()) // 2, mean_rms=float(rms.mean().item()))
    res.update(ms=ms, intersections_per_s=N * S / (ms * 1e-3))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
