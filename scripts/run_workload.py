#!/usr/bin/env python3
"""One workload through the C ABI, a fixed number of times — the thing rocprofv3 wraps (scripts/pmc_sq.sh) and the
thing interleaved A/B timings call.  Prints one JSON line.

  python scripts/run_workload.py config3 --mode stats|full|lookback|fused [--policy fast|ieee] [--pupil 2048] [--reps 5]
  python scripts/run_workload.py config2 --mode summary|history|stats|full
  python scripts/run_workload.py config1 --mode full|stats [--field 1.0]      (one-call pipeline, host buffers)
  python scripts/run_workload.py config5 --mode stats|hits [--dtype f32|f64] [--instances 10000] [--retrace]
        stats: ort_spot_batch_f32 (one C call, host arrays in, 16 B per (instance, field) out: the statistics kernel);
        hits:  the same number of rays through the Float32 summary kernel (image-plane hits, 8 B per ray out)
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
    ap.add_argument("workload", choices=["config1", "config2", "config3", "config4", "config5"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="config5: arithmetic of the pupil trace")
    ap.add_argument("--instances", type=int, default=10000, help="config5: perturbed instances")
    ap.add_argument("--retrace", action="store_true", help="print the retrace counters of an -DORT_COUNT_RETRACE build")
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

    if a.workload == "config4":
        # BASELINE config 4 on one GPU: 32 zoom positions x 5 index columns x 5 fields x 512^2 pupil, summary trace into the
        # packed [2][n] hit slab
        k4 = a.pupil or 512
        mats = np.array([workloads.double_gauss(line, -1.5 + 3.0 * z / 31) for z in range(32) for line in (0, 1, 2, 1, 2)])
        plan = batch.ImageHitsPlan(mats, workloads.DG_A, workloads.DG_H, (0.0, 0.5, 0.7, 0.85, 1.0), k4, engine=eng)
        hits = plan.new_hits()
        ms = timed(lambda: plan.trace(hits))
        res.update(rays=plan.n_rays, intersections=plan.n_rays * 12, ms=ms, intersections_per_s=plan.n_rays * 12 / (ms * 1e-3),
                   checksum=float(torch.nan_to_num(hits).sum().item()))
        print(json.dumps(res), flush=True)
        return

    if a.workload == "config5":
        # BASELINE config 5: 10^4 perturbed Double-Gauss instances x 2 fields x 256 x 128 half pupil (stats) — or the same
        # 6.55e8 rays as instances / 2 x 2 fields x 256 x 256 full pupil through the summary kernel (hits)
        k5 = a.pupil or 256
        dt = np.float32 if a.dtype == "f32" else np.float64
        mats = workloads.config5(None, ninst=a.instances)
        if a.mode == "stats":
            cols = batch.split_columns(mats)                      # R, t, n as the C ABI takes them, once
            fn = lambda: batch.spot_batch(cols, workloads.DG_A, workloads.DG_H, (0.0, 1.0), k5, engine=eng, dtype=dt)
            rays = a.instances * 2 * k5 * (k5 // 2)
        else:
            plan = batch.ImageHitsPlan(mats[:a.instances // 2], workloads.DG_A, workloads.DG_H, (0.0, 1.0), k5, engine=eng, dtype=dt)
            hits = plan.new_hits()
            fn = lambda: (plan.trace(hits), eng.ctx.synchronize())
            rays = plan.n_rays
        fn(); eng.ctx.synchronize()
        if a.retrace:
            cnt = (C.c_ulonglong * 2)()
            lib.ort_debug_retrace_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            lib.ort_debug_retrace_counts.restype = C.c_int
            _capi.check(lib.ort_debug_retrace_counts(h, cnt, 1))
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); r = fn(); eng.ctx.synchronize(); ts.append(time.perf_counter() - t0)
        res.update(dtype=a.dtype, instances=a.instances, rays=rays, intersections=rays * 12, wall_ms_median=float(np.median(ts)) * 1e3,
                   wall_ms_min=float(np.min(ts)) * 1e3, intersections_per_s=rays * 12 / float(np.median(ts)))
        if a.mode == "stats":
            res.update(mean_rms=float(np.nanmean(r["rms"])), count_mean=float(r["count"].mean()))
        if a.retrace:
            _capi.check(lib.ort_debug_retrace_counts(h, cnt, 0))
            res.update(tile_waves=int(cnt[0]), tile_waves_retraced=int(cnt[1]), retrace_fraction=cnt[1] / max(1, cnt[0]))
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
            vec = [torch.zeros((nb, 2 * rpb), dtype=torch.float64, device=dev) for _ in range(4)]
            ptrs = tuple(v.data_ptr() for v in vec)
        ffl = fl | (_capi.ORT_FT_LOOKBACK if a.mode == "lookback" else 0) | (_capi.ORT_FT_FUSED if a.mode == "fused" else 0)
        ms = timed(lambda: _capi.check(lib.ort_full_trace_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, *ptrs,
                                                              cnt.data_ptr(), rms.data_ptr(), ffl)))
        res.update(survivors=int(cnt.sum().item()) // 2, mean_rms=float(rms.mean().item()))
        if a.mode != "stats":      # bit-level checksum of the four slabs (unwritten entries are zero): routes must agree
            res.update(checksum=[int(v.view(torch.int64).sum().item()) for v in vec],
                       rms_bits=int(rms.view(torch.int64).sum().item()))
    res.update(ms=ms, intersections_per_s=N * S / (ms * 1e-3))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
