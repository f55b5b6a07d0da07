#!/usr/bin/env python3
"""bench.py — ray-surface intersections/s and achieved HBM GB/s of the skew-trace hot path.

Workload (BASELINE.json configs[1]): Double-Gauss (10 spherical surfaces + stop plane + image
plane, S = 12 loop iterations per ray), 3 fields x 3 index columns, 1024 x 1024 pupil per
bundle, Float64: 9,437,184 rays = 113,246,208 ray-surface intersections per step.
A step = ONE launch of the trace kernel over that batch in history mode — the API-faithful
output of `raytrace(surfaces, y, x, U, V, Vector{RealRay})` (reference
src/PupilSampling.jl:34-65): x and y on every surface, 16 B written per intersection; rays are
generated on the device from the bundle axes (0 B read per ray); inputs resident in HBM.

N > 1: one process per GPU (torchrun), weak scaling: every rank traces its own 9 bundles (its
own zoom position of the same lens), no data-path collective.  The reassembly all-gather of the
image-plane hit points (the last history row) runs once after the timed region and is reported
beside the throughput (`allgather`).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # same guide: 6.29 TB/s measured float4 copy


def cpu_baseline(api, pres, bundles, axes, k_full: int, target_s: float = 12.0):
    """Oracle (C restatement of the reference loop, oracle/ort_oracle.c) timed on this box's host
    cores on a bounded sample of the SAME workload: the first bundles' pupil rows."""
    import numpy as np
    from oracle.cpu import _Sys, _p, lib
    L = lib()
    bd = bundles[0]
    s = _Sys(pres, bd["system"])
    xa = np.ascontiguousarray(axes[bd["xaxis_off"]:bd["xaxis_off"] + k_full])
    S = s.rows - 1

    def run(ny, threads, reps=1):
        ya = np.ascontiguousarray(axes[bd["yaxis_off"]:bd["yaxis_off"] + ny])
        n = ny * k_full
        xv = np.empty((S, n)); yv = np.empty((S, n))
        t0 = time.perf_counter()
        cnt = 0
        for _ in range(reps):
            cnt += L.orc_trace_skew_grid(*s.args(), ny, _p(ya), k_full, _p(xa), bd["U"], bd["V"], _p(xv), _p(yv), n,
                                         None, threads)
        return cnt, time.perf_counter() - t0

    cnt, dt = run(min(8, k_full), 1)                      # calibrate
    rate1 = cnt / dt
    rows = target_s * rate1 / (k_full * S)                # pupil rows worth ~target_s of one core
    ny = int(max(1, min(k_full, rows)))
    reps = int(max(1, round(rows / ny)))
    cnt, dt = run(ny, 1, reps)
    rate1 = cnt / dt
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                        # the GPU box's CPU share for one GPU
    cnt_all, dt_all = run(ny, cores, reps * max(1, cores // 2))
    return {
        "value": rate1, "unit": "ray-surface intersections/s", "cores": 1, "kind": "port",
        "sample": f"bundle 0 of the workload (Double-Gauss d-line, H=0): {reps} pass(es) over the first {ny} of "
                  f"{k_full} pupil rows x {k_full} columns x S={S} = {cnt} intersections in {dt:.2f} s, 1 thread, "
                  f"history written; oracle/ort_oracle.c (gcc -O2 -ffp-contract=off)",
        "all_cores": {"value": cnt_all / dt_all, "cores": cores,
                      "sample": f"{cnt_all} intersections in {dt_all:.2f} s, OpenMP static over {cores} threads"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pupil", type=int, default=1024, help="pupil grid edge per bundle (config 2: 1024)")
    ap.add_argument("--policy", default="fast", choices=["fast", "ieee"],
                    help="arithmetic policy of the timed kernel: fast = direction-cosine form (<= 1e-12 relative vs "
                         "the reference sequence, bar 1e-10); ieee = the reference's IEEE operation sequence "
                         "(bit-identical to the CPU oracle).  The other policy is timed too and reported beside it.")
    ap.add_argument("--fast-math", action="store_true", help="alias of --policy fast")
    ap.add_argument("--no-lds", action="store_true", help="surface table through scalar loads instead of LDS")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="history", choices=["history", "summary", "full_trace"])
    args = ap.parse_args()

    import numpy as np
    import torch
    import ctypes as C
    import opticalraytracing_jl_amd as ort
    from opticalraytracing_jl_amd import _capi, api, dist as odist, workloads

    if args.fast_math:
        args.policy = "fast"
    args.fast_math = args.policy == "fast"
    rank, world, local_rank = odist.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ORT_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of N > 1 on a 1-GPU box
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: one process per GPU")
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # where collective buffers live
    dist = None
    if world > 1:
        dist = odist.init_process_group(backend)

    stream = torch.cuda.current_stream(dev)
    flags0 = (_capi.ORT_FAST_MATH if args.fast_math else 0) | (_capi.ORT_NO_LDS if args.no_lds else 0)
    eng = ort.HipEngine(local_dev, stream=stream.cuda_stream, fast_math=args.fast_math, use_lds=not args.no_lds)
    ort.set_default_engine(eng)
    info = eng.ctx.device_info()

    # ---- untimed setup: solve (paraxial + ABCD kernels), bundle descriptors, device buffers ----
    k = args.pupil
    gap = 0.35 * (rank - (world - 1) / 2.0) if world > 1 else 0.0          # per-rank zoom position
    pres, bundles, axes = workloads.config2(api, k, engine=eng, gap_shift=gap)
    nb, rpb = len(bundles), k * k
    N, S = nb * rpb, pres.rows - 1
    inter = N * S
    sysd = eng.system(pres)
    barr = _capi.make_bundles(bundles)
    d_axes = torch.from_numpy(axes).to(dev)
    out = _capi.ort_grid_out_f64()
    keep = []
    if args.mode == "history":
        xv = torch.empty((S, N), dtype=torch.float64, device=dev)
        yv = torch.empty((S, N), dtype=torch.float64, device=dev)
        out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
        keep += [xv, yv]
        algo_bytes = 16.0 * inter
        hits = (xv[S - 1], yv[S - 1])
    elif args.mode == "summary":
        xf = torch.empty(N, dtype=torch.float64, device=dev); yf = torch.empty_like(xf)
        xs = torch.empty_like(xf); ys = torch.empty_like(xf)
        st = torch.empty(N, dtype=torch.int32, device=dev)
        out.xf, out.yf, out.xs, out.ys, out.status = xf.data_ptr(), yf.data_ptr(), xs.data_ptr(), ys.data_ptr(), st.data_ptr()
        keep += [xf, yf, xs, ys, st]
        algo_bytes = 36.0 * N
        hits = (xf, yf)
    else:
        cap = 2 * rpb
        ex = torch.empty((nb, cap), dtype=torch.float64, device=dev); ey = torch.empty_like(ex)
        rho = torch.empty_like(ex); th = torch.empty_like(ex)
        cnt = torch.empty(nb, dtype=torch.int64, device=dev); rms = torch.empty(nb, dtype=torch.float64, device=dev)
        keep += [ex, ey, rho, th, cnt, rms]
        algo_bytes = 32.0 * N * (math.pi / 4) * 2
        hits = None

    torch.cuda.synchronize(dev)
    lib, h = eng.ctx.lib, eng.ctx.h
    fl = flags0 | _capi.ORT_DEVICE_PTRS

    def step():
        if args.mode == "full_trace":
            _capi.check(lib.ort_full_trace_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                               ex.data_ptr(), ey.data_ptr(), rho.data_ptr(), th.data_ptr(),
                                               cnt.data_ptr(), rms.data_ptr(), fl))
        else:
            _capi.check(lib.ort_trace_grid_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                               C.byref(out), fl))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        odist.barrier(local_dev)
    torch.cuda.synchronize(dev)
    eng.ctx.timer_start()                                   # hipEventRecord on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = eng.ctx.timer_stop()                            # hipEventSynchronize + elapsed
    torch.cuda.synchronize(dev)
    if dist is not None:
        odist.barrier(local_dev)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall, ev_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tw[0]), float(tw[1])

    # ---- the other arithmetic policy, same launch, reported beside the headline (rank 0, N = 1) ----
    other = None
    if world == 1 and args.mode != "full_trace":
        ofl = (fl & ~_capi.ORT_FAST_MATH) if args.fast_math else (fl | _capi.ORT_FAST_MATH)
        def ostep():
            _capi.check(lib.ort_trace_grid_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, C.byref(out), ofl))
        for _ in range(2):
            ostep()
        torch.cuda.synchronize(dev)
        eng.ctx.timer_start()
        osteps = max(3, args.steps // 2)
        for _ in range(osteps):
            ostep()
        oms = eng.ctx.timer_stop() / osteps
        other = {"policy": "ieee" if args.fast_math else "fast", "kernel_ms": oms,
                 "value": inter / (oms * 1e-3), "achieved_GBps": algo_bytes / (oms * 1e-3) / 1e9,
                 "frac": algo_bytes / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # ---- the single reassembly all-gather of image-plane hits (untimed region, timed alone) ----
    gather = None
    if dist is not None and hits is not None:
        torch.cuda.synchronize(dev); odist.barrier(local_dev)
        g0 = time.perf_counter()
        gx, gy = odist.allgather_hits(hits[0].to(cdev), hits[1].to(cdev))
        torch.cuda.synchronize(dev); odist.barrier(local_dev)
        gdt = time.perf_counter() - g0
        gbytes = 16.0 * N * world
        gather = {"ms": gdt * 1e3, "bytes_assembled_per_rank": gbytes, "GBps_per_rank": gbytes / gdt / 1e9,
                  "entries": int(gx.numel())}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    kernel_ms = ev_ms / args.steps
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    value = inter * world * args.steps / wall
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.mode}_k{k}_{'fast' if args.fast_math else 'ieee'}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "ray-surface intersections/sec (skew real-ray trace, per-surface history) + achieved HBM GB/s vs roofline",
        "value": value, "unit": "ray-surface intersections/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Double-Gauss 10 spherical surfaces + stop + image (S={S}), 3 fields x 3 index "
                               f"columns, {k}x{k} pupil, Float64, {args.mode} output, "
                               f"{'fast' if args.fast_math else 'ieee'} arithmetic policy",
                   "rays_per_step_per_gpu": N, "intersections_per_step_per_gpu": inter,
                   "surface_table": "scalar-loads" if args.no_lds else "lds",
                   "parallelism": f"{world} x independent bundle shards (weak)", "device": info["name"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "ort::k_trace<double,...>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "frac_of_measured_copy_rate": achieved / HBM_MEASURED_COPY_GBS},
    }
    if other is not None:
        res["other_policy"] = other
    if gather is not None:
        res["allgather"] = gather
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(api, pres, bundles, axes, k)
        res["cpu_baseline"]["gpu_over_cpu_1core"] = value / res["cpu_baseline"]["value"]
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
