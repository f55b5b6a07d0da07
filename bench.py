#!/usr/bin/env python3
"""bench.py — ray-surface intersections/s and achieved HBM GB/s of the skew-trace hot path.

N = 1 (BASELINE.json configs[1]): Double-Gauss (10 spherical surfaces + stop plane + image plane, S = 12 loop
iterations per ray), 3 fields x 3 index columns, 1024 x 1024 pupil per bundle, Float64: 9,437,184 rays =
113,246,208 ray-surface intersections per step.  A step = ONE launch of the trace kernel over that batch in
history mode — the API-faithful output of `raytrace(surfaces, y, x, U, V, Vector{RealRay})` (reference
src/PupilSampling.jl:34-65): x and y on every surface, 16 B written per intersection; rays are generated on the
device from the bundle axes (0 B read per ray); inputs resident in HBM.  Beside the headline the same run
measures: the other arithmetic policy, a sustained figure over >= 1 s of launches, summary mode, BASELINE
config 3 (aspheric, 2048^2 pupil) through full_trace with the stop-filter compaction, an oracle check of a
strided sample of the timed kernel's output, and the CPU baseline.

N > 1: the SAME workload and metric under WEAK scaling — the path shards over independent (system, field, index column,
pupil row) units (src/PupilSampling.jl:34-65 is a pure function of its arguments), so rank r traces its own instance of
the batch (zoom position r of the same Double-Gauss: the same 9 bundles x 1024^2 rays x S = 12) with NO collective in the
data path; the K steps are bracketed by a barrier + synchronize on both sides, the slowest rank's time counts, `value` =
N x 113,246,208 x K / that time.  One curve over N = 1, 2, 4, 8 is therefore one workload.
The exchange step north_star names — reassembling the image-plane hits of a sharded sweep — is measured in the same run
and reported under `extra.config4_allgather` (BASELINE.json configs[3]): zoom-lens sweep, 32 positions x 5 index columns
x 5 fields x 512^2 pupil = 209,715,200 rays, 2.52e9 intersections per step, STRONG scaling: the 800 bundles are split
into contiguous rank-ordered slabs, every rank traces its slab in summary mode into a packed [2][n] hit slab, and ONE
RCCL all-gather per step reassembles the hits of the whole sweep on every rank in the reference's append order
(src/PupilSampling.jl:134-137), inside that leg's timed region, on the communicator's own stream, overlapped with the
next step's trace.  The leg carries its gather-exclusive rate, the all-gather alone, rank 0's bit-for-bit check of the
gathered hits against its own single-rank trace and the spot-statistics form of the sweep (one `ort_spot_batch` call per
rank and a 16-B-per-bundle all-gather — nothing ray-sized moves).  `--workload config4` runs that leg alone.

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
METRIC = "ray-surface intersections/sec (skew real-ray trace) + achieved HBM GB/s vs roofline"


def newest_profile(suffix: str):
    """profiles/rNN_<suffix> of the latest round that committed one (static annotations only), or None."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return fs[-1] if fs else None


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


def verify_sample(pres, bundles, axes, k, xv, yv, policy: str, stride: int = 4099):
    """Re-trace a strided sample of the timed kernel's rays with the CPU oracle and compare with what the kernel
    left in HBM.  ieee: bit-identical (NaN patterns included); fast: NaN patterns identical and <= 1e-10 relative."""
    import numpy as np
    import torch
    from oracle.cpu import OracleEngine
    N = xv.shape[1]
    idx = np.unique(np.concatenate([np.arange(0, N, stride), [N - 1]]))
    ti = torch.from_numpy(idx).to(xv.device)
    gx = xv.index_select(1, ti).cpu().numpy(); gy = yv.index_select(1, ti).cpu().numpy()
    rpb = k * k
    b = idx // rpb; r = idx - b * rpb
    iy, ix = r // k, r - (r // k) * k
    orc = OracleEngine(nthreads=4)
    worst, pat, exact = 0.0, 0, True
    for bi in np.unique(b):
        sel = b == bi
        bd = bundles[int(bi)]
        y = axes[bd["yaxis_off"] + iy[sel]]; x = axes[bd["xaxis_off"] + ix[sel]]
        ox, oy = orc.skew(pres, y, x, np.full(y.size, bd["U"]), np.full(y.size, bd["V"]), isys=bd["system"])
        for g, o in ((gx[:, sel], ox), (gy[:, sel], oy)):
            pat += int((np.isnan(g) != np.isnan(o)).sum())
            exact = exact and bool(np.array_equal(g, o, equal_nan=True))
            d = np.abs(g - o) / np.maximum(1.0, np.abs(o))
            d = d[np.isfinite(d)]
            if d.size:
                worst = max(worst, float(d.max()))
    ok = (pat == 0) and (exact if policy == "ieee" else worst <= 1e-10)
    return {"verified": bool(ok), "sample_rays": int(idx.size), "sample": f"every {stride}th ray of the last timed launch's history, all {gx.shape[0]} surfaces",
            "checker": "oracle/ort_oracle.c (CPU)", "bit_identical": bool(exact), "max_rel_deviation": worst,
            "nan_pattern_mismatches": pat, "bar": "bit-identical" if policy == "ieee" else "<= 1e-10 relative, NaN patterns identical"}


def timed_launches(eng, fn, steps: int, warmup: int = 2, preroll_s: float = 0.0):
    """hipEvent-timed back-to-back launches on the engine's stream -> ms per launch.  preroll_s > 0: the launches are preceded
    by that many seconds of the same call back to back (the regime the headline is timed in behind its `sustained` leg): the
    first ~10 ms of work after an idle period run at a lower clock (profiles/r04_config3_rep_sweep.log)."""
    for _ in range(warmup):
        fn()
    eng.ctx.synchronize()
    if preroll_s > 0.0:
        t_end = time.perf_counter() + preroll_s
        while time.perf_counter() < t_end:
            for _ in range(16):
                fn()
            eng.ctx.synchronize()
    eng.ctx.timer_start()
    for _ in range(steps):
        fn()
    return eng.ctx.timer_stop() / steps


def bench_single(args, torch, rank, world, local_rank, emit=True):
    """The headline workload (BASELINE config 2, one trace launch per step).  world > 1: WEAK scaling — every rank traces
    its own instance of the batch (zoom position `rank` of the same Double-Gauss: the same 9 bundles x k^2 rays x S = 12),
    no collective anywhere in the data path; the K steps are bracketed by a barrier + synchronize on both sides and the
    slowest rank's time counts.  Returns the result dict on rank 0 (printed when `emit`)."""
    import ctypes as C
    import numpy as np
    import opticalraytracing_jl_amd as ort
    from opticalraytracing_jl_amd import _capi, api, dist as odist, workloads

    fast = args.policy == "fast"
    backend = os.environ.get("ORT_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of N > 1 on a 1-GPU box
    if world > 1 and backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPUs: one process per GPU")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    stream = torch.cuda.current_stream(dev)
    flags0 = _capi.ORT_FAST_MATH if fast else 0
    eng = ort.HipEngine(local_dev, stream=stream.cuda_stream, fast_math=fast)
    ort.set_default_engine(eng)
    info = eng.ctx.device_info()
    lib, h = eng.ctx.lib, eng.ctx.h
    fl = flags0 | _capi.ORT_DEVICE_PTRS

    # ---- untimed setup: solve (paraxial + ABCD kernels), bundle descriptors, device buffers ----
    k = args.pupil
    dist = odist.init_process_group(backend) if world > 1 else None
    # rank r > 0: zoom position r of BASELINE config 4's sweep (the two air gaps around the stop moved by +-0.1 r mm):
    # another prescription, the same shape and cost; rank 0 = the N = 1 system
    pres, bundles, axes = workloads.config2(api, k, engine=eng, gap_shift=0.1 * rank)
    nb, rpb = len(bundles), k * k
    N, S = nb * rpb, pres.rows - 1
    inter = N * S
    sysd = eng.system(pres)
    barr = _capi.make_bundles(bundles)
    d_axes = torch.from_numpy(axes).to(dev)

    def grid_step(out, flags):
        def f():
            _capi.check(lib.ort_trace_grid_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k, C.byref(out), flags))
        return f

    out = _capi.ort_grid_out_f64()
    xv = yv = None
    placement = None
    if args.mode == "history":
        # Where the two 906-MB output arrays come to lie decides 0.71 .. 0.85 of the HBM spec for the SAME kernel (one process, one
        # GPU: profiles/r04_history_placement_probe.log): sixteen candidate pairs are allocated (29 GB for a moment; about one place in five is a fast one), the launch is timed into each and
        # the best-placed pair is kept for the run (opticalraytracing_jl_amd/placement.py); every candidate's time is reported
        from opticalraytracing_jl_amd.placement import best_placed

        def time_pair(pair):
            o = _capi.ort_grid_out_f64(); o.xv, o.yv, o.ld = pair[0].data_ptr(), pair[1].data_ptr(), N
            return timed_launches(eng, grid_step(o, fl), 30, warmup=10)
        mk = lambda: (torch.empty((S, N), dtype=torch.float64, device=dev), torch.empty((S, N), dtype=torch.float64, device=dev))
        if args.placement_candidates > 1:                       # (settled clocks for the comparison)
            warm = mk(); w = _capi.ort_grid_out_f64(); w.xv, w.yv, w.ld = warm[0].data_ptr(), warm[1].data_ptr(), N
            timed_launches(eng, grid_step(w, fl), 10, warmup=2, preroll_s=0.5)
            del warm, w
        (xv, yv), placement = best_placed(mk, time_pair, args.placement_candidates)
        torch.cuda.empty_cache()                                # the other candidates go back to the driver
        if placement["candidates_ms"]:
            placement.update(candidates_frac_of_hbm_spec=[16.0 * inter / (m * 1e-3) / 1e9 / HBM_PEAK_GBS for m in placement["candidates_ms"]],
                             what="the headline launch timed (30 launches behind 10) into each of the candidate pairs of output arrays "
                                  "allocated one after the other; the run uses the fastest pair")
        out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
        algo_bytes = 16.0 * inter
        step = grid_step(out, fl)
    elif args.mode == "summary":
        xf = torch.empty(N, dtype=torch.float64, device=dev); yf = torch.empty_like(xf)
        xs = torch.empty_like(xf); ys = torch.empty_like(xf)
        st = torch.empty(N, dtype=torch.int32, device=dev)
        out.xf, out.yf, out.xs, out.ys, out.status = xf.data_ptr(), yf.data_ptr(), xs.data_ptr(), ys.data_ptr(), st.data_ptr()
        algo_bytes = 36.0 * N
        step = grid_step(out, fl)
    else:
        cap = 2 * rpb
        ex = torch.empty((nb, cap), dtype=torch.float64, device=dev); ey = torch.empty_like(ex)
        rho = torch.empty_like(ex); th = torch.empty_like(ex)
        cnt = torch.empty(nb, dtype=torch.int64, device=dev); rms = torch.empty(nb, dtype=torch.float64, device=dev)
        algo_bytes = None                                     # survivors only: known after the first call

        def step():
            _capi.check(lib.ort_full_trace_f64(h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                               ex.data_ptr(), ey.data_ptr(), rho.data_ptr(), th.data_ptr(),
                                               cnt.data_ptr(), rms.data_ptr(), fl | (_capi.ORT_FT_LOOKBACK if args.ft_lookback else 0) | (_capi.ORT_FT_FUSED if args.ft_fused else 0)))

    step(); torch.cuda.synchronize(dev)                     # first launch: allocations, code load
    if algo_bytes is None:
        algo_bytes = 64.0 * float(cnt.sum().item()) / 2           # 2 halves x 32 B per survivor
    # ---- sustained: >= 1 s of back-to-back launches.  It runs BEFORE the W warm-up + K timed steps so that those
    # see the clock the chip settles at under this load (FP64 + ~5 TB/s of stores) rather than its ramp from idle ----
    sustained = None
    if args.sustain_s > 0:
        batch, n_l, t1 = 200, 0, time.perf_counter()
        eng.ctx.timer_start()
        while True:
            for _ in range(batch):
                step()
            n_l += batch
            eng.ctx.synchronize()
            if time.perf_counter() - t1 >= args.sustain_s:
                break
        s_ms = eng.ctx.timer_stop() / n_l
        sustained = {"launches": n_l, "seconds": time.perf_counter() - t1, "kernel_ms": s_ms, "value": inter / (s_ms * 1e-3),
                     "achieved_GBps": algo_bytes / (s_ms * 1e-3) / 1e9, "frac": algo_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "order": "before the warm-up and the timed steps"}
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
    wall = time.perf_counter() - t0
    per_rank = None
    if dist is not None:                                    # the slowest rank's time counts; every rank's own time is reported
        tw = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allw = [torch.empty_like(tw) for _ in range(world)]
        dist.all_gather(allw, tw)
        per_rank = [float(t[0]) / args.steps * 1e3 for t in allw]
        wall = max(float(t[0]) for t in allw)

    kernel_ms = ev_ms / args.steps
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    value = world * inter * args.steps / wall

    # ---- correctness of what was just timed: a strided sample of the history against the CPU oracle ----
    verify = None
    if args.mode == "history" and not args.no_verify:
        try:
            verify = verify_sample(pres, bundles, axes, k, xv, yv, args.policy)
        except Exception as exc:                                # noqa: BLE001 — reported as not verified, never hidden
            verify = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}

    if dist is not None and verify is not None:             # every rank checked its own history: all of them must agree
        vf = torch.tensor([1 if verify.get("verified") else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(vf, op=dist.ReduceOp.MIN)
        verify["verified_on_every_rank"] = bool(int(vf[0]))
        verify["verified"] = bool(verify.get("verified")) and bool(int(vf[0]))

    # ---- the other arithmetic policy, same launch, reported beside the headline ----
    other = None
    if args.mode != "full_trace" and world == 1:
        ofl = (fl & ~_capi.ORT_FAST_MATH) if fast else (fl | _capi.ORT_FAST_MATH)
        ostep = grid_step(out, ofl)
        osus = None
        if args.sustain_s > 0:                                          # the headline's protocol: its settled rate first
            n_l, t1 = 0, time.perf_counter()                            # (>= sustain_s / 2 of launches) ...
            eng.ctx.timer_start()
            while True:
                for _ in range(100):
                    ostep()
                n_l += 100
                eng.ctx.synchronize()
                if time.perf_counter() - t1 >= args.sustain_s / 2:
                    break
            osus = eng.ctx.timer_stop() / n_l
        oms = timed_launches(eng, ostep, max(3, args.steps))            # ... then the same number of timed launches
        # what bounds the reference-sequence kernel: FP64 issue slots (committed SQ counter pass of this kernel, static)
        valu_note = None
        sqh = newest_profile("pmc_sq_history_both_policies.json")
        if k == 1024 and sqh:
            try:
                sq = json.load(open(sqh))
                c = sq[[kk for kk in sq if f"<double, {0 if fast else 1}," in kk][0]]
                if "SQ_BUSY_CU_CYCLES" in c:
                    valu_note = {"instructions_per_intersection": c["SQ_INSTS_VALU"] * 64.0 / inter,
                                 "issue_slot_utilisation": c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CU_CYCLES"],
                                 "source": f"profiles/{os.path.basename(sqh)} (static; scripts/final_profile.sh)"}
            except Exception:                                   # noqa: BLE001 — an optional annotation
                valu_note = None
        other = {"policy": "ieee" if fast else "fast", "kernel_ms": oms, "value": inter / (oms * 1e-3), "valu_roofline": valu_note,
                 "sustained_kernel_ms": osus, "sustained_frac": None if osus is None else algo_bytes / (osus * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "achieved_GBps": algo_bytes / (oms * 1e-3) / 1e9, "frac": algo_bytes / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "parity": "bit-identical to the CPU oracle (reference operation sequence)" if fast else
                           "<= 1e-10 relative, no status flip observed on any ray (tests/test_gpu_parity.py::_fast_attribution)"}

    # ---- extras: summary mode (config 2), BASELINE config 3 through full_trace + compaction ----
    def run_extras():
        """summary mode (config 2), BASELINE config 3 through both full_trace routes + statistics-only, config 5."""
        nonlocal xv, yv
        extra = {}
        xv = yv = None
        out.xv = out.yv = None
        torch.cuda.empty_cache()
        xf = torch.empty(N, dtype=torch.float64, device=dev); yf = torch.empty_like(xf)
        xs = torch.empty_like(xf); ys = torch.empty_like(xf)
        st = torch.empty(N, dtype=torch.int32, device=dev)
        so = _capi.ort_grid_out_f64()
        so.xf, so.yf, so.xs, so.ys, so.status = xf.data_ptr(), yf.data_ptr(), xs.data_ptr(), ys.data_ptr(), st.data_ptr()
        ms = timed_launches(eng, grid_step(so, fl), args.steps, preroll_s=0.25)
        extra["config2_summary"] = {
            "workload": f"same batch, summary output (x_f, y_f, x_stop, y_stop, status: 36 B per ray), {args.policy} policy",
            "kernel_ms": ms, "value": inter / (ms * 1e-3), "algorithmic_bytes_per_launch": 36.0 * N,
            "achieved_GBps": 36.0 * N / (ms * 1e-3) / 1e9, "frac": 36.0 * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "bound": "FP64 VALU (3 B per intersection out)"}
        # the bound that applies here, from the committed SQ counter passes of this kernel (static, not measured in this run)
        sqp = newest_profile("pmc_sq_summary_both_policies.json")
        if args.pupil == 1024 and sqp:
            try:
                sq = json.load(open(sqp))
                key = [k for k in sq if f"<double, {1 if args.policy == 'fast' else 0}," in k][0]
                c = sq[key]
                extra["config2_summary"]["valu_roofline"] = {
                    "instructions_per_intersection": c["SQ_INSTS_VALU"] * 64.0 / inter,
                    "issue_slots_per_intersection": c["SQ_ACTIVE_INST_VALU"] * 64.0 / inter,
                    "issue_slot_utilisation": c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CU_CYCLES"],
                    "note": "SQ_ACTIVE_INST_VALU (4-cycle issue slots; an FP64 v_rsq / v_rcp takes four) over SQ_BUSY_CU_CYCLES x 4 SIMDs / 4",
                    "source": f"profiles/{os.path.basename(sqp)} (static; scripts/pmc_sq_summary.sh)"}
            except Exception:                                    # noqa: BLE001 — an optional annotation
                pass
        # what was just timed, checked: a strided sample of the summary arrays against the CPU oracle
        try:
            from oracle.cpu import OracleEngine
            idx = np.unique(np.concatenate([np.arange(0, N, 40009), [N - 1]]))
            ti = torch.from_numpy(idx).to(dev)
            gxf, gyf, gst = xf[ti].cpu().numpy(), yf[ti].cpu().numpy(), st[ti].cpu().numpy()
            bi2 = idx // (k * k); r2 = idx - bi2 * k * k
            orc2 = OracleEngine(nthreads=4)
            worst2, bad2 = 0.0, 0
            for bb in np.unique(bi2):
                m = bi2 == bb
                bd = bundles[int(bb)]
                yy = axes[bd["yaxis_off"] + r2[m] // k]; xx = axes[bd["xaxis_off"] + r2[m] % k]
                ox, oy, os2 = orc2.skew(pres, yy, xx, np.full(int(m.sum()), bd["U"]), np.full(int(m.sum()), bd["V"]), isys=bd["system"], want_status=True)
                bad2 += int((gst[m] & 0xffff != os2).sum()) + int((np.isnan(gxf[m]) != np.isnan(ox[-1])).sum())
                for g, o in ((gxf[m], ox[-1]), (gyf[m], oy[-1])):
                    d = np.abs(g - o) / np.maximum(1.0, np.abs(o)); d = d[np.isfinite(d)]
                    worst2 = max(worst2, float(d.max()) if d.size else 0.0)
            ok2 = bool(bad2 == 0 and worst2 <= 1e-10)
            extra["config2_summary"]["verify"] = {"verified": ok2, "sample_rays": int(idx.size), "max_rel_deviation": worst2,
                                                  "status_or_nan_mismatches": bad2, "bar": "<= 1e-10 relative, status identical",
                                                  "checker": "oracle/ort_oracle.c (CPU) on every 40009th ray"}
            extra["config2_summary"]["verified"] = ok2
        except Exception as exc:                                    # noqa: BLE001 — reported as not verified, never hidden
            extra["config2_summary"]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
            extra["config2_summary"]["verified"] = False
        del xf, yf, xs, ys, st
        torch.cuda.empty_cache()
        k3 = args.pupil3
        p3, b3, a3 = workloads.config3(api, k3, engine=eng)
        nb3, rpb3 = len(b3), k3 * k3
        N3, S3 = nb3 * rpb3, p3.rows - 1
        sys3 = eng.system(p3); barr3 = _capi.make_bundles(b3)
        d_a3 = torch.from_numpy(a3).to(dev)
        cnt = torch.empty(nb3, dtype=torch.int64, device=dev); rms = torch.empty(nb3, dtype=torch.float64, device=dev)

        def ft_on(bufs, full, extra_flags=0):
            def f():
                _capi.check(lib.ort_full_trace_f64(h, sys3.h, nb3, barr3, d_a3.data_ptr(), a3.size, k3, k3,
                                                   *((b.data_ptr() for b in bufs) if full else (None, None, None, None)),
                                                   cnt.data_ptr(), rms.data_ptr(), fl | extra_flags))
            return f
        PRE3 = 0.25                                                     # seconds of back-to-back calls ahead of each config-3 timing
        reps3 = max(3, args.steps // 4)
        # the four output slabs (4 x 604 MB) where they are written fastest: the placement pass is HBM-bound and the place of a
        # large buffer decides its store rate — candidates timed through the default route, the best set kept
        from opticalraytracing_jl_amd.placement import best_placed
        mk3 = lambda: tuple(torch.empty((nb3, 2 * rpb3), dtype=torch.float64, device=dev) for _ in range(4))
        first3 = mk3()
        ms_first = timed_launches(eng, ft_on(first3, True), reps3, warmup=1)   # (rounds 1-3 quoted this: a few calls after an idle period,
        del first3                                                             #  into the first allocation)
        nc3 = max(1, min(5, args.placement_candidates))
        if nc3 > 1:
            warm3 = mk3(); timed_launches(eng, ft_on(warm3, True), 4, warmup=1, preroll_s=PRE3); del warm3
        (ex, ey, rho, th), place3 = best_placed(mk3, lambda bufs: timed_launches(eng, ft_on(bufs, True), 8, warmup=2), nc3)
        torch.cuda.empty_cache()

        def ft(full, extra_flags=0):
            return ft_on((ex, ey, rho, th), full, extra_flags)
        ms = timed_launches(eng, ft(True), 4 * reps3, warmup=0, preroll_s=PRE3)
        kept = int(cnt.sum().item()) // 2
        ab = 64.0 * kept
        extra["config3_full_trace"] = {
            "workload": f"BASELINE config 3: Double-Gauss with 4 aspheric (conic + even polynomial) surfaces, 3 fields x 3 index "
                        f"columns, {k3}x{k3} pupil, Float64, full_trace: stop filter + order-preserving ballot / prefix-sum "
                        f"compaction + mirror + rho, theta + RMS, error vectors out; {args.policy} policy",
            "rays": N3, "intersections": N3 * S3, "survivors": kept, "pipeline_ms": ms, "value": N3 * S3 / (ms * 1e-3),
            "pipeline_ms_first_calls": ms_first, "output_placement": place3,
            "timing": f"pipeline_ms: mean of {4 * reps3} calls behind {PRE3} s of the same call back to back (sustained clocks, as the "
                      f"headline); pipeline_ms_first_calls: {reps3} calls after one warm-up call (what rounds 1-3 quoted)",
            "algorithmic_bytes_per_call": ab, "algorithmic_bytes_note": "2 halves x 32 B (ex, ey, rho, theta) per survivor",
            "achieved_GBps": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "bound": "FP64 VALU in the trace kernel (polynomial rows), HBM in the mirror pass",
            "mean_rms": float(rms.mean().item())}
        extra["config3_full_trace"]["hbm_traffic_model_B_per_ray"] = "25 staged + 25 re-read + 50 out = 100 (tile-local compaction + placement)"
        # what was just timed, checked: (1) every bundle's survivor count equals the reference-sequence policy's count of the
        # same call (the stop filter is decided identically, src/PupilSampling.jl:132) and the RMS agrees to 1e-10;
        # (2) a strided sample of one bundle's rays, traced in the timed policy, against the CPU oracle
        try:
            cnt_t, rms_t = cnt.clone(), rms.clone()
            c_i = torch.empty_like(cnt); r_i = torch.empty_like(rms)
            _capi.check(lib.ort_full_trace_f64(h, sys3.h, nb3, barr3, d_a3.data_ptr(), a3.size, k3, k3, None, None, None, None,
                                               c_i.data_ptr(), r_i.data_ptr(), (fl & ~_capi.ORT_FAST_MATH)))
            eng.ctx.synchronize()
            counts_equal = bool(torch.equal(c_i, cnt_t))
            rms_dev = float(((r_i - rms_t).abs() / r_i).max())
            from oracle.cpu import OracleEngine
            b4 = min(4, nb3 - 1)
            bd = b3[b4]
            idx = np.unique(np.concatenate([np.arange(0, rpb3, 10007), [rpb3 - 1]]))
            ti = torch.from_numpy(idx).to(dev)
            sx = torch.empty(rpb3, dtype=torch.float64, device=dev); sy = torch.empty_like(sx)
            sst = torch.empty(rpb3, dtype=torch.int32, device=dev)
            so3 = _capi.ort_grid_out_f64(); so3.xf, so3.yf, so3.status = sx.data_ptr(), sy.data_ptr(), sst.data_ptr()
            _capi.check(lib.ort_trace_grid_f64(h, sys3.h, 1, _capi.make_bundles([bd]), d_a3.data_ptr(), a3.size, k3, k3, C.byref(so3), fl))
            eng.ctx.synchronize()
            yy = a3[bd["yaxis_off"] + idx // k3]; xx = a3[bd["xaxis_off"] + idx % k3]
            orc = OracleEngine(nthreads=4)
            ox, oy, os_ = orc.skew(p3, yy, xx, np.full(idx.size, math.tan(bd["U"])), np.zeros(idx.size), isys=bd["system"], slopes=True,
                                   want_status=True)
            gx, gy, gs = sx[ti].cpu().numpy(), sy[ti].cpu().numpy(), sst[ti].cpu().numpy()
            dev_max = float(max(np.nanmax(np.abs(gx - ox[-1]) / np.maximum(1.0, np.abs(ox[-1]))),
                                np.nanmax(np.abs(gy - oy[-1]) / np.maximum(1.0, np.abs(oy[-1])))))
            status_ok = bool(np.array_equal(gs & 0xffff, os_))
            ok = counts_equal and rms_dev <= 1e-10 and status_ok and dev_max <= 1e-10
            extra["config3_full_trace"]["verify"] = {
                "verified": bool(ok), "survivor_counts_equal_reference_sequence_policy": counts_equal, "rms_max_rel_deviation": rms_dev,
                "oracle_sample_rays": int(idx.size), "oracle_sample_max_rel_deviation": dev_max, "oracle_sample_status_identical": status_ok,
                "checker": "oracle/ort_oracle.c (CPU) on every 10007th ray of bundle %d; counts against the same call in the ieee policy" % b4}
            extra["config3_full_trace"]["verified"] = bool(ok)
            del sx, sy, sst
        except Exception as exc:                                # noqa: BLE001 — reported as not verified, never hidden
            extra["config3_full_trace"]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
            extra["config3_full_trace"]["verified"] = False
        # the default route's outputs, bit for bit, as four checksums (the fused route must reproduce them)
        slab_sums = lambda: [int(v.view(torch.int64).sum().item()) for v in (ex, ey, rho, th)]
        for v in (ex, ey, rho, th):
            v.zero_()
        ft(True)(); eng.ctx.synchronize()
        sums_default, rms_default = slab_sums(), rms.clone()
        ms = timed_launches(eng, ft(True, _capi.ORT_FT_LOOKBACK), 4 * reps3, warmup=1, preroll_s=PRE3)
        extra["config3_full_trace_lookback"] = {
            "workload": "same call with ORT_FT_LOOKBACK: survivors written once at their final place (decoupled look-back), mirror pass",
            "pipeline_ms": ms, "value": N3 * S3 / (ms * 1e-3),
            "hbm_traffic_model_B_per_ray": "25 first half + 25 re-read + 6 rho + 25 mirror = 81"}

        def same_as_default_route(tag, rms_bar):
            """counts identical to, RMS within rms_bar of, the default route's results of the same bundles (itself verified above)"""
            try:
                eng.ctx.synchronize()
                ce = bool(torch.equal(cnt, cnt_t)); rd = float(((rms - rms_t).abs() / rms_t).max())
                ok = ce and rd <= rms_bar and bool(extra["config3_full_trace"].get("verified"))
                extra[tag]["verify"] = {"verified": ok, "survivor_counts_equal_default_route": ce, "rms_max_rel_deviation": rd,
                                        "bar": f"counts identical, RMS within {rms_bar:g} of the default route (verified against the oracle above)"}
            except Exception as exc:                                # noqa: BLE001 — reported as not verified, never hidden
                extra[tag]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
            extra[tag]["verified"] = extra[tag]["verify"]["verified"]
        same_as_default_route("config3_full_trace_lookback", 1e-12)
        ms = timed_launches(eng, ft(True, _capi.ORT_FT_FUSED), 4 * reps3, warmup=1, preroll_s=PRE3)
        extra["config3_full_trace_fused"] = {
            "workload": "same call with ORT_FT_FUSED: offsets, placement of both halves and squared deviations inside the trace launch "
                        "(workgroup i traces tile i and places tile i - lag of an earlier, complete bundle; sc1 hand-off, no fence)",
            "pipeline_ms": ms, "value": N3 * S3 / (ms * 1e-3)}
        try:
            for v in (ex, ey, rho, th):
                v.zero_()
            ft(True, _capi.ORT_FT_FUSED)(); eng.ctx.synchronize()
            sums_fused = slab_sums()
            okf = bool(sums_fused == sums_default and torch.equal(rms.view(torch.int64), rms_default.view(torch.int64)) and torch.equal(cnt, cnt_t)
                       and extra["config3_full_trace"].get("verified"))
            extra["config3_full_trace_fused"]["verify"] = {
                "verified": okf, "slab_checksums_equal_default_route": sums_fused == sums_default,
                "bar": "ex, ey, rho, theta (64-bit integer sums of the four zero-initialised slabs), counts and RMS bit-identical to the "
                       "default route's (verified against the oracle above)"}
        except Exception as exc:                                    # noqa: BLE001 — reported as not verified, never hidden
            extra["config3_full_trace_fused"]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
        extra["config3_full_trace_fused"]["verified"] = extra["config3_full_trace_fused"]["verify"]["verified"]
        ms = timed_launches(eng, ft(False), 4 * reps3, warmup=1, preroll_s=PRE3)
        extra["config3_statistics_only"] = {
            "workload": "same bundles, statistics-only route (count, RMS per bundle: 16 B per bundle out); a workgroup walks spans of "
                        "tiles of its bundle, one (n, mean, M2) partial per wave and span (FT_WALK)",
            "pipeline_ms": ms, "value": N3 * S3 / (ms * 1e-3), "bound": "FP64 VALU", "mean_rms": float(rms.mean().item())}
        same_as_default_route("config3_statistics_only", 1e-10)
        sq3 = newest_profile("sq_config3_trace_before_after.json")       # committed counter + clock pass of this kernel (static)
        if sq3 and fast:
            try:
                c3 = json.load(open(sq3))
                c3 = c3[sorted(c3)[-1]]                                   # the newest build of the pass
                extra["config3_statistics_only"]["valu_roofline"] = {
                    "instructions_per_intersection": c3["valu_per_wave"] * 64.0 / (128.0 * S3),   # lane-instructions, as in config2_summary
                    "issue_slot_utilisation": c3["valu_issue_utilisation"], "effective_clock_GHz": c3["clock_GHz"],
                    "note": "SQ_ACTIVE_INST_VALU over SQ_BUSY_CU_CYCLES; clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time",
                    "source": f"profiles/{os.path.basename(sq3)} (static; scripts/clock_config3.sh)"}
            except Exception:                                   # noqa: BLE001 — an optional annotation
                pass
        del ex, ey, rho, th
        torch.cuda.empty_cache()
        # BASELINE config 5: Seidel / spot Monte-Carlo over perturbed instances, Float32 pupil trace, ONE C call
        from opticalraytracing_jl_amd import batch
        ninst, k5 = args.instances5, 256
        mats5 = workloads.config5(api, ninst=ninst)
        call5 = lambda m: batch.spot_batch(m, workloads.DG_A, workloads.DG_H, (0.0, 1.0), k5, engine=eng, dtype=np.float32)
        # the C ABI takes R, t, n as three [ninst][rows] arrays: split the [ninst][rows][3] matrices once, outside the timed calls
        # (numpy needs ~0.5 ms for it; a tolerance loop does it once)
        cols5 = batch.split_columns(mats5)
        call5(tuple(c[:64] for c in cols5))                          # allocations
        call5(cols5)                                                 # ... at full size
        t_end = time.perf_counter() + 0.25                           # sustained clocks first (timed_launches, preroll_s)
        while time.perf_counter() < t_end:
            call5(cols5)
        walls5, devs5 = [], []
        for _ in range(5):
            eng.ctx.timer_start()
            t0 = time.perf_counter()
            r5 = call5(cols5)
            walls5.append(time.perf_counter() - t0)
            devs5.append(eng.ctx.timer_stop())                       # hipEvents on the engine's stream around the whole call
        dt5 = float(np.median(walls5))
        rays5 = ninst * 2 * k5 * (k5 // 2)
        extra["config5_spot_batch_f32"] = {
            "workload": f"BASELINE config 5: {ninst} perturbed Double-Gauss instances x 2 fields x {k5}x{k5 // 2} half pupil (mirrored, "
                        f"reference mode), Float32 trace, first-order solve + Seidel sums + aiming + spot RMS per instance in ONE "
                        f"C call (ort_spot_batch_f32), host arrays in (R, t, n : [instances][rows], as the C ABI takes them), 16 B per (instance, "
                        f"field) + the first-order struct per instance out",
            "rays": rays5, "intersections": rays5 * 12, "wall_ms": dt5 * 1e3, "wall_ms_runs": [w * 1e3 for w in walls5],
            "device_ms_hip_events": float(np.median(devs5)), "value": rays5 * 12 / dt5,
            "rms_mean": float(np.nanmean(r5["rms"])), "count_mean": float(r5["count"].mean()),
            "bound": "FP32 VALU issue (the statistics kernel walks the tiles of a bundle per workgroup, FT_WALK); wall = median of 5 calls, "
                     "each incl. the H2D of the prescriptions, the setup kernels and the D2H of the results"}
        sq5 = newest_profile("sq_config5.json")                       # committed counter + clock pass of this kernel (static)
        if sq5:
            try:
                c5 = json.load(open(sq5))
                key = [kk for kk in c5 if "stats" in kk and "false, 4, 2>" in kk]
                if key:
                    c5 = c5[key[0]]
                    extra["config5_spot_batch_f32"]["valu_roofline"] = {
                        "instructions_per_intersection": c5["valu_per_intersection"], "effective_clock_GHz": c5["clock_GHz"],
                        "wait_inst_any_share_of_wave_cycles": c5["wait_share"], "kernel_ms_under_counters": c5["kernel_ms"],
                        "fp32_issue_floor_ms": c5["valu_per_intersection"] * rays5 * 12 / 64 / 1024 * 2 / (c5["clock_GHz"] * 1e9) * 1e3,
                        "fp32_issue_floor_ms_at_measured_rate": c5["valu_per_intersection"] * rays5 * 12 / (84.7 * 256) / (c5["clock_GHz"] * 1e9) * 1e3,
                        "note": "floor = VALU wave-instructions / 1024 SIMDs x 2 cycles (a wave64 FP32 instruction issues over 2 cycles, "
                                "guide: 'v_fma_f32 (wave64) 2 cyc'; transcendentals 4) at the measured clock; ..._at_measured_rate: the same "
                                "instructions at the 84.7 lane-instructions per clock and CU an FMA microbenchmark sustains on this part "
                                "(tools/ubench_f32.hip, profiles/r02_f32_packed_math.log)",
                        "source": f"profiles/{os.path.basename(sq5)} (static; scripts/clock_config5.sh)"}
            except Exception:                                       # noqa: BLE001 — an optional annotation
                pass
        # what was just timed, checked: a strided sample of instances through the Float64 call (ort_spot_batch_f64) — survivor
        # counts within 2e-4, RMS within 1e-3 (Float32 hits are good to a few 1e-5 mm on ~0.02 mm spots); a bundle's result
        # does not depend on the batch it is in (tests/test_gpu_parity.py::test_config5_full_size_properties)
        try:
            sel = np.arange(0, ninst, max(1, ninst // 200))
            r64 = batch.spot_batch(tuple(c[sel] for c in cols5), workloads.DG_A, workloads.DG_H, (0.0, 1.0), k5, engine=eng)
            dc = float(np.abs(r5["count"][sel] / r64["count"] - 1.0).max())
            dr = float(np.abs(r5["rms"][sel] / r64["rms"] - 1.0).max())
            ok5 = bool(dc <= 2e-4 and dr <= 1e-3 and np.isfinite(r5["rms"]).all() and (r5["count"] > 0).all())
            extra["config5_spot_batch_f32"]["verify"] = {
                "verified": ok5, "instances_sampled": int(sel.size), "count_max_rel_deviation": dc, "rms_max_rel_deviation": dr,
                "bar": "counts within 2e-4, RMS within 1e-3 of the Float64 call; every RMS finite, every count > 0",
                "checker": "ort_spot_batch_f64 on every %dth instance (itself checked against the CPU oracle by the GPU tests)" % max(1, ninst // 200)}
            extra["config5_spot_batch_f32"]["verified"] = ok5
        except Exception as exc:                                    # noqa: BLE001 — reported as not verified, never hidden
            extra["config5_spot_batch_f32"]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
            extra["config5_spot_batch_f32"]["verified"] = False

        # BASELINE config 4 on this one GPU: the zoom sweep's summary trace into the packed hit slab (what each rank of the
        # N > 1 exchange leg runs on its slab; nothing to gather at N = 1)
        mats4 = np.array([workloads.double_gauss(line, -1.5 + 3.0 * z / max(1, args.zoom - 1))
                          for z in range(args.zoom) for line in (0, 1, 2, 1, 2)])
        f4 = (0.0, 0.5, 0.7, 0.85, 1.0)
        plan = batch.ImageHitsPlan(mats4, workloads.DG_A, workloads.DG_H, f4, args.pupil4, engine=eng, dtype=np.float64)
        def time4(hh):                                               # (ms per sweep into this hit slab)
            plan.trace(hh); eng.ctx.synchronize()
            t0_ = time.perf_counter()
            for _ in range(6):
                plan.trace(hh)
            eng.ctx.synchronize()
            return (time.perf_counter() - t0_) / 6 * 1e3
        from opticalraytracing_jl_amd.placement import best_placed as _bp4
        warm4 = plan.new_hits(); time4(warm4); time4(warm4); del warm4   # settled clocks for the comparison
        h4, place4 = _bp4(plan.new_hits, time4, max(1, min(4, args.placement_candidates)))   # the 3.4-GB hit slab where it is written fastest
        torch.cuda.empty_cache()
        plan.trace(h4); eng.ctx.synchronize()
        t_end = time.perf_counter() + 0.25                           # sustained clocks first (timed_launches, preroll_s)
        while time.perf_counter() < t_end:
            for _ in range(8):
                plan.trace(h4)
            eng.ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            plan.trace(h4)
        eng.ctx.synchronize()
        t4 = (time.perf_counter() - t0) / 10
        rays4 = mats4.shape[0] * len(f4) * args.pupil4 * args.pupil4
        extra["config4_single_gpu"] = {
            "workload": f"BASELINE config 4 on one GPU: {args.zoom} zoom positions x 5 index columns x 5 fields x {args.pupil4}^2 pupil, "
                        "Float64, summary trace into the packed [2][n] image-plane hit slab (16 B per ray)",
            "rays": rays4, "intersections": rays4 * S, "ms_per_sweep": t4 * 1e3, "value": rays4 * S / t4, "bound": "FP64 VALU",
            "output_placement": place4,
            "finite_fraction": float(torch.isfinite(h4[0]).float().mean().item())}
        # what was just timed, checked: a strided sample of the hit slab against the CPU oracle tracing the same rays (pupil
        # coordinates from the plan's own axes, the aimed field angle of each bundle)
        try:
            from oracle.cpu import OracleEngine
            kk = args.pupil4
            idx = np.unique(np.concatenate([np.arange(0, rays4, 50021), [rays4 - 1]]))
            ti = torch.from_numpy(idx).to(dev)
            gx4, gy4 = h4[0][ti].cpu().numpy(), h4[1][ti].cpu().numpy()
            ax4 = plan.d_axes.cpu().numpy()
            b4i = idx // (kk * kk); r4 = idx - b4i * kk * kk
            yy4 = ax4[b4i * 2 * kk + r4 // kk]; xx4 = ax4[b4i * 2 * kk + kk + r4 % kk]
            orc4 = OracleEngine(nthreads=4)
            worst4, pat4 = 0.0, 0
            for bi in np.unique(b4i):
                m = b4i == bi
                ox, oy = orc4.skew(plan.ext, yy4[m], xx4[m], np.full(int(m.sum()), math.tan(plan.aim_U[bi])), np.zeros(int(m.sum())),
                                   isys=int(plan.inst[bi]), slopes=True)
                for g, o in ((gx4[m], ox[-1]), (gy4[m], oy[-1])):
                    pat4 += int((np.isnan(g) != np.isnan(o)).sum())
                    d = np.abs(g - o) / np.maximum(1.0, np.abs(o)); d = d[np.isfinite(d)]
                    worst4 = max(worst4, float(d.max()) if d.size else 0.0)
            ok4 = bool(pat4 == 0 and worst4 <= 1e-10)
            extra["config4_single_gpu"]["verify"] = {"verified": ok4, "sample_rays": int(idx.size), "max_rel_deviation": worst4,
                                                     "nan_pattern_mismatches": pat4, "bar": "<= 1e-10 relative, NaN patterns identical",
                                                     "checker": "oracle/ort_oracle.c (CPU) on every 50021st ray of the hit slab"}
            extra["config4_single_gpu"]["verified"] = ok4
        except Exception as exc:                                    # noqa: BLE001 — reported as not verified, never hidden
            extra["config4_single_gpu"]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
            extra["config4_single_gpu"]["verified"] = False
        del h4, plan
        torch.cuda.empty_cache()
        # BASELINE config 1: the reference's own call, full_trace(solve(Cooke triplet), H, 64) (test/runtests.jl:19-35), ONE C
        # call with the error vectors back in host memory; wall time seen from Python (ctypes marshalling included — the plain-C
        # caller of examples/cooke_full_trace.c is ~25 us lower: profiles/rNN_config1_c_abi_wall.log)
        c1 = {}
        for H in (0.0, 1.0):
            fn = lambda: batch.full_trace_systems(workloads.COOKE[None], workloads.COOKE_A, workloads.COOKE_H, (H,), 64, engine=eng)
            for _ in range(5):
                r1 = fn()
            ts = []
            for _ in range(100):
                t0 = time.perf_counter(); r1 = fn(); ts.append(time.perf_counter() - t0)
            c1[f"H={H:g}"] = {"wall_us_median": float(np.median(ts)) * 1e6, "wall_us_min": float(np.min(ts)) * 1e6,
                              "rays_kept": int(r1[1][0]["count"]), "rms": float(r1[1][0]["rms"])}
        extra["config1_reference_call"] = {
            "workload": "BASELINE config 1: Cooke triplet (test/runtests.jl:19-35), full_trace(system, H, 64): first-order solve, "
                        "aiming, 64 x 32 half-pupil trace, stop filter + compaction + mirror, RMS — ort_full_trace_batch_f64, host "
                        "arrays in, RealRayError vectors out", "rays_traced": 2048, "calls": c1, "bound": "latency (3 dependent launches)"}
        # what was just timed, checked: survivor count and RMS of both calls against the whole per-call route through the CPU
        # oracle (solve -> aiming -> grid trace -> filter -> mirror -> sigma); the aiming loops stop at sqrt(eps): RMS to 1e-6
        try:
            from oracle.cpu import OracleEngine
            orc1 = OracleEngine()
            s1 = api.solve(workloads.COOKE.copy(), workloads.COOKE_A, workloads.COOKE_H, engine=orc1)
            v1 = {}
            for H in (0.0, 1.0):
                e1 = api.full_trace(s1, H, 64, engine=orc1)
                got = c1[f"H={H:g}"]
                v1[f"H={H:g}"] = {"rays_kept_oracle": int(len(e1.x)), "rms_oracle": float(e1.RMS),
                                  "rms_rel_deviation": abs(got["rms"] - float(e1.RMS)) / float(e1.RMS)}
            ok1 = all(c1[kk]["rays_kept"] == v["rays_kept_oracle"] and v["rms_rel_deviation"] <= 1e-6 for kk, v in v1.items())
            extra["config1_reference_call"]["verify"] = {"verified": bool(ok1), "calls": v1, "bar": "survivor count identical, RMS within 1e-6 relative",
                                                         "checker": "the same full_trace(solve(...), H, 64) through oracle/ort_oracle.c (CPU)"}
            extra["config1_reference_call"]["verified"] = bool(ok1)
        except Exception as exc:                                    # noqa: BLE001 — reported as not verified, never hidden
            extra["config1_reference_call"]["verify"] = {"verified": False, "error": f"{type(exc).__name__}: {exc}"}
            extra["config1_reference_call"]["verified"] = False
        return extra

    extra = {}
    if not args.no_extras and args.mode == "history" and world == 1:
        try:
            extra = run_extras()
        except Exception as exc:                                # noqa: BLE001 — the headline line must still be printed
            extra = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()

    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.mode}_k{k}_{args.policy}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = (f"profiles/pmc_traffic.json[{key}] (static: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of "
                                  f"{tj.get(key, {}).get('source', 'an earlier run of this command')}, not measured in this run)")
        except Exception:
            traffic = None
    res = {
        "metric": METRIC, "value": value, "unit": "ray-surface intersections/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Double-Gauss 10 spherical surfaces + stop + image (S={S}), 3 fields x 3 index "
                               f"columns, {k}x{k} pupil, Float64, {args.mode} output, {args.policy} arithmetic policy",
                   "rays_per_step_per_gpu": N, "intersections_per_step_per_gpu": inter,
                   "surface_table": "lds",
                   "policy": args.policy,
                   "policy_parity": ("bit-identical to the CPU oracle on every ray (reference operation sequence)" if not fast else
                                     "<= 1e-10 relative vs the reference sequence; no status flip observed on any ray of the parity "
                                     "suites and soaks: a wave holding a ray within 1e-9 (normalised) of a miss / TIR / equator / "
                                     "stop-edge branch, a totally reflected ray, a far-cap hit, a backward direction or a polynomial row "
                                     "outside its conic retraces with the reference sequence (bit-identical there); "
                                     "tests/test_gpu_parity.py::_fast_attribution, tests/test_device_emulation.py"),
                   "parallelism": "1 GPU" if world == 1 else
                                  (f"{world} ranks (one process per GPU), WEAK scaling: every rank traces its own zoom position of the same "
                                   "batch — the path shards over independent (system, field, index column) units, no data-path collective; "
                                   "barrier + synchronize around the K steps, slowest rank counts"),
                   "device": info["name"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": f"ort::k_trace<double, {1 if fast else 0}, ...>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "frac_of_measured_copy_rate": achieved / HBM_MEASURED_COPY_GBS},
    }
    if args.layout_ceiling and args.mode == "history":
        c = args.layout_ceiling
        res["roofline"]["layout_store_ceiling"] = {
            "GBps": c["GBps"], "kernel_ms": c["ms"], "runs_GBps": c.get("runs_GBps"),
            "what": c["kernel"] + f": the best-placed of six pairs of arrays allocated one after the other, {c['seconds']} s sustained each "
            "(runs_GBps: all six), before the bench (tools/store_ceiling.hip, a child process).  A reference rate, not a bound: pure "
            "back-to-back stores into the best place that process found; the trace kernel runs into the best of ITS six candidate "
            "pairs (output_placement)",
            "frac_of_it": achieved / c["GBps"],
            "sustained_frac_of_it": None if sustained is None else sustained["achieved_GBps"] / c["GBps"]}
    if placement is not None and placement.get("candidates_ms"):
        res["roofline"]["output_placement"] = placement
    if verify is not None:
        res["verify"] = verify
        res["verified"] = verify["verified"]
    if sustained is not None:
        res["sustained"] = sustained
    if other is not None:
        res["other_policy"] = other
    if extra:
        res["extra"] = extra
    if world > 1:
        res["ms_per_step_per_rank"] = per_rank
        res["nranks_seen"] = dist.get_world_size()
        # what ONE rank makes of this same workload in this same run (its own K steps): value / that = the scaling over N
        res["single_rank_same_workload_value"] = inter / (per_rank[0] * 1e-3)
        res["roofline"]["note"] = "per GPU: rank 0's kernel, hipEvents on its launch stream"
        res["backend"] = backend
    if not args.no_cpu_baseline and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline(api, pres, bundles, axes, k)
            res["cpu_baseline"]["gpu_over_cpu_1core"] = value / res["cpu_baseline"]["value"]
        except Exception as exc:                                # noqa: BLE001
            res["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    if emit and rank == 0:
        print(json.dumps(res), flush=True)
    del xv, yv
    torch.cuda.empty_cache()
    return res if rank == 0 else None


def bench_multi(args, torch, rank, world, local_rank, emit=True, headline=None):
    """BASELINE config 4, strong scaling over the ranks: the sweep's bundles split into rank-ordered slabs + ONE all-gather
    of the image-plane hits per step (see the module docstring).  Returns the result dict on rank 0 (printed when `emit`).
    `headline`: the already measured weak-scaling line this leg is an extra of (printed with the error if the
    communicator cannot be built)."""
    import numpy as np
    import opticalraytracing_jl_amd as ort
    from opticalraytracing_jl_amd import batch, dist as odist, workloads

    ndev = torch.cuda.device_count()
    backend = os.environ.get("ORT_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of N > 1 on a 1-GPU box
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: one process per GPU")
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = odist.init_process_group(backend) if world > 1 else None
    fast = args.policy == "fast"
    eng = ort.HipEngine(local_dev, fast_math=fast)
    info = eng.ctx.device_info()

    c5 = args.workload == "config5"
    if c5:        # BASELINE config 5's hit payload: perturbed instances x 256^2 FULL pupil, Float32 hits (8 B per ray)
        mats = workloads.config5(None, ninst=args.instances5)
        fields, k, hdt, tdt = (0.0,), 256, np.float32, torch.float32
        wl = (f"BASELINE config 5: {args.instances5} perturbed Double-Gauss instances (seed 12345; sigma_R/R 1e-3, sigma_t 10 um, sigma_n 1e-4) x "
              f"{k}x{k} pupil on axis, Float32 trace and hits")
    else:
        mats = np.array([workloads.double_gauss(line, -1.5 + 3.0 * z / max(1, args.zoom - 1))
                         for z in range(args.zoom) for line in (0, 1, 2, 1, 2)])
        fields, k, hdt, tdt = (0.0, 0.5, 0.7, 0.85, 1.0), args.pupil4, np.float64, torch.float64
        wl = (f"BASELINE config 4: zoom-lens sweep, {args.zoom} positions x 5 index columns x 5 fields x {k}x{k} pupil "
              f"(Double-Gauss), Float64")
    esz = 4 if c5 else 8
    nb_total = mats.shape[0] * len(fields)
    S = mats.shape[1]                                          # extended system: rows + 1 rows -> rows iterations
    unit = "bundle" if nb_total % world == 0 else "row"        # rows: slabs within one pupil row of equal for any world
    plan = batch.ImageHitsPlan(mats, workloads.DG_A, workloads.DG_H, fields, k, engine=eng, shard=(rank, world), unit=unit, dtype=hdt)
    per = k if unit == "row" else 1
    bounds = odist.shard_bounds(nb_total * per, world)
    rays_of = [(hi - lo) * (k if unit == "row" else k * k) for lo, hi in bounds]
    slab = max(rays_of)                                        # equal message size: short slabs are padded at their end
    total_rays = sum(rays_of)
    inter_total = total_rays * S

    # ---- the communicator: native RCCL through the C ABI (ort_comm_*); gloo rehearsal: torch on CPU tensors ----
    native = backend == "nccl" and os.environ.get("ORT_BENCH_GATHER", "native") == "native"
    comm, native_note = None, None
    if native:
        # ort_comm_* (the C ABI's own RCCL communicator).  If it cannot be set up on this node, every rank falls back —
        # together — to torch.distributed's all_gather_into_tensor on the same device buffers (RCCL as well, one collective)
        try:
            box = [odist.RcclComm.unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(box, src=0)
            # ncclCommInitRank is a rendezvous of all ranks: run it under a watchdog so that a rank that cannot reach its
            # peers falls back (with everybody else, below) instead of hanging the whole job
            import threading
            made = {}

            def make():
                try:
                    made["comm"] = odist.RcclComm(eng, world, rank, box[0])
                except Exception as exc:                        # noqa: BLE001
                    made["err"] = exc
            th = threading.Thread(target=make, daemon=True)
            th.start()
            th.join(float(os.environ.get("ORT_BENCH_COMM_TIMEOUT_S", "180")))
            if th.is_alive():
                # ncclCommInitRank did not return: a thread of this process is still inside RCCL on this GPU.  Nothing may
                # run beside a half-built communicator — report and leave (every rank times out the same way)
                err = ("RCCL rendezvous timeout: ort_comm_create (ncclCommInitRank) did not return within "
                       f"{os.environ.get('ORT_BENCH_COMM_TIMEOUT_S', '180')} s")
                if rank == 0 and headline is not None:          # the weak-scaling line stands; this leg is reported as failed
                    headline.setdefault("extra", {})["config4_allgather"] = {"error": err, "nranks_seen": None}
                    surface_exchange_leg(headline, {"error": err})
                    print(json.dumps(headline), flush=True)
                elif rank == 0:
                    print(json.dumps({"metric": METRIC, "value": None, "unit": "ray-surface intersections/s", "n_gpus": world,
                                      "error": err, "nranks_seen": None}), flush=True)
                sys.stdout.flush()
                sys.stderr.write("bench.py: " + err + "\n"); sys.stderr.flush()
                # a failed exchange leg is a failed run, headline in hand or not: distinct non-zero codes, straight out of the
                # process (a thread is still inside RCCL: no interpreter shutdown, and never an exec)
                os._exit(EXIT_EXCHANGE_TIMEOUT_WITH_HEADLINE if headline is not None else EXIT_EXCHANGE_TIMEOUT)
            if "err" in made:
                raise made["err"]
            comm = made["comm"]
            ok = 1
        except Exception as exc:                                # noqa: BLE001 — reported in the JSON, not swallowed
            ok, native_note = 0, f"{type(exc).__name__}: {exc}"
        if dist is not None:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            if comm is not None:
                comm.close()
            comm, native = None, False
    hits = [torch.zeros((2, slab), dtype=tdt, device=dev) for _ in range(2)]
    gathered = [torch.empty((world, 2, slab), dtype=tdt, device=dev) for _ in range(2)]

    def gather(b, wait):
        if native:
            comm.allgather_hits_packed(hits[b], gathered[b], wait=wait)
        elif backend == "nccl":                                 # torch's RCCL communicator, device buffers, engine stream order
            eng.ctx.synchronize()
            if dist is not None:
                dist.all_gather_into_tensor(gathered[b].view(world * 2, slab), hits[b])
            else:
                gathered[b][0].copy_(hits[b])
        else:
            eng.ctx.synchronize()
            g = torch.empty((world, 2, slab), dtype=tdt)
            if dist is not None:
                dist.all_gather_into_tensor(g.view(world * 2, slab), hits[b].cpu())
            else:
                g[0] = hits[b].cpu()
            gathered[b].copy_(g)

    def sync_all():
        eng.ctx.synchronize()
        if comm is not None:
            comm.synchronize()
        torch.cuda.synchronize(dev)
        if dist is not None:
            odist.barrier(local_dev)

    def run(steps, with_gather):
        sync_all()
        t0 = time.perf_counter()
        for s in range(steps):
            b = s & 1
            if with_gather and native and s >= 2:
                comm.wait_lag(1)                                # hits[b] was last read by the gather of step s - 2
            elif with_gather and backend == "nccl" and s >= 2:
                torch.cuda.synchronize(dev)                     # fallback path: torch's collective stream is not ours to order
            plan.trace(hits[b])
            if with_gather:
                gather(b, wait=False)
        sync_all()
        return time.perf_counter() - t0

    run(args.warmup, True)
    wall = run(args.steps, True)                                # the timed region: trace + all-gather, overlapped
    wall_trace = run(args.steps, False)                         # gather-exclusive
    # the all-gather alone
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        gather(s & 1, wait=False)
    sync_all()
    wall_gather = time.perf_counter() - t0
    tw = torch.tensor([wall, wall_trace, wall_gather], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall, wall_trace, wall_gather = (float(v) for v in tw)

    # ---- checks: the gathered slab of THIS rank is its own hits; rank 0 re-traces the whole sweep alone (N = 1 same workload) ----
    last = (args.steps - 1) & 1
    own_ok = bool(torch.equal(torch.nan_to_num(gathered[last][rank]), torch.nan_to_num(hits[last])))
    ref = None
    verified = own_ok
    if rank == 0 and not args.no_verify:
        whole = batch.ImageHitsPlan(mats, workloads.DG_A, workloads.DG_H, fields, k, engine=eng, dtype=hdt)
        wh = whole.new_hits()
        whole.trace(wh); eng.ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            whole.trace(wh)
        eng.ctx.synchronize()
        t1 = (time.perf_counter() - t0) / 3
        ref = {"n_gpus": 1, "value": inter_total / t1, "ms_per_step": t1 * 1e3,
               "note": f"rank 0 alone, the same {nb_total} bundles, summary trace only (nothing to gather at N = 1), measured after the timed region"}
        g = gathered[last]
        same = True
        off = 0
        for r in range(world):
            n_r = rays_of[r]
            same = same and bool(torch.equal(torch.nan_to_num(g[r, :, :n_r]), torch.nan_to_num(wh[:, off:off + n_r])))
            off += n_r
        verified = own_ok and same
        del wh
    # ---- the same sweep when only spot statistics are wanted ("prefer reducing on device"): every rank runs
    # ONE C call over its instances (solve, aiming, axes, reference-mode half-pupil trace, statistics on the device) and
    # the ranks exchange 16 B per bundle — no ray-sized collective, so this one does divide by N
    stats = None
    ninst = mats.shape[0]
    if not args.no_extras and ninst % world == 0:
        lo, hi = rank * (ninst // world), (rank + 1) * (ninst // world)
        sub = mats[lo:hi]
        nloc = (hi - lo) * len(fields)
        gath = torch.empty((world, 2, nloc), dtype=torch.float64, device=dev if backend == "nccl" else "cpu")

        def stats_step():
            r = batch.spot_batch(sub, workloads.DG_A, workloads.DG_H, fields, k, engine=eng, dtype=hdt)
            mine = torch.from_numpy(np.stack([r["count"].reshape(-1).astype(np.float64), r["rms"].reshape(-1)]))
            mine = mine.to(gath.device)
            if dist is not None:
                dist.all_gather_into_tensor(gath.view(world * 2, nloc), mine)
            else:
                gath[0].copy_(mine)
            return r

        stats_step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stats_step()
        sync_all()
        ts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        rays_stats = ninst * len(fields) * k * (k // 2)
        stats = {"what": f"spot statistics of the same sweep: ort_spot_batch_{'f32' if c5 else 'f64'} per rank over its {hi - lo} "
                         f"instances x {len(fields)} fields x {k}x{k // 2} rays (reference mode: half pupil + mirror), first-order solve "
                         "and aiming included, then ONE all-gather of (count, RMS) per bundle",
                 "ms_per_step": float(ts[0]) / args.steps * 1e3, "intersections_per_step": rays_stats * S,
                 "value": rays_stats * S * args.steps / float(ts[0]), "unit": "ray-surface intersections/s",
                 "collective_bytes_per_rank": 16 * nloc, "scaling": "strong",
                 "finite": bool(torch.isfinite(gath[:, 1]).all()), "kept_fraction": float(gath[:, 0].sum()) / rays_stats / 2.0}
    if dist is not None:
        odist.barrier(local_dev)
    res = None
    if rank == 0:
        msg_bytes = 2.0 * esz * slab
        res = {
            "metric": METRIC, "value": inter_total * args.steps / wall, "unit": "ray-surface intersections/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if c5 else "f64", "data": "synthetic",
            "config": {"workload": wl + f" (S={S}), summary trace of each rank's slab + ONE all-gather of the image-plane hits per step "
                                        f"(inside the timed region, overlapped with the next step's trace), {args.policy} arithmetic policy",
                       "rays_per_step": total_rays, "intersections_per_step": inter_total,
                       "shard_unit": unit, "rays_per_rank": rays_of, "policy": args.policy,
                       "parallelism": f"{world} ranks x contiguous rank-ordered slabs, no data-path collective, 1 reassembly all-gather",
                       "device": info["name"]},
            "allgather": {"impl": f"ort_allgather_hits_packed_{'f32' if c5 else 'f64'} (native RCCL, one ncclAllGather on the communicator's stream)" if native
                                  else ("torch.distributed all_gather_into_tensor (RCCL, device buffers)" if backend == "nccl"
                                        else "torch.distributed gloo on CPU tensors (rehearsal)"),
                          "native_fallback_reason": native_note,
                          "nranks_seen": comm.nranks_seen if comm is not None else (dist.get_world_size() if dist is not None else 1),
                          "message_bytes_per_rank": msg_bytes, "bytes_assembled_per_rank": msg_bytes * world,
                          "alone_ms": wall_gather / args.steps * 1e3,
                          "alone_GBps_per_rank": msg_bytes * world / (wall_gather / args.steps) / 1e9},
            "gather_exclusive": {"ms_per_step": wall_trace / args.steps * 1e3, "value": inter_total * args.steps / wall_trace},
            "overlap": {"trace_plus_gather_serial_ms": (wall_trace + wall_gather) / args.steps * 1e3,
                        "measured_ms": wall / args.steps * 1e3},
            "roofline": {"bound": "valu", "note": f"summary trace writes {2 * esz} B per ray ({2 * esz / S:.2f} B per intersection): {'FP32' if c5 else 'FP64'} VALU-bound, see the "
                                                  "N = 1 line for the HBM-bound history kernel", "achieved": 2.0 * esz * total_rays / world / (wall_trace / args.steps) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 2.0 * esz * total_rays / world / (wall_trace / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic": None},
            "verified": verified,
            "verify": {"own_slab_in_gathered": own_ok, "gathered_equals_single_gpu_trace": None if ref is None else verified},
            # for a curve over N drawn from ONE workload: the ranks the collective really spanned, and what rank 0 alone makes
            # of this same sweep (the N = 1 bench line is BASELINE config 2, another workload)
            "nranks_seen": comm.nranks_seen if comm is not None else (dist.get_world_size() if dist is not None else 1),
            "single_rank_same_workload_value": None if ref is None else ref["value"],
        }
        if ref is not None:
            res["strong_scaling_reference"] = ref
        if stats is not None:
            res["extra"] = {"spot_statistics": stats}
        if emit:
            print(json.dumps(res), flush=True)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    return res


EXIT_EXCHANGE_TIMEOUT = 3                   # the exchange leg was the workload and its RCCL rendezvous timed out
EXIT_EXCHANGE_TIMEOUT_WITH_HEADLINE = 5     # ... it was the extra leg of the N > 1 line: the headline was printed, the run still failed
EXIT_EXCHANGE_FAILED = 6                    # the extra leg raised or did not verify: line printed with exchange_leg_ok false
EXIT_RANKS_TIMEOUT = 7                      # spawn_ranks: the ranks outlived their deadline and were killed


def surface_exchange_leg(res: dict, leg: dict) -> None:
    """Copies of the exchange leg's figures (extra.config4_allgather) at the TOP level of the N > 1 line, where a reader that
    keeps only top-level keys still sees them: the one number of this bench that tests RCCL over xGMI."""
    ag = leg.get("allgather", {}) if isinstance(leg, dict) else {}
    res["exchange_value"] = leg.get("value")
    res["exchange_unit"] = "ray-surface intersections/s (BASELINE config 4, strong scaling, all-gather inside the timed region)"
    res["exchange_ms_per_step"] = leg.get("ms_per_step")
    res["allgather_impl"] = ag.get("impl")
    res["allgather_GBps_per_rank"] = ag.get("alone_GBps_per_rank")
    res["allgather_alone_ms"] = ag.get("alone_ms")
    res["exchange_nranks_seen"] = leg.get("nranks_seen")
    res["exchange_verified"] = bool(leg.get("verified")) and "error" not in leg
    res["exchange_leg_ok"] = res["exchange_verified"]
    if "error" in leg:
        res["exchange_error"] = leg["error"]


def spawn_ranks(n: int, argv, timeout_s: float = 3300.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks as ONE child process tree
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU) BEFORE
    this process imports torch or touches the GPU, relay the children's output, and return (return code, rank 0's JSON
    line or None).  Never exec: the parent stays a plain process and exits with the children's code."""
    import socket
    import subprocess
    with socket.socket() as sk:                                  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL across processes needs it on this host driver
    env["ORT_BENCH_SPAWNED"] = "1"
    import signal
    import threading
    timeout_s = float(os.environ.get("ORT_BENCH_RANKS_TIMEOUT_S", timeout_s))
    line = [None]
    # its own session: on expiry the whole tree (launcher + ranks) is killed by process group, never by name
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)

    def relay():                                                 # relay as it comes: a long run keeps printing
        for out in proc.stdout:
            if out.startswith("{") and line[0] is None:
                line[0] = out.strip()
            sys.stdout.write(out); sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    try:
        rc = proc.wait(timeout=timeout_s)                        # the deadline covers a rank stuck inside a collective:
    except subprocess.TimeoutExpired:                            # its pipe never reaches EOF, so the reader must not be what waits
        sys.stderr.write(f"bench.py: the ranks did not finish within {timeout_s:.0f} s: killing their process group\n")
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = EXIT_RANKS_TIMEOUT
    except BaseException:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        raise
    th.join(timeout=10)
    return rc, line[0]


def selftest_ranks():
    """What a spawned rank does under --selftest-ranks: rendezvous over gloo on the CPU (no GPU needed), agree on the
    world size, rank 0 prints one JSON line.  tests/test_dist_gloo.py drives the spawn path through this."""
    import torch
    import torch.distributed as dist
    from opticalraytracing_jl_amd import dist as odist
    rank, world, _ = odist.env_rank_world()
    odist.init_process_group("gloo")
    if os.environ.get("ORT_BENCH_SELFTEST_HANG") == str(rank):   # the launcher's deadline test: this rank never reaches the collective
        time.sleep(3600)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"selftest": True, "world": world, "rank_sum": int(t[0]), "spawned": os.environ.get("ORT_BENCH_SPAWNED") == "1"}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pupil", type=int, default=1024, help="N = 1: pupil grid edge per bundle (config 2: 1024)")
    ap.add_argument("--pupil3", type=int, default=2048, help="N = 1 extras: pupil edge of the config-3 full_trace run")
    ap.add_argument("--instances5", type=int, default=10000, help="N = 1 extras: instances of the config-5 Monte-Carlo run")
    ap.add_argument("--pupil4", type=int, default=512, help="N > 1: pupil edge of the zoom sweep (config 4: 512)")
    ap.add_argument("--zoom", type=int, default=32, help="N > 1: zoom positions (config 4: 32)")
    ap.add_argument("--policy", default="fast", choices=["fast", "ieee"],
                    help="arithmetic policy of the timed kernel: fast = direction-cosine form (<= 1e-10 relative, status exact; "
                         "anomalous waves retrace with the reference sequence); ieee = the reference's IEEE operation sequence "
                         "(bit-identical to the CPU oracle).  The other policy is timed too and reported beside it.")
    ap.add_argument("--fast-math", action="store_true", help="alias of --policy fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the summary / config-3 extras")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed output")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the layout store-ceiling measurement (tools/store_ceiling)")
    ap.add_argument("--sustain-s", type=float, default=1.0, help="seconds of back-to-back launches for the sustained figure (0 = off)")
    ap.add_argument("--mode", default="history", choices=["history", "summary", "full_trace"])
    ap.add_argument("--placement-candidates", type=int, default=16,
                    help="history mode: candidate pairs of output arrays to choose the best-placed from (1 = take the first allocation)")
    ap.add_argument("--ft-lookback", action="store_true", help="--mode full_trace: the ORT_FT_LOOKBACK route")
    ap.add_argument("--ft-fused", action="store_true", help="--mode full_trace: the ORT_FT_FUSED route")
    ap.add_argument("--workload", default="auto", choices=["auto", "config2", "config4", "config5"],
                    help="auto: config 2 at N = 1, config 4 (sharded + all-gather) at N > 1; config5: the Float32 hit payload of the "
                         "Monte-Carlo run, sharded + all-gathered the same way")
    ap.add_argument("--selftest-ranks", action="store_true", help="spawned ranks only rendezvous (gloo, CPU) and report: the launcher's own test")
    args = ap.parse_args()
    if args.fast_math:
        args.policy = "fast"

    from opticalraytracing_jl_amd import dist as odist
    rank, world, local_rank = odist.env_rank_world()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called the way the driver calls --gpus 1: launch the ranks ourselves, before anything here touches the GPU
        rc, line = spawn_ranks(args.gpus, sys.argv[1:])
        if line is None and rc == 0:
            rc = 4
            sys.stderr.write("bench.py: the ranks exited without a JSON line\n")
        raise SystemExit(rc)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.selftest_ranks:
        selftest_ranks()
        return
    # N = 1: what store rate does the history LAYOUT itself sustain on this box (tools/store_ceiling: the same launch
    # shape and stores, no ray tracing)?  Run as a child process BEFORE this process touches the GPU.
    args.layout_ceiling = None
    exe = os.path.join(ROOT, "tools", "store_ceiling")
    if world == 1 and args.workload in ("auto", "config2") and not args.no_ceiling and os.path.exists(exe):
        try:
            import subprocess
            runs = []
            for _ in range(1):                                  # (its first half second brings the box out of idle clocks)
                out = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=60).stdout
                runs.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
            args.layout_ceiling = max(runs, key=lambda r: r["GBps"])
            args.layout_ceiling["runs_GBps"] = args.layout_ceiling.get("candidates_GBps", [r["GBps"] for r in runs])
        except Exception:                                       # noqa: BLE001 — a missing helper only drops the extra field
            args.layout_ceiling = None
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.workload in ("config4", "config5"):
        bench_multi(args, torch, rank, world, local_rank)
    elif world > 1 and args.workload == "auto":
        # N > 1: the headline workload under weak scaling, then the exchange step (config 4 + all-gather) as an extra of the line
        res = bench_single(args, torch, rank, world, local_rank, emit=False)
        leg = None
        if not args.no_extras:
            try:
                leg = bench_multi(args, torch, rank, world, local_rank, emit=False, headline=res)
            except Exception as exc:                            # noqa: BLE001 — the headline line must still be printed
                leg = {"error": f"{type(exc).__name__}: {exc}"}
        failed = False
        if rank == 0:
            if leg is not None:
                res.setdefault("extra", {})["config4_allgather"] = leg
                surface_exchange_leg(res, leg)
                failed = not res["exchange_leg_ok"]
            print(json.dumps(res), flush=True)
        if failed:                                              # the line is out; a failed or unverified exchange leg fails the run
            sys.stdout.flush()
            raise SystemExit(EXIT_EXCHANGE_FAILED)
    else:
        bench_single(args, torch, rank, world, local_rank)


if __name__ == "__main__":
    main()
