#!/usr/bin/env python3
"""GPU box: soak of the ORT_FT_FUSED route (full_trace's second pass inside the trace launch; in-launch hand-offs by sc1 stores /
sc1 loads, no fence).  Random launches — 2..12 bundles of 33..1300 tiles, Float64 and Float32, both arithmetic policies, the two
pupil-launch rules — each run through the default route and the fused route of ONE context, every output compared bit for bit
(ex, ey, rho, theta, count, RMS); consecutive launches reuse the same workspace slots, offsets and aggregates with other contents.
The second half of the run repeats the comparison while another host thread keeps the chip loaded with the store-bound history
kernel on a context of its own (uneven load: a hand-off that only holds on an idle chip would show here).

  python scripts/soak_fused.py [seconds = 60] [seed = 1]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opticalraytracing_jl_amd as ort                     # noqa: E402
from opticalraytracing_jl_amd import api, workloads         # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    engines = {"ieee": ort.HipEngine(0), "fast": ort.HipEngine(0, fast_math=True)}
    base = ort.default_engine()
    sys_plain = [api.solve(workloads.double_gauss(line), workloads.DG_A, workloads.DG_H, engine=base) for line in (0, 1, 2)]
    sys_asph = []
    for line in (0, 1, 2):
        M4, coef = workloads.double_gauss_aspheric(line)
        lay = api.Layout(M4[:, 0], M4[:, 1], M4[:, 2], M4[:, 3], [c for c in coef])
        sys_asph.append(api.solve(lay, workloads.DG_A, workloads.DG_H, engine=base))

    stop = threading.Event()

    def loader():
        eng = ort.HipEngine(0, fast_math=True)
        pres, bundles, axes = workloads.config2(api, 512, engine=base)
        while not stop.is_set():
            eng.grid(pres, bundles[:3], axes, 512, 512)      # history + summary through host buffers: kernels, copies, gaps

    stats = dict(launches=0, loaded_launches=0, bundles=0, survivors=0, mismatches=0)
    t_end = time.time() + seconds
    t_half = time.time() + seconds / 2
    th = None
    while time.time() < t_end:
        if th is None and time.time() >= t_half:
            th = threading.Thread(target=loader, daemon=True); th.start()
        systems = sys_asph if rng.random() < 0.5 else sys_plain
        k = int(rng.integers(130, 820))                      # 33 .. 1313 tiles of 512 rays per bundle
        nf = int(rng.integers(1, 5))
        fields = tuple(float(x) for x in rng.uniform(0.0, 1.0, nf))
        nl = int(rng.integers(1, 4))
        pres, bundles, axes = workloads.square_pupil_bundles(api, systems[:nl], k, fields=fields)
        if len(bundles) < 2:
            continue
        policy = "fast" if rng.random() < 0.5 else "ieee"
        dtype = np.float32 if rng.random() < 0.3 else np.float64
        eng = engines[policy]
        a = eng.full_trace_grid(pres, bundles, axes, k, k, dtype=dtype)
        b = eng.full_trace_grid(pres, bundles, axes, k, k, dtype=dtype, fused=True)
        bad = 0
        for ra, rb in zip(a, b):
            same = ra["count"] == rb["count"] and np.float64(ra["rms"]).tobytes() == np.float64(rb["rms"]).tobytes()
            for key in ("ex", "ey", "rho", "theta"):
                same = same and ra[key].tobytes() == rb[key].tobytes()
            bad += 0 if same else 1
            stats["survivors"] += ra["count"] // 2
        stats["launches"] += 1; stats["bundles"] += len(bundles); stats["mismatches"] += bad
        stats["loaded_launches"] += 1 if th is not None else 0
        if bad:
            print(f"MISMATCH: k={k} fields={fields} lines={nl} policy={policy} dtype={np.dtype(dtype).name}: {bad} of {len(bundles)} bundles differ", flush=True)
    stop.set()
    if th is not None:
        th.join(timeout=30)
    print(f"soak_fused seed {seed}, {seconds:.0f} s: {stats['launches']} random launches ({stats['loaded_launches']} of them beside a loaded chip), "
          f"{stats['bundles']} bundles, {stats['survivors']:,} survivors placed twice: {stats['mismatches']} bundles differ between the fused and the default route")
    sys.exit(1 if stats["mismatches"] else 0)


if __name__ == "__main__":
    main()
