#!/usr/bin/env python3
"""Where the device time of the reference's own call goes (config 1: Cooke triplet, full_trace(system, H, 64)).
Needs a library built with -DORT_PHASE_CLOCKS (block 0 of the small-problem kernels stamps the shader clock and the
100 MHz wall clock at its phase boundaries):

  hipcc ... -DORT_PHASE_CLOCKS -o build/variants/libort_phase.so opticalraytracing.jl_amd/csrc/ort_hip.hip
  python scripts/phase_clocks.py build/variants/libort_phase.so [--field 1.0] [--mode full|stats]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

INNER = {4: "last Newton round: before the trace", 9: "last Newton round: after the trace"}
NAMES = {0: "prepare: start", 1: "prepare: first-order done", 2: "prepare: tables done", 3: "aim: chief + marginal Newton done",
         5: "aim: edge search starts", 6: "aim: done", 7: "prepare: axes written", 8: "trace: start", 14: "trace: table staged", 15: "trace: surfaces done",
         10: "finish: start", 11: "finish: scan done", 12: "finish: placed", 13: "finish: end"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--field", type=float, default=1.0)
    ap.add_argument("--mode", default="full")
    a = ap.parse_args()
    os.environ["ORT_HIP_LIB"] = a.lib if os.path.isabs(a.lib) else os.path.join(ROOT, a.lib)
    import opticalraytracing_jl_amd as ort
    from opticalraytracing_jl_amd import batch
    from tests import common as cm
    eng = ort.HipEngine()
    lib = eng.ctx.lib
    fn = lib.ort_debug_phase_clocks
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    fn.restype = C.c_int
    mats = cm.cooke()[None]
    call = ((lambda: batch.full_trace_systems(mats, cm.COOKE_A, cm.COOKE_H, (a.field,), 64, engine=eng)) if a.mode == "full" else
            (lambda: batch.spot_batch(mats, cm.COOKE_A, cm.COOKE_H, (a.field,), 64, engine=eng)))
    rows = []
    for _ in range(12):
        call()
        buf = (C.c_ulonglong * 32)()
        assert fn(eng.ctx.h, buf) == 0
        rows.append(np.array(list(buf), dtype=np.float64))
    d = np.array(rows[4:])
    ks = [0, 1, 2, 3, 5, 6, 7, 8, 14, 15, 10, 11, 12, 13]
    out = {}
    for prev, cur in zip(ks[:-1], ks[1:]):
        cyc = np.median(d[:, cur] - d[:, prev]); wall = np.median(d[:, 16 + cur] - d[:, 16 + prev])
        out[f"{NAMES[prev]} -> {NAMES[cur]}"] = {"shader_cycles": float(cyc), "wall_us": float(wall) / 100.0}
    out["one aiming trace (last round of the edge search, to the stop)"] = {
        "shader_cycles": float(np.median(d[:, 9] - d[:, 4])), "wall_us": float(np.median(d[:, 16 + 9] - d[:, 16 + 4])) / 100.0}
    tot = np.median(d[:, 16 + 13] - d[:, 16 + 0]) / 100.0
    print(json.dumps({"mode": a.mode, "field": a.field, "phases": out, "first_stamp_to_last_us": float(tot)}, indent=1))


if __name__ == "__main__":
    main()
