#!/usr/bin/env python3
"""Generates tests/golden/*.json from the 50-digit mpmath restatement (oracle/mp_model.py).

The reference is Julia source and cannot run in the build container (no Julia runtime), so
these vectors are restatement-derived — they pin the C oracle against an independent,
arbitrary-precision reading of the same reference lines; the reference's own known answers are
checked separately (tests/test_oracle_reference_vectors.py).
Prescriptions are the reference's test data (test/runtests.jl:19-35, :335-338, :377-382;
docs/setup.jl:4-15) plus the authored Double-Gauss.   Run:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mp_model as mpm          # noqa: E402
from tests import common as cm              # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def f(v):
    return [float(a) for a in v]


def ext(M, focus):
    e = np.vstack([M, [math.inf, 0.0, 1.0]])
    e[-2, 1] = focus
    return e


def jl(a):
    """JSON-safe float list (inf/nan as strings)."""
    out = []
    for v in np.asarray(a, dtype=float).ravel():
        out.append("inf" if v == math.inf else "-inf" if v == -math.inf else "nan" if v != v else float(v))
    return out


def skew_cases():
    rng = np.random.default_rng(1)
    cases = []
    systems = {
        "cooke_ext": (ext(cm.cooke(), 77.40534796682427), None, None, 14.7),
        "tessar_ext": (ext(cm.tessar(), 40.0), None, None, 9.0),
        "catadioptric": (cm.catadioptric(), None, None, 14.0),
        "double_gauss_ext": (ext(cm.double_gauss(), 57.8), None, None, 20.0),
    }
    P = cm.parabola_M()
    systems["parabola"] = (P[:, :3], P[:, 3], None, 28.0)
    M4, coef = cm.double_gauss_aspheric()
    e4 = np.vstack([M4, [math.inf, 0.0, 1.0, 0.0]]); e4[-2, 1] = 57.8
    systems["double_gauss_aspheric_ext"] = (e4[:, :3], e4[:, 3], np.vstack([coef, np.zeros((1, coef.shape[1]))]), 16.0)
    for name, (M, K, C, a1) in systems.items():
        rays = [(5.0, 3.0, 0.1, -0.05)] if name == "cooke_ext" else []
        for _ in range(12):
            rays.append((rng.uniform(-0.9 * a1, 0.9 * a1), rng.uniform(-0.9 * a1, 0.9 * a1),
                         rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15)))
        if name == "cooke_ext":
            rays += [(38.0, 0.0, 0.0, 0.0), (60.0, 0.0, 0.3, 0.0), (10.0, 10.0, 0.9, 0.9)]   # misses / steep
        out = []
        for (y, x, U, V) in rays:
            xv, yv = mpm.trace_skew(M[:, 0], M[:, 1], M[:, 2], K, None if C is None else [list(c) for c in C], y, x, U, V)
            out.append({"y": y, "x": x, "U": U, "V": V, "xv": jl(f(xv)), "yv": jl(f(yv))})
        cases.append({"system": name, "R": jl(M[:, 0]), "t": jl(M[:, 1]), "n": jl(M[:, 2]),
                      "K": None if K is None else jl(K), "coef": None if C is None else [jl(c) for c in C],
                      "rays": out})
    return cases


def meridional_cases():
    rng = np.random.default_rng(2)
    cases = []
    P = cm.parabola_M()
    for name, M, K, layout_mode, a1 in (("cooke", cm.cooke(), None, False, 12.0),
                                        ("catadioptric", cm.catadioptric(), None, False, 14.0),
                                        ("tessar_layout", cm.tessar(), None, True, 8.0),
                                        ("parabola", P[:, :3], P[:, 3], True, 28.0)):
        out = []
        for _ in range(8):
            y, U = rng.uniform(-a1, a1), (0.0 if name == "parabola" else rng.uniform(-0.15, 0.15))
            ys, Us, ts = mpm.trace_meridional(M[:, 0], M[:, 1], M[:, 2], K, None, layout_mode, y, U)
            out.append({"y": y, "U": U, "ys": jl(f(ys)), "Us": jl(f(Us)), "ts": jl(f(ts))})
        cases.append({"system": name, "R": jl(M[:, 0]), "t": jl(M[:, 1]), "n": jl(M[:, 2]),
                      "K": None if K is None else jl(K), "layout_mode": layout_mode, "rays": out})
    return cases


def paraxial_cases():
    import opticalraytracing_jl_amd as ort
    L = ort.Lens(cm.cooke())
    tau, phi = L.M[:, 0], L.M[:, 1]
    out = []
    for y, w in ((1.0, 0.0), (0.0, 1.0), (-3.5, 0.02)):
        ys, ws = mpm.trace_paraxial(tau, phi, y, w)
        out.append({"y": y, "w": w, "ys": f(ys), "ws": f(ws)})
    M = mpm.abcd(tau, phi)
    return {"tau": f(tau), "phi": f(phi), "rays": out, "abcd": [f(M[0]), f(M[1])]}


def full_trace_case():
    """The reference's singlet (test/runtests.jl:364-366), H = 0, k = 16, with aiming scalars
    produced by the host logic on the C oracle and frozen here as inputs."""
    import opticalraytracing_jl_amd as ort
    from oracle.cpu import OracleEngine
    eng = OracleEngine()
    system = ort.solve(cm.singlet(), [20.0, 20.0], 17.787, engine=eng)
    aim = ort.full_trace_aim(system.layout, system, 0.7, engine=eng)
    pres = ort.extended_prescription(system.layout, aim.focus)
    k = 16
    yax, xax = ort.linrange(aim.y1, aim.y2, k), ort.linrange(0.0, aim.y_EP, k // 2)
    ex, ey, rho, th, rms, m = mpm.full_trace_grid(pres.R[0], pres.t[0], pres.n[0], None, None, list(yax), list(xax),
                                                  aim.U, 0.0, aim.stop, aim.a_stop, aim.hprime)
    return {"R": jl(pres.R[0]), "t": jl(pres.t[0]), "n": jl(pres.n[0]), "yaxis": f(yax), "xaxis": f(xax),
            "U": aim.U, "stop": aim.stop, "a_stop": aim.a_stop, "hprime": float(aim.hprime),
            "ex": f(ex), "ey": f(ey), "rho": f(rho), "theta": f(th), "rms": float(rms), "survivors": m}


if __name__ == "__main__":
    meta = {"generator": "tests/golden/make_golden.py", "precision_digits": 50,
            "provenance": "mpmath restatement of /root/reference src/PupilSampling.jl, src/RayTracing.jl, "
                          "src/TransferMatrix.jl — NOT output of the Julia reference (no Julia runtime available)"}
    json.dump({"meta": meta, "cases": skew_cases()}, open(os.path.join(HERE, "skew_mp.json"), "w"), indent=0)
    json.dump({"meta": meta, "cases": meridional_cases()}, open(os.path.join(HERE, "meridional_mp.json"), "w"), indent=0)
    json.dump({"meta": meta, "case": paraxial_cases()}, open(os.path.join(HERE, "paraxial_mp.json"), "w"), indent=0)
    json.dump({"meta": meta, "case": full_trace_case()}, open(os.path.join(HERE, "full_trace_mp.json"), "w"), indent=0)
    print("wrote goldens to", HERE)
