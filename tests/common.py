"""Prescriptions and helpers shared by the tests.

Cooke triplet, apertures, h′ and the Smith tables: /root/reference/test/runtests.jl:9-84.
Tessar: /root/reference/docs/setup.jl:4-21.  Singlet: test/runtests.jl:364-366.
Parabola: :335-338.  Catadioptric: :377-382.  These are data (prescriptions and expected
numbers), not source text.  The Double-Gauss is authored here (the reference ships
none): a 10-surface, 6-element f/3-style double Gauss with a flat stop plane.
"""
import math

import numpy as np

INF = math.inf
AIR, SK4, SF2 = 1.0, 1.61272, 1.64769


def cooke():
    return np.array([
        [INF, 0.0, AIR],
        [37.40, 5.90, SK4],
        [-341.48, 12.93, AIR],
        [-42.65, 2.50, SF2],
        [36.40, 2.00, AIR],
        [INF, 9.85, AIR],
        [204.52, 5.90, SK4],
        [-37.05, 0.0, AIR],
    ])


COOKE_A = np.array([14.7, 14.7, 10.8, 10.8, 10.3, 11.6, 11.6])
COOKE_H = 21.248
COOKE_DN = np.array([0.0, 0.010450, 0.0, 0.019151, 0.0, 0.0, 0.010450, 0.0])

# Smith, Modern Optical Engineering ch. 6 — test/runtests.jl:62-84
COOKE_YUI = np.array([
    [14.6, 0.0, 0.0],
    [14.6, -0.148315, 0.390374],
    [13.724943, -0.263817, -0.188507],
    [10.313791, -0.065055, -0.505641],
    [10.151154, 0.073436, 0.213823],
    [10.298026, 0.073436, 0.073436],
    [11.021371, 0.025062, 0.127325],
    [11.169234, -0.144296, -0.276402],
    [0.0, -0.144296, -0.144296],
])
COOKE_YUI_CHIEF = np.array([
    [0.0, 0.21, 0.21],
    [-6.411174, 0.195343, 0.038578],
    [-5.25865, 0.324469, 0.210743],
    [-1.063264, 0.187124, 0.349399],
    [-0.595454, 0.297727, 0.170765],
    [-3.3307e-15, 0.297727, 0.297727],
    [2.932611, 0.179164, 0.312066],
    [3.989677, 0.222961, 0.07148],
    [21.248022, 0.222961, 0.222961],
])


def tessar():
    return np.array([
        [INF, 0.0, 1.0],
        [16.28, 3.57, 1.6116],
        [-275.7, 1.89, 1.0],
        [-34.57, 0.81, 1.6053],
        [15.82, 2.345, 1.0],
        [INF, 0.905, 1.0],
        [INF, 2.17, 1.5123],
        [19.2, 3.96, 1.6116],
        [-24.0, 0.0, 1.0],
    ])


TESSAR_A = np.array([9.5, 9.5, 9.0, 9.0, 7.63, 8.5, 8.5, 8.5])
TESSAR_H = 21.5

NBK7 = 1.5168


def singlet():
    return np.array([[INF, 0.0, 1.0], [100.0, 10.0, NBK7], [-100.0, 0.0, 1.0]])


def parabola_M():
    return np.array([[INF, 0.0, 1.0, 0.0], [-100.0, 0.0, -1.0, -1.0]])


def catadioptric():
    return np.array([
        [INF, 0.0, 1.0],
        [-100.0, -24.0, -1.0],
        [50.0, -3.0, -1.5],
        [-50.0, 0.0, -1.0],
    ])


# ---- authored Double-Gauss: lives with the product's synthetic workloads --------------------
from opticalraytracing_jl_amd.workloads import (DG_A, DG_H, double_gauss,  # noqa: E402,F401
                                                double_gauss_aspheric)


REPORT = []          # lines the parity tests want in the run's output (tests/conftest.py prints them in the terminal summary)


def report(line: str) -> None:
    REPORT.append(line)


def rel_err(got, ref, scale):
    """|got - ref| / max(|ref|, scale): coordinates legitimately cross 0."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    both_nan = np.isnan(got) & np.isnan(ref)
    d = np.abs(got - ref) / np.maximum(np.abs(ref), scale)
    d = np.where(both_nan, 0.0, d)
    return np.where(np.isnan(d), np.inf, d)


def build_c_example(name: str = "cooke_full_trace") -> str:
    """gcc-compile examples/<name>.c against include/ort.h and the in-tree libort_hip.so; returns the binary."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from opticalraytracing_jl_amd import _capi
    _capi.load()                                              # builds the library if it is missing
    libdir = os.path.dirname(_capi.LIB_PATH)
    out = os.path.join(root, "build", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", name + ".c"),
           "-o", out, "-L" + libdir, "-lort_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link,/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out
