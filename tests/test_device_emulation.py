"""The device's per-surface step functions (csrc/ort_device.hpp) run on the HOST (tests/emu: gfx950 builtins replaced
by stand-ins whose seeds are 2^-24 accurate like the hardware's) against the CPU oracle — no GPU needed.  What this
pins without hardware: the reference-sequence arms are the oracle's operations (bit-identical), the MATH_FAST arms
(centre-form spheres, general spheres, conics, flat rows, the even-form and general polynomial arms) agree to rounding,
and the near-branch rule makes MATH_FAST status-exact: every ray whose miss / TIR / equator margin is small is flagged
`odd` (the kernel retraces it with the reference sequence), so the NaN patterns — the surface-hit index — are
identical on every ray.  The GPU suite repeats the same checks through the C ABI (tests/test_gpu_parity.py)."""
import math

import numpy as np
import pytest

from opticalraytracing_jl_amd import Prescription
from tests import common as cm
from tests import emu

TOL = 1e-10


def _ext(surfaces, focus):
    e = np.vstack([surfaces, [math.inf, 0.0, 1.0]])
    e[-2, 1] = focus
    return e


def _status(xv, yv):
    ok = ~(np.isnan(xv) | np.isnan(yv))
    return 1 + ok.cumprod(axis=0).sum(axis=0)


def _rays(n, a, seed, ang=0.2):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-a, a, n), rng.uniform(-a, a, n), np.tan(rng.uniform(-ang, ang, n)), np.tan(rng.uniform(-ang, ang, n)))


@pytest.mark.parametrize("name", ["cooke", "tessar", "catadioptric", "double_gauss"])
def test_reference_sequence_arms_are_the_oracles_operations(oracle_engine, name):
    M = {"cooke": _ext(cm.cooke(), 77.40534796682427), "tessar": _ext(cm.tessar(), 40.0),
         "catadioptric": cm.catadioptric(), "double_gauss": _ext(cm.double_gauss(), 57.8)}[name]
    pres = Prescription.from_matrix(M)
    y, x, u, v = _rays(3000, {"cooke": 14.7, "tessar": 9.5, "catadioptric": 15.0, "double_gauss": 29.0}[name], 1)
    ox, oy = oracle_engine.skew(pres, y, x, u, v, slopes=True)
    ex, ey, _ = emu.trace(pres, y, x, u, v, fast=False)
    assert np.array_equal(ex, ox, equal_nan=True) and np.array_equal(ey, oy, equal_nan=True)
    fx, fy, odd = emu.trace_fast_with_retrace(pres, y, x, u, v)
    assert np.array_equal(np.isnan(fx), np.isnan(ox)) and np.array_equal(np.isnan(fy), np.isnan(oy))
    assert max(cm.rel_err(fx, ox, 1.0).max(), cm.rel_err(fy, oy, 1.0).max()) <= 1e-11


def test_polynomial_arms_even_and_general_forms(oracle_engine):
    """BASELINE config 3's rows (conic + even polynomial: the 4-term even form), a 10th-order even asphere (6-term), odd
    coefficients (general forms of <= 8 and <= 12 coefficients) and a flat row carrying a polynomial (Schmidt-like)."""
    M4, coef = cm.double_gauss_aspheric()
    ext = np.vstack([M4, [math.inf, 0.0, 1.0, 0.0]]); ext[-2, 1] = 57.8

    def pres_with(c):
        c = np.vstack([c, np.zeros((1, c.shape[1]))])
        return Prescription(ext[:, 0], ext[:, 1], ext[:, 2], ext[:, 3], c[None])

    y, x, u, v = _rays(4000, 14.0, 3, ang=0.1)
    c10 = np.zeros((coef.shape[0], 11)); c10[:, :7] = coef; c10[1, 8] = 3e-14; c10[5, 10] = -2e-16          # even, 6 terms
    c_odd = coef.copy(); c_odd[7, 3] = 4e-6; c_odd[1, 5] = -3e-9                                               # general, <= 8
    c_odd12 = np.zeros((coef.shape[0], 12)); c_odd12[:, :7] = c_odd; c_odd12[11, 11] = 1e-18; c_odd12[5, 9] = 2e-15
    cases = {"config3": coef, "even6": c10, "odd8": c_odd, "odd12": c_odd12}
    builds = {"config3": 2, "even6": 2, "odd8": 3, "odd12": 3}          # ARMS_EVEN for the even aspheres, ARMS_POLY otherwise
    for tag, c in cases.items():
        pres = pres_with(c)
        emu.trace(pres, y[:1], x[:1], u[:1], v[:1], fast=True)
        assert emu.last_arms == builds[tag], (tag, emu.last_arms)
        ox, oy = oracle_engine.skew(pres, y, x, u, v, slopes=True)
        ex, ey, _ = emu.trace(pres, y, x, u, v, fast=False)
        # the reference takes p' by a complex step (RayTracing.jl:103), the device analytically: O(eps^2) apart
        assert max(cm.rel_err(ex, ox, 1.0).max(), cm.rel_err(ey, oy, 1.0).max()) <= 1e-11, tag
        fx, fy, odd = emu.trace_fast_with_retrace(pres, y, x, u, v)
        assert np.array_equal(np.isnan(fx), np.isnan(ox)), tag
        assert max(cm.rel_err(fx, ox, 1.0).max(), cm.rel_err(fy, oy, 1.0).max()) <= 1e-11, tag
        assert odd.mean() < 0.01, tag
    # an even asphere beside a row the even-asphere build does not carry (|R| > 1e3: vertex-form sphere) -> the full build
    weak = ext.copy(); weak[3, 0] = 2500.0
    pres = Prescription(weak[:, 0], weak[:, 1], weak[:, 2], weak[:, 3], np.vstack([coef, np.zeros((1, coef.shape[1]))])[None])
    ox, oy = oracle_engine.skew(pres, y, x, u, v, slopes=True)
    fx, fy, odd = emu.trace_fast_with_retrace(pres, y, x, u, v)
    assert emu.last_arms == 3
    assert np.array_equal(np.isnan(fx), np.isnan(ox)) and max(cm.rel_err(fx, ox, 1.0).max(), cm.rel_err(fy, oy, 1.0).max()) <= 1e-11
    # a plane surface with a polynomial: sag = 0 WITHOUT p(y) (PupilSampling.jl:12), tilt = p' only (:18)
    S = np.array([[math.inf, 0.0, 1.0, 0.0], [math.inf, 4.0, 1.52, 0.0], [-80.0, 30.0, 1.0, 0.0], [math.inf, 0.0, 1.0, 0.0]])
    c = np.zeros((4, 7)); c[1, 2] = 1e-4; c[1, 4] = -3e-7
    pres = Prescription(S[:, 0], S[:, 1], S[:, 2], S[:, 3], c[None])
    y, x, u, v = _rays(2000, 8.0, 5, ang=0.05)
    ox, oy = oracle_engine.skew(pres, y, x, u, v, slopes=True)
    fx, fy, odd = emu.trace_fast_with_retrace(pres, y, x, u, v)
    assert not odd.any() and max(cm.rel_err(fx, ox, 1.0).max(), cm.rel_err(fy, oy, 1.0).max()) <= 1e-12


def _random_system(rng, rows, aspheric):
    R = rng.uniform(20.0, 500.0, rows) * rng.choice([-1.0, 1.0], rows)
    R[rng.random(rows) < 0.2] = math.inf
    R[0] = math.inf
    t = rng.uniform(0.5, 12.0, rows); t[0] = rng.uniform(0.0, 5.0); t[-1] = 0.0
    n = np.ones(rows)
    glass = False
    for i in range(1, rows):
        glass = not glass if rng.random() < 0.7 else glass
        n[i] = rng.uniform(1.45, 1.9) if glass else 1.0
    K = np.zeros(rows); coef = np.zeros((rows, 11 if aspheric == "even" else 7))
    if aspheric == "even":
        # the usual aspheric lens: even polynomial terms (4th .. 10th order, sometimes a 2nd-order term) on curved rows, a conic
        # constant only where there is a polynomial -> the even-asphere kernel build (ARMS_EVEN), both of its forms
        for i in range(1, rows):
            if math.isfinite(R[i]) and rng.random() < 0.4:
                K[i] = rng.uniform(-1.5, 0.5) if rng.random() < 0.7 else 0.0
                coef[i, 4] = rng.uniform(-2e-7, 2e-7); coef[i, 6] = rng.uniform(-5e-10, 5e-10)
                if rng.random() < 0.5:
                    coef[i, 8] = rng.uniform(-1e-12, 1e-12); coef[i, 10] = rng.uniform(-2e-15, 2e-15)
                if rng.random() < 0.3:
                    coef[i, 2] = rng.uniform(-2e-4, 2e-4)
    elif aspheric:
        for i in range(1, rows):
            if math.isfinite(R[i]) and rng.random() < 0.4:
                K[i] = rng.uniform(-1.5, 0.5)
            if rng.random() < 0.25:
                coef[i, 4] = rng.uniform(-2e-7, 2e-7); coef[i, 6] = rng.uniform(-5e-10, 5e-10)
    return R, t, n, K, coef


@pytest.mark.parametrize("wide", [False, True])
def test_fast_policy_is_status_exact_by_construction(oracle_engine, wide):
    """Random prescriptions (flat rows, both curvature signs, conics, polynomial rows, TIR and miss sequences; `wide`:
    |R| 6.5-30 mm under +-14 mm, +-0.2 rad bundles — thousands of far-cap hits): what a MATH_FAST kernel leaves
    (fast forms, `odd` rays retraced with the reference sequence) has the oracle's NaN pattern / status on EVERY ray,
    every ray whose branch margin is below 1e-10 IS flagged, and the coordinates are within the bar on every ray."""
    rng = np.random.default_rng(31337 if wide else 2024)
    ntot = nodd = nill = 0
    for case in range(40):
        rows = int(rng.integers(3, 15))
        aspheric = (True, "even", False)[case % 3]
        R, t, n, K, coef = _random_system(rng, rows, aspheric)
        if wide:
            fin = np.isfinite(R); R[fin] = np.sign(R[fin]) * rng.uniform(6.5, 30.0, int(fin.sum()))
        pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
        m = 600
        a, ang = (14.0, 0.2) if wide else (6.0, 0.1)
        y = rng.uniform(-a, a, m); x = rng.uniform(-a, a, m)
        u = np.tan(rng.uniform(-ang, ang, m)); v = np.tan(rng.uniform(-ang, ang, m))
        ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        d = 1e-13
        px, py = oracle_engine.skew(pres, y * (1 + d), x * (1 - d), u * (1 + d), v * (1 - d), slopes=True)
        fx, fy, odd = emu.trace_fast_with_retrace(pres, y, x, u, v)
        assert np.array_equal(_status(fx, fy), os_), case
        assert np.array_equal(np.isnan(fx), np.isnan(ox)) and np.array_equal(np.isnan(fy), np.isnan(oy)), case
        scale = np.maximum(1.0, np.maximum(np.nanmax(np.abs(ox), axis=0, initial=0.0), np.nanmax(np.abs(oy), axis=0, initial=0.0)))
        dev = lambda ax, ay: np.maximum(np.nanmax(np.abs(ax - ox), axis=0, initial=0.0), np.nanmax(np.abs(ay - oy), axis=0, initial=0.0)) / scale
        err, sens = dev(fx, fy), dev(px, py)
        bad = ~(err <= np.maximum(TOL, 100.0 * sens))
        assert not bad.any(), (case, int(bad.sum()), float(err[bad].max()))
        mg = oracle_engine.skew_margins(pres, y, x, u, v)
        near = np.min(np.abs(mg[:, :3]), axis=1) < 1e-10
        assert odd[near].all(), (case, "a near-branch ray was not flagged")
        ntot += m; nodd += int(odd.sum()); nill += int((err > TOL).sum())
    assert nill <= 2e-3 * ntot, (nill, ntot)
    assert nodd < (0.5 if wide else 0.05) * ntot, (nodd, ntot)       # the retrace is the exception, not the rule


def test_fast_atan2_against_libm():
    """theta under the FAST policy (ort::fast_atan2, one division + a degree-9 polynomial) against numpy's atan2 on every
    octant, tiny and huge ratios, the axes and NaN: <= 1e-15 absolute (the parity bar is 1e-10)."""
    rng = np.random.default_rng(5)
    n = 400000
    x = rng.uniform(-10, 10, n); y = rng.uniform(-10, 10, n)
    x[::7] *= 1e-6; y[::11] *= 1e-7; x[::13] *= 1e5
    got = emu.fast_atan2(y, x); ref = np.arctan2(y, x)
    assert np.abs(got - ref).max() <= 1e-15
    ys = np.array([0.0, 1.0, -1.0, 0.0, 0.0, 2.0, -2.0, 1e-30, np.nan, 1.0])
    xs = np.array([1.0, 0.0, 0.0, -1.0, 0.0, 2.0, -2.0, 1e-30, 1.0, np.nan])
    g = emu.fast_atan2(ys, xs); r = np.arctan2(ys, xs)
    assert np.allclose(g[:8], r[:8], rtol=0, atol=1e-15) and np.isnan(g[8]) and np.isnan(g[9])
