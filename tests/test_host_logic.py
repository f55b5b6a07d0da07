"""Host-side logic of the API mirror (no GPU): range restatement, containers, dispatch."""
import math

import numpy as np
import pytest

import opticalraytracing_jl_amd as ort
from oracle import cpu as oc
from tests import common as cm


def test_linrange_matches_binary128_lerp():
    rng = np.random.default_rng(3)
    for _ in range(40):
        a, b = rng.uniform(-50, 50, 2)
        n = int(rng.integers(2, 200))
        mine = ort.linrange(a, b, n)
        ref = oc.linrange(a, b, n)
        assert np.array_equal(mine, ref)
        assert mine[0] == a and mine[-1] == b
    assert ort.linrange(0.0, 1.0, 11)[3] == 0.3          # Julia: range(0, 1, 11)[4] == 0.3 exactly
    assert np.array_equal(ort.linrange(2.0, 2.0, 1), [2.0])


def test_lens_mutates_input_like_reference():
    s = np.array([[math.inf, math.inf, 1.0], [50.0, 3.0, 1.5], [-50.0, 0.0, 1.0]])
    L = ort.Lens(s)
    assert s[0, 1] == 0.0                                # t[1] *= isfinite(t[1])  (RayTracing.jl:42)
    assert L.M.shape == (2, 2)                           # last row dropped (t[end] == 0)
    s2 = np.array([[math.inf, 0.0, 1.0], [50.0, 3.0, 1.5], [-50.0, 7.0, 1.0]])
    L2 = ort.Lens(s2)
    assert L2.M.shape == (3, 2) and L2.M[-1, 1] == 0.0   # :50


def test_layout_variants():
    lay = ort.Layout(cm.cooke())
    assert lay.profile is ort.Spherical and lay.M.shape == (8, 4) and not lay.K.any()
    asp = ort.Layout(cm.parabola_M(), profile=ort.Aspheric)
    assert asp.profile is ort.Aspheric and asp.K[1] == -1.0
    five = ort.Layout([math.inf, 10.0], [0.0, 0.0], [1.0, 1.5], [0.0, -0.5], [None, [0.0, 0.0, 1e-3]])
    assert five.profile is ort.Aspheric and five.coef_table().shape == (2, 3)
    assert five.prescription().coef.shape == (1, 2, 3)


def test_paraxial_ray_fields(oracle_engine):
    s = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    m = s.marginal
    assert len(m.y) == len(m.u) == len(m.z) == 9
    assert m.z[-1] - m.z[-2] == pytest.approx(s.EBFD, rel=1e-12)
    assert s.trace.shape == (9, 4)
    assert s.layout.profile is ort.Spherical


def test_raytrace_dispatch_and_batch(oracle_engine):
    surf = cm.cooke()
    p = ort.raytrace(surf, 1.0, 0.0, engine=oracle_engine)
    assert isinstance(p, ort.ParaxialRay) and p.kind is ort.Tangential
    r = ort.raytrace(surf, 1.0, 0.0, ort.RealRay, engine=oracle_engine)
    assert isinstance(r, ort.RealRayT) and len(r.y) == 8 and np.allclose(np.diff(r.z), r.z[1:] - r.z[:-1])
    xv, yv = ort.raytrace(surf, 1.0, 0.5, 0.0, 0.0, ort.VectorRealRay, engine=oracle_engine)
    assert xv.shape == (7,)
    XV, YV = ort.raytrace(surf, [1.0, 2.0, 3.0], 0.5, 0.0, 0.0, ort.VectorRealRay, engine=oracle_engine)
    assert XV.shape == (7, 3) and np.array_equal(XV[:, 0], xv)
    rays = ort.raytrace(surf, [1.0, 2.0], 0.0, ort.RealRay, engine=oracle_engine)
    assert len(rays) == 2 and np.array_equal(rays[0].y, r.y)


def test_extended_prescription():
    lay = ort.Layout(cm.cooke())
    pres = ort.extended_prescription(lay, 77.4)
    assert pres.rows == 9 and pres.t[0, -2] == 77.4 and math.isinf(pres.R[0, -1]) and pres.n[0, -1] == 1.0


def test_full_trace_raybasis_route(oracle_engine):
    """full_trace(surfaces, ray_basis): finite-conjugate branch of src/PupilSampling.jl:104-108,124-127
    (per-ray U = (ybar - y)/z0, V = -x/z0 placed in the ANGLE slots, Q8) — runs end to end and is
    consistent with tracing the same rays explicitly."""
    surf = cm.cooke()
    s = ort.solve(surf, cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    rays = ort.raytrace(s, -8.0, -400.0)                       # RayBasis for an object 400 mm in front
    assert isinstance(rays, ort.RayBasis)
    lay = ort.Layout(surf)
    aim = ort.full_trace_aim(lay, rays, 1.0, engine=oracle_engine)
    assert aim.raybasis and aim.z0 == rays.marginal.z[0]
    err = ort.full_trace(surf, rays, 16, engine=oracle_engine)  # (surfaces, ray_basis, k_rays)
    assert err.H == 1.0 and len(err.x) == len(err.y) and len(err.x) > 0 and math.isfinite(err.RMS)
    # the same grid traced explicitly with the per-ray angles reproduces the first survivor
    pres = ort.extended_prescription(lay, aim.focus)
    yax, xax = ort.linrange(aim.y1, aim.y2, 16), ort.linrange(0.0, aim.y_EP, 8)
    U = (aim.ybar - yax[0]) / aim.z0; V = -xax[0] / aim.z0
    xv, yv = oracle_engine.skew(pres, yax[0], xax[0], U, V)
    if np.hypot(xv[aim.stop - 1, 0], yv[aim.stop - 1, 0]) <= aim.a_stop:
        assert err.x[0] == xv[-1, 0] and err.y[0] == yv[-1, 0] - aim.hprime


def test_real_trace_keyword_forms(oracle_engine):
    """raytrace(surfaces, y, U, RealRay; K, p) with explicit keywords == the Layout{Aspheric} method
    for K (both take atan, Q16: K != 0), and a zero polynomial row behaves like `zero`."""
    P = cm.parabola_M()
    lay = ort.Layout(P, profile=ort.Aspheric)
    r1 = ort.raytrace(lay, 12.0, 0.0, ort.RealRay, engine=oracle_engine)
    r2 = ort.raytrace(P[:, :3], 12.0, 0.0, ort.RealRay, K=P[:, 3], engine=oracle_engine)
    assert np.array_equal(r1.y, r2.y) and np.array_equal(r1.u, r2.u)
    surf = cm.cooke()
    base = ort.raytrace(surf, 3.0, 0.02, ort.RealRay, engine=oracle_engine)
    zero_p = ort.raytrace(surf, 3.0, 0.02, ort.RealRay, p=[None] * 8, engine=oracle_engine)
    assert np.array_equal(base.y, zero_p.y)
    bent = ort.raytrace(surf, 3.0, 0.02, ort.RealRay, p=[None, [0, 0, 0, 0, 1e-5]] + [None] * 6, engine=oracle_engine)
    assert not np.array_equal(base.y, bent.y) and np.allclose(base.y, bent.y, atol=0.1)
    xv, yv = ort.raytrace(surf, 3.0, 1.0, 0.02, 0.0, ort.VectorRealRay, K=np.zeros(8), engine=oracle_engine)
    xv0, yv0 = ort.raytrace(surf, 3.0, 1.0, 0.02, 0.0, ort.VectorRealRay, engine=oracle_engine)
    assert np.array_equal(xv, xv0) and np.array_equal(yv, yv0)


def test_paraxial_primitives_and_raypoints(oracle_engine):
    """transfer / refract scalar forms (RayTracing.jl:55-69), scale! (:9-12), raypoints (RayPlot.jl:4-24)."""
    import math
    assert ort.transfer(1.0, 0.1, 5.0) == 1.5 and ort.transfer(1.0, 0.1, math.inf) == 1.0
    assert ort.transfer(1.0, 0.1, 5.0, 0.02) == (1.5, 0.1 - 1.5 * 0.02) and ort.refract(2.0, 0.3, 0.1) == 0.3 - 0.2
    system = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    from tests import ref_consumers as rc
    z, ys = rc.raypoints(system)
    assert len(ys) == 6 and all(len(y) == len(z) for y in ys)
    assert np.array_equal(ys[2], -ys[1]) and np.allclose(ys[4] - ys[3], ys[1]) and np.allclose(ys[5] - ys[3], ys[2])
    lens = ort.Lens(cm.cooke())
    phi = lens.M[:, 1].copy()
    assert rc.scale(lens) is lens and np.allclose(lens.M[:, 1], phi * 1e-3)


def test_wavegrad_is_the_references_two_operations(oracle_engine):
    """wavegrad(eps, lambda) = map(f -> getfield(eps, f) * eps.nu / lambda, (:x, :y)) (src/PupilSampling.jl:165-167): the
    transverse errors in waves, product first, then quotient; default lambda = 587.5618e-6 (SeidelAberrations.jl:2)."""
    import numpy as np
    import opticalraytracing_jl_amd as ort
    from tests import common as cm
    s = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    e = ort.full_trace(s, 1.0, engine=oracle_engine)
    wx, wy = ort.wavegrad(e)
    assert np.array_equal(wx, (e.x * e.nu) / 587.5618e-6) and np.array_equal(wy, (e.y * e.nu) / 587.5618e-6)
    wx2, _ = ort.wavegrad(e, 500e-6)
    assert np.allclose(wx2 * 500e-6, wx * 587.5618e-6, rtol=1e-15)
    # a 1 um transverse error at n'u' = -0.2 is 0.34 waves at the d line
    assert abs(1e-3 * -0.2 / 587.5618e-6 + 0.340390) < 1e-6


def test_best_placed_keeps_the_fastest_candidate():
    """opticalraytracing_jl_amd/placement.py: candidates are all alive while they are timed, the fastest is returned, the report
    lists every candidate's time; one candidate = no probing."""
    from opticalraytracing_jl_amd.placement import best_placed
    made, alive_when_timed = [], []
    times = {0: 3.0, 1: 1.5, 2: 2.0, 3: 1.75}

    def make():
        made.append(len(made)); return made[-1]

    def time_ms(b):
        alive_when_timed.append(len(made)); return times[b]
    chosen, rep = best_placed(make, time_ms, candidates=4)
    assert chosen == 1 and rep["chosen"] == 1 and rep["candidates_ms"] == [3.0, 1.5, 2.0, 1.75]
    assert alive_when_timed == [4, 4, 4, 4]                       # every candidate existed before the first was timed
    made.clear()
    chosen, rep = best_placed(make, time_ms, candidates=1)
    assert chosen == 0 and rep["candidates_ms"] == [] and len(made) == 1
