"""Host-side logic of the API mirror (no GPU): range restatement, containers, dispatch."""
import math

import numpy as np
import pytest

import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import api
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
