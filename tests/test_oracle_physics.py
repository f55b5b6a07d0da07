"""The oracle against first principles, off the meridional plane.

The reference's own known answers pin the skew loop (`src/PupilSampling.jl:34-65`) on MERIDIONAL rays only (3-D = 2-D
identities, `test/runtests.jl:355-363`) plus two spot sizes at the per-cent level.  These tests close the
gap from the other side: they use nothing of the reference's formulas — only the prescription's geometry (vertex
positions, the conic z(r) and its gradient) and the hit points the oracle returns — and check that consecutive hit
points obey the VECTOR law of refraction at every surface for general skew rays (x != 0, V != 0), that the skew
invariant n (x M - y L) is conserved through the system, and that rotating a ray about the optical axis rotates its
hit points with it.  A restatement error in the x path, in the normal, in the root choice or in the refraction would
break these at the 1e-3 level; they hold at 1e-11.

What they do NOT cover are the places where the reference is deliberately not the geometry (rays continuing after a
miss / TIR with NaN or an unchanged direction, far-cap hits, the polynomial term p(y) that depends on y alone): those
are the restatement's to carry, line by line, and the bit-exact suites compare the device with it.
"""
import math

import numpy as np
import pytest

from tests import common as cm
from oracle.cpu import OracleEngine
from opticalraytracing_jl_amd.api import Prescription


@pytest.fixture(scope="module")
def orc():
    return OracleEngine(nthreads=4)


def _ext(surfaces, focus):
    e = np.vstack([surfaces, [math.inf, 0.0, 1.0]])
    e[-2, 1] = focus
    return e


def _conic_z_and_normal(x, y, R, K):
    """z(x, y) of the conic with vertex at 0 and the (unnormalised) normal (-dz/dx, -dz/dy, 1)."""
    if not math.isfinite(R):
        return np.zeros_like(x), np.stack([np.zeros_like(x), np.zeros_like(x), np.ones_like(x)])
    c = 1.0 / R
    r2 = x * x + y * y
    root = np.sqrt(1.0 - (1.0 + K) * c * c * r2)
    z = c * r2 / (1.0 + root)
    return z, np.stack([-c * x / root, -c * y / root, np.ones_like(x)])


def _points(R, t, n, K, y0, x0, xv, yv):
    """Hit points P_0 .. P_S in one frame: P_0 the launch point (a plane t[0] before the first vertex)."""
    S = xv.shape[0]
    zv = np.concatenate([[0.0], np.cumsum(t[1:S])])            # vertex of surface i+1 (0-based i)
    P = [np.stack([x0, y0, np.full_like(x0, -t[0])])]
    N = [None]
    for i in range(S):
        z, nr = _conic_z_and_normal(xv[i], yv[i], R[i + 1], K[i + 1])
        P.append(np.stack([xv[i], yv[i], zv[i] + z]))
        N.append(nr)
    return P, N


def _unit(d):
    """Direction of travel along the chord d between two hit points.  These systems are all-refractive and every ray
    travels towards +z; where a steep surface reaches past the next vertex plane (Cooke surface 4 -> stop at large
    heights) the NEXT hit lies behind the current one and the chord points backwards along the same line."""
    return d * (np.sign(d[2]) / np.sqrt((d * d).sum(axis=0)))


def _snell_residuals(R, t, n, K, y0, x0, xv, yv):
    """max over surfaces of | n1 d1 x N - n2 d2 x N | / |N| per ray, and the drift of the skew invariant."""
    P, N = _points(R, t, n, K, y0, x0, xv, yv)
    S = xv.shape[0]
    worst = np.zeros(x0.shape)
    d_in = _unit(P[1] - P[0])
    skew0 = n[0] * (P[0][0] * d_in[1] - P[0][1] * d_in[0])
    drift = np.zeros(x0.shape)
    for i in range(1, S):                                       # surfaces 1 .. S-1 have a successor
        d1 = _unit(P[i] - P[i - 1]); d2 = _unit(P[i + 1] - P[i])
        nr = N[i] / np.sqrt((N[i] * N[i]).sum(axis=0))
        res = n[i - 1] * np.cross(d1, nr, axis=0) - n[i] * np.cross(d2, nr, axis=0)
        worst = np.maximum(worst, np.sqrt((res * res).sum(axis=0)))
        assert (np.sign((d1 * nr).sum(axis=0)) == np.sign((d2 * nr).sum(axis=0))).all()   # transmitted, not reflected
        sk = n[i] * (P[i][0] * d2[1] - P[i][1] * d2[0])
        drift = np.maximum(drift, np.abs(sk - skew0))
    return worst, drift


def _skew_rays(nr, a1, seed):
    rng = np.random.default_rng(seed)
    y = rng.uniform(-0.8 * a1, 0.8 * a1, nr); x = rng.uniform(-0.8 * a1, 0.8 * a1, nr)
    u = np.tan(rng.uniform(-0.15, 0.15, nr)); v = np.tan(rng.uniform(-0.15, 0.15, nr))
    return y, x, u, v


SYSTEMS = {
    "cooke": (lambda: _ext(cm.cooke(), 77.40534796682427), 12.0),
    "tessar": (lambda: _ext(cm.tessar(), 40.0), 7.0),
    "double_gauss": (lambda: _ext(cm.double_gauss(), 57.8), 20.0),
    "singlet": (lambda: _ext(cm.singlet(), 90.0), 18.0),
}


@pytest.mark.parametrize("name", sorted(SYSTEMS))
def test_skew_rays_obey_vector_snell_and_skew_invariant(orc, name):
    make, a1 = SYSTEMS[name]
    M = make()
    R, t, n = M[:, 0], M[:, 1], M[:, 2]
    K = np.zeros(len(R))
    y, x, u, v = _skew_rays(20000, a1, 11)
    xv, yv, st = orc.skew(Prescription.from_matrix(M), y, x, u, v, slopes=True, want_status=True)
    ok = st == len(R)                                           # reached the image plane: no miss, no TIR
    assert ok.mean() > 0.6, ok.mean()
    worst, drift = _snell_residuals(R, t, n, K, y[ok], x[ok], xv[:, ok], yv[:, ok])
    assert worst.max() <= 1e-11, worst.max()
    assert drift.max() <= 1e-10, drift.max()                    # mm: |x M - y L| is of order 1
    assert np.abs(x[ok] * u[ok] - y[ok] * v[ok]).max() > 1.0    # the set really is skew


def test_conic_systems_obey_vector_snell(orc):
    """Random conic prescriptions (K in [-1.5, 0.5], both curvature signs, flat rows, glass / air sequences)."""
    rng = np.random.default_rng(77)
    checked = 0
    for case in range(40):
        rows = int(rng.integers(3, 12))
        R = rng.uniform(25.0, 400.0, rows) * rng.choice([-1.0, 1.0], rows)
        R[rng.random(rows) < 0.2] = math.inf
        R[0] = math.inf; R[-1] = math.inf
        t = rng.uniform(1.0, 10.0, rows); t[0] = rng.uniform(0.0, 5.0); t[-1] = 0.0
        n = np.ones(rows); glass = False
        for i in range(1, rows - 1):
            glass = not glass if rng.random() < 0.7 else glass
            n[i] = rng.uniform(1.45, 1.9) if glass else 1.0
        K = np.where(np.isfinite(R), rng.uniform(-1.5, 0.5, rows), 0.0); K[0] = 0.0
        y, x, u, v = _skew_rays(2000, 6.0, 1000 + case)
        xv, yv, st = orc.skew(Prescription(R, t, n, K, None), y, x, u, v, slopes=True, want_status=True)
        ok = st == rows
        if ok.sum() < 100:
            continue
        worst, drift = _snell_residuals(R, t, n, K, y[ok], x[ok], xv[:, ok], yv[:, ok])
        assert worst.max() <= 1e-10, (case, worst.max())
        assert drift.max() <= 1e-9, (case, drift.max())
        checked += int(ok.sum())
    assert checked > 30000


@pytest.mark.parametrize("name", ["cooke", "double_gauss"])
def test_rotation_about_the_axis_rotates_the_hits(orc, name):
    """(x, y, v, u) -> rotation by phi: every hit point rotates by phi (spheres / conics: no p(y) rows)."""
    make, a1 = SYSTEMS[name]
    M = make()
    pres = Prescription.from_matrix(M)
    y, x, u, v = _skew_rays(5000, a1, 5)
    xv, yv, st = orc.skew(pres, y, x, u, v, slopes=True, want_status=True)
    for phi in (0.3, 1.0, math.pi / 2, 2.5):
        c, s = math.cos(phi), math.sin(phi)
        xr, yr = c * x - s * y, s * x + c * y
        vr, ur = c * v - s * u, s * v + c * u                   # slopes dx/dz, dy/dz rotate like (x, y)
        xw, yw, sw = orc.skew(pres, yr, xr, ur, vr, slopes=True, want_status=True)
        ok = (st == M.shape[0]) & (sw == M.shape[0])
        assert (st == sw).mean() > 0.999                        # (a ray grazing a miss may flip under rotation)
        ex = np.abs(xw[:, ok] - (c * xv[:, ok] - s * yv[:, ok])).max()
        ey = np.abs(yw[:, ok] - (s * xv[:, ok] + c * yv[:, ok])).max()
        assert max(ex, ey) <= 1e-11 * max(1.0, np.abs(xv[:, ok]).max()), (phi, ex, ey)
