"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): Float64 coordinates within 1e-10 relative
(denominator max(|ref|, 1 mm) — coordinates legitimately cross 0), status / surface-hit index
bit-exact.  The default arithmetic policy performs the reference's operation sequence with
IEEE / and sqrt, so wherever no libm call is involved we assert BITWISE equality.
"""
import math

import numpy as np
import pytest

import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import Prescription
from tests import common as cm

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _random_rays(n, a1, seed=1):
    rng = np.random.default_rng(seed)                       # SURVEY §8d random-ray parity set
    y = rng.uniform(-0.9 * a1, 0.9 * a1, n)
    x = rng.uniform(-0.9 * a1, 0.9 * a1, n)
    U = rng.uniform(-0.2, 0.2, n)
    V = rng.uniform(-0.2, 0.2, n)
    return y, x, U, V


def _ext(surfaces, focus):
    e = np.vstack([surfaces, [math.inf, 0.0, 1.0]])
    e[-2, 1] = focus
    return e


def _engine(policy):
    """The default (reference-sequence) engine or a FAST-policy one on the same device."""
    return ort.default_engine() if policy == "ieee" else ort.HipEngine(0, fast_math=True)


def _same(policy, g, o, what=None):
    """Bit-identical in the reference-sequence policy; NaN pattern identical and <= 1e-10 relative in the FAST one."""
    if policy == "ieee":
        assert np.array_equal(g, o, equal_nan=True), what
    else:
        assert np.array_equal(np.isnan(g), np.isnan(o)), what
        assert cm.rel_err(g, o, 1.0).max() <= TOL, (what, cm.rel_err(g, o, 1.0).max())


@pytest.mark.parametrize("name", ["cooke", "tessar", "catadioptric", "double_gauss"])
def test_skew_list_bitexact_with_slopes(hip_engine, oracle_engine, name):
    M = {"cooke": _ext(cm.cooke(), 77.40534796682427), "tessar": _ext(cm.tessar(), 40.0),
         "catadioptric": cm.catadioptric(), "double_gauss": _ext(cm.double_gauss(), 57.8)}[name]
    pres = Prescription.from_matrix(M)
    a1 = {"cooke": 14.7, "tessar": 9.5, "catadioptric": 15.0, "double_gauss": 29.0}[name]
    y, x, U, V = _random_rays(20001, a1)                    # odd count: exercises the tail lane
    u, v = np.tan(U), np.tan(V)
    gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    assert np.array_equal(gs, os_)
    assert np.array_equal(gx, ox, equal_nan=True)
    assert np.array_equal(gy, oy, equal_nan=True)


def test_skew_list_angles(hip_engine, oracle_engine):
    """tan() runs on the device here (ocml vs glibc: last-ulp differences)."""
    pres = Prescription.from_matrix(_ext(cm.cooke(), 77.40534796682427))
    y, x, U, V = _random_rays(4096, 14.7, seed=2)
    gx, gy, gs = hip_engine.skew(pres, y, x, U, V, want_status=True)
    ox, oy, os_ = oracle_engine.skew(pres, y, x, U, V, want_status=True)
    assert np.array_equal(gs, os_)
    assert cm.rel_err(gx, ox, 1.0).max() <= TOL
    assert cm.rel_err(gy, oy, 1.0).max() <= TOL


def test_miss_and_tir_status(hip_engine, oracle_engine):
    """Edge cases of the reference: surface miss -> NaN from that surface on
    (PupilSampling.jl:9); TIR leaves the ray undeviated (Q1)."""
    pres = Prescription.from_matrix(_ext(cm.cooke(), 77.4))
    y = np.array([0.0, 36.0, 38.0, 60.0, -45.0, 10.0, 5.0, 14.0])
    x = np.array([0.0, 10.0, 0.0, 0.0, 30.0, 10.0, 40.0, 0.0])
    U = np.array([0.0, 0.0, 0.0, 0.3, -0.2, 0.9, 0.0, -1.2])
    V = np.array([0.0, 0.0, 0.0, 0.0, 0.1, 0.9, 0.0, 0.0])
    u, v = np.tan(U), np.tan(V)
    gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    assert (os_ < pres.rows).any() and (os_ == pres.rows).any()      # both kinds present
    assert np.array_equal(gs, os_)
    assert np.array_equal(np.isnan(gx), np.isnan(ox))
    assert np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True)
    # a dense singlet of high index: steep rays meet the glass->air TIR branch
    tir = Prescription.from_matrix(np.array([[math.inf, 0.0, 1.0], [30.0, 25.0, 1.9], [-30.0, 10.0, 1.0],
                                             [math.inf, 0.0, 1.0]]))
    yy = np.linspace(-26.0, 26.0, 257)
    g = hip_engine.skew(tir, yy, 0.3 * yy, 0.0 * yy, 0.0 * yy, slopes=True, want_status=True)
    o = oracle_engine.skew(tir, yy, 0.3 * yy, 0.0 * yy, 0.0 * yy, slopes=True, want_status=True)
    assert np.array_equal(g[2], o[2])
    assert np.array_equal(g[0], o[0], equal_nan=True) and np.array_equal(g[1], o[1], equal_nan=True)


def _dg_bundles(engine, k, fields=(0.0, 0.7, 1.0), lines=(0, 1, 2)):
    """3 fields x 3 index columns of the Double-Gauss, square pupil k x k (BASELINE config 2)."""
    from opticalraytracing_jl_amd import api, workloads
    systems = [ort.solve(cm.double_gauss(line), cm.DG_A, cm.DG_H, engine=engine) for line in lines]
    return workloads.square_pupil_bundles(api, systems, k, fields=fields)


@pytest.mark.parametrize("policy", ["ieee", "fast"])
@pytest.mark.parametrize("H", [0.0, 1.0])
def test_grid_cooke_config1_bitexact(oracle_engine, H, policy):
    """BASELINE config 1 (H = 0 and H = 1): Cooke triplet, 1 field, 64 x 32 half pupil (reference
    mode) — grid history / summary bit-identical (FAST: <= 1e-10, status and survivor counts exact), and the full_trace
    pipeline (filter, ordered compaction, mirror, rho, theta, RMS) against the oracle's."""
    hip_engine = _engine(policy)
    system = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    aim = ort.full_trace_aim(system.layout, system, H, engine=oracle_engine)
    pres = ort.extended_prescription(system.layout, aim.focus)
    axes = np.concatenate([ort.linrange(aim.y1, aim.y2, 64), ort.linrange(0.0, aim.y_EP, 32)])
    b = dict(system=0, stop=aim.stop, U=aim.U, V=0.0, a_stop=aim.a_stop, hprime=aim.hprime, yaxis_off=0, xaxis_off=64)
    g = hip_engine.grid(pres, [b], axes, 64, 32)
    o = oracle_engine.grid(pres, [b], axes, 64, 32)
    for key in ("xv", "yv", "xf", "yf", "xs", "ys"):
        _same(policy, g[key], o[key], key)
    assert np.array_equal(g["status"], o["status"])
    of = oracle_engine.full_trace_grid(pres, [b], axes, 64, 32)[0]
    for lookback in (False, True):                              # both compaction routes
        gf = hip_engine.full_trace_grid(pres, [b], axes, 64, 32, lookback=lookback)[0]
        assert gf["count"] == of["count"] > 0
        _same(policy, gf["ex"], of["ex"], "ex"); _same(policy, gf["ey"], of["ey"], "ey")
        assert cm.rel_err(gf["rho"], of["rho"], 1e-3).max() <= TOL and cm.rel_err(gf["theta"], of["theta"], 1e-3).max() <= TOL
        assert abs(gf["rms"] - of["rms"]) <= TOL * of["rms"]


def test_grid_double_gauss_config2_small(hip_engine, oracle_engine):
    """Config 2 shape at 96 x 96 per bundle (oracle-sized): 9 bundles, 3 systems, bit-exact."""
    pres, bundles, axes = _dg_bundles(oracle_engine, 96)
    g = hip_engine.grid(pres, bundles, axes, 96, 96)
    o = oracle_engine.grid(pres, bundles, axes, 96, 96)
    for key in ("xv", "yv", "xf", "yf", "xs", "ys"):
        assert np.array_equal(g[key], o[key], equal_nan=True), key
    assert np.array_equal(g["status"], o["status"])
    assert ((o["status"] >> 16) & 1).any()                   # some rays fail the stop filter


def test_grid_odd_sizes(oracle_engine):
    """Ragged tiles (rays per bundle not a multiple of the 512-ray tile, odd nx)."""
    eng = ort.default_engine()
    pres, bundles, axes = _dg_bundles(oracle_engine, 37, fields=(0.0, 1.0), lines=(0, 2))
    g = eng.grid(pres, bundles, axes, 37, 37)
    o = oracle_engine.grid(pres, bundles, axes, 37, 37)
    for key in ("xv", "yv", "xf", "yf"):
        assert np.array_equal(g[key], o[key], equal_nan=True), key
    assert np.array_equal(g["status"], o["status"])


def test_fast_math_within_tolerance(oracle_engine):
    eng = ort.HipEngine(0, fast_math=True)
    pres, bundles, axes = _dg_bundles(oracle_engine, 64)
    g = eng.grid(pres, bundles, axes, 64, 64)
    o = oracle_engine.grid(pres, bundles, axes, 64, 64)
    assert np.array_equal(g["status"] & 0xffff, o["status"] & 0xffff)
    worst = max(cm.rel_err(g[k], o[k], 1.0).max() for k in ("xv", "yv"))
    assert worst <= 1e-12, worst                             # far inside the 1e-10 bar
    flips = int(np.count_nonzero(g["status"] != o["status"]))
    assert flips == 0, f"{flips} stop-filter flips"


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_aspheric_config3_small(oracle_engine, policy):
    """Config 3 shape: conic + polynomial terms on 4 surfaces.  The device uses the analytic
    p'; the reference a complex step (RayTracing.jl:103) — equal to O(eps^2).  FAST: the even-form polynomial arm
    (ort_device.hpp, surface_step_fast_poly) on realistic rays at the plain bar, no amplification."""
    hip_engine = _engine(policy)
    M4, coef = cm.double_gauss_aspheric()
    ext = np.vstack([M4, [math.inf, 0.0, 1.0, 0.0]])
    ext[-2, 1] = 57.8
    cext = np.vstack([coef, np.zeros((1, coef.shape[1]))])
    pres = Prescription(ext[:, 0], ext[:, 1], ext[:, 2], ext[:, 3], cext[None])
    y, x, U, V = _random_rays(30000, 16.0, seed=3)
    U *= 0.5; V *= 0.5
    u, v = np.tan(U), np.tan(V)
    gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    assert np.array_equal(gs, os_)
    assert max(cm.rel_err(gx, ox, 1.0).max(), cm.rel_err(gy, oy, 1.0).max()) <= 1e-12


@pytest.mark.parametrize("policy", ["ieee", "fast"])
@pytest.mark.parametrize("H", [0.0, 0.7, 1.0])
def test_full_trace_singlet(oracle_engine, H, policy):
    """The reference's own full_trace case (test/runtests.jl:364-372), grid stage on the GPU."""
    hip_engine = _engine(policy)
    system = ort.solve(cm.singlet(), [20.0, 20.0], 17.787, engine=oracle_engine)
    aim = ort.full_trace_aim(system.layout, system, H, engine=oracle_engine)
    g = ort.full_trace_grid(system.layout, aim, 64, engine=hip_engine)
    o = ort.full_trace_grid(system.layout, aim, 64, engine=oracle_engine)
    assert len(g.x) == len(o.x)                              # same survivors, same order
    _same(policy, g.x, o.x, "x"); _same(policy, g.y, o.y, "y")
    assert cm.rel_err(g.r, o.r, 1e-3).max() <= TOL           # hypot: ocml vs glibc
    assert cm.rel_err(g.t, o.t, 1e-3).max() <= TOL           # atan2
    assert abs(g.RMS - o.RMS) <= TOL * o.RMS
    assert abs(g.RMS - {0.0: 0.739649, 0.7: 1.1, 1.0: 1.4}[H]) < 0.07


def test_full_trace_end_to_end_on_gpu(hip_engine, oracle_engine):
    """solve + aiming + grid all through the GPU engine == all through the oracle."""
    sg = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=hip_engine)
    so = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    assert sg.f == so.f and sg.EBFD == so.EBFD and sg.stop == so.stop
    assert np.array_equal(sg.M.M, so.M.M)
    eg = ort.full_trace(sg, 0.7, engine=hip_engine)
    eo = ort.full_trace(so, 0.7, engine=oracle_engine)
    assert len(eg.x) == len(eo.x)
    assert cm.rel_err(eg.x, eo.x, 1.0).max() <= 1e-9        # the reference-sequence policy aims through device trig
    assert abs(eg.RMS - eo.RMS) <= 1e-9 * eo.RMS
    # ORT_FAST_MATH: the aiming loops trace a plain prescription without trigonometric calls; the loops stop at |residual| <=
    # sqrt(eps) (RayTracing.jl:1,229,282), so the two forms of the same function may leave their last iterate a few 1e-9 apart
    fast = ort.HipEngine(0, fast_math=True)
    ef = ort.full_trace(ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=fast), 0.7, engine=fast)
    assert len(ef.x) == len(eo.x)
    assert cm.rel_err(ef.x, eo.x, 1.0).max() <= 1e-7 and abs(ef.RMS - eo.RMS) <= 1e-7 * eo.RMS


def test_full_trace_multi_bundle_and_raybasis(hip_engine, oracle_engine):
    pres, bundles, axes = _dg_bundles(oracle_engine, 50)
    g = hip_engine.full_trace_grid(pres, bundles, axes, 50, 50)
    o = oracle_engine.full_trace_grid(pres, bundles, axes, 50, 50)
    for gb, ob in zip(g, o):
        assert gb["count"] == ob["count"] and gb["count"] > 0
        assert np.array_equal(gb["ex"], ob["ex"]) and np.array_equal(gb["ey"], ob["ey"])
        assert cm.rel_err(gb["rho"], ob["rho"], 1e-3).max() <= TOL
        assert abs(gb["rms"] - ob["rms"]) <= TOL * ob["rms"]
    # finite-conjugate (RayBasis) rule, PupilSampling.jl:124-127: angles per ray, tan on device
    for b in bundles:
        b["ybar"], b["z0"] = -40.0, -900.0
    g = hip_engine.full_trace_grid(pres, bundles[:2], axes, 50, 50, raybasis=True)
    o = oracle_engine.full_trace_grid(pres, bundles[:2], axes, 50, 50, raybasis=True)
    for gb, ob in zip(g, o):
        assert gb["count"] == ob["count"]
        assert cm.rel_err(gb["ex"], ob["ex"], 1.0).max() <= TOL
        assert abs(gb["rms"] - ob["rms"]) <= TOL * max(ob["rms"], 1e-3)
    # the same rule through the grid entry point (history + summary), both policies; bundles that SHARE their axes but not
    # their (ybar, z0): the launch slopes are taken per bundle (k_make_slope_axes), not per axis
    shared = [dict(b) for b in bundles[:3]]
    for j, b in enumerate(shared):
        b["yaxis_off"], b["xaxis_off"] = shared[0]["yaxis_off"], shared[0]["xaxis_off"]
        b["ybar"], b["z0"] = -40.0 + 15.0 * j, -900.0 - 100.0 * j
    og = oracle_engine.grid(pres, shared, axes, 50, 50, raybasis=True)
    for eng in (hip_engine, ort.HipEngine(0, fast_math=True)):
        gg = eng.grid(pres, shared, axes, 50, 50, raybasis=True)
        assert np.array_equal(gg["status"], og["status"])
        for key in ("xv", "yv", "xf", "yf", "xs", "ys"):
            assert np.array_equal(np.isnan(gg[key]), np.isnan(og[key])), key
            assert cm.rel_err(gg[key], og[key], 1.0).max() <= TOL, key
    assert not np.array_equal(og["xf"][:2500], og["xf"][2500:5000])        # the bundles do differ


def test_meridional(hip_engine, oracle_engine):
    for M, layout_mode in ((cm.cooke(), False), (cm.catadioptric(), False), (cm.tessar(), True)):
        pres = Prescription.from_matrix(M)
        rng = np.random.default_rng(5)
        y = rng.uniform(-8, 8, 999); U = rng.uniform(-0.2, 0.2, 999)
        g = hip_engine.meridional(pres, y, U, layout_mode)
        o = oracle_engine.meridional(pres, y, U, layout_mode)
        for a, b in zip(g, o):
            assert cm.rel_err(a, b, 1.0).max() <= TOL
    # conic, reflecting: the parabola of test/runtests.jl:335-338
    P = cm.parabola_M()
    pres = Prescription(P[:, 0], P[:, 1], P[:, 2], P[:, 3])
    y = np.linspace(1.0, 30.0, 64)
    g = hip_engine.meridional(pres, y, 0.0 * y, True)
    o = oracle_engine.meridional(pres, y, 0.0 * y, True)
    for a, b in zip(g, o):
        assert cm.rel_err(a, b, 1.0).max() <= TOL


def test_paraxial_and_abcd_bitexact(hip_engine, oracle_engine):
    rng = np.random.default_rng(11)
    nl, k = 7, 9
    tau = rng.uniform(0.0, 12.0, (nl, k)); phi = rng.uniform(-0.02, 0.02, (nl, k))
    tau[2, 0] = math.inf                                     # transfer skips non-finite τ (Q18)
    a = rng.uniform(3.0, 12.0, (nl, k))
    y = rng.uniform(-6, 6, nl * 300); w = rng.uniform(-0.2, 0.2, nl * 300)
    for clip in (False, True):
        g = hip_engine.paraxial(tau, phi, y, w, a, clip)
        o = oracle_engine.paraxial(tau, phi, y, w, a, clip)
        assert np.array_equal(g[0], o[0], equal_nan=True) and np.array_equal(g[1], o[1], equal_nan=True)
        if clip:
            assert np.isnan(o[0]).any() and not np.isnan(o[0]).all()
    tau[2, 0] = 1.0
    Mg, Mo = hip_engine.abcd(tau, phi), oracle_engine.abcd(tau, phi)
    assert np.array_equal(Mg, Mo)
    v = rng.uniform(-1, 1, (500, 2)); t1 = rng.uniform(-50, 50, 500); t2 = rng.uniform(-50, 50, 500)
    for rev in (False, True):
        assert np.array_equal(hip_engine.abcd_transfer(Mg[0], v, t1, t2, rev),
                              oracle_engine.abcd_transfer(Mo[0], v, t1, t2, rev))


def test_f32_build_extension(oracle_engine):
    """Float32 instantiation (BASELINE config 5; the reference itself is Float64-only, Q21):
    compared with the same loop in float on the CPU: bit-identical."""
    import ctypes as C
    from opticalraytracing_jl_amd import _capi
    from oracle import cpu as oc
    eng = ort.default_engine()
    M = _ext(cm.double_gauss(), 57.8)
    pres = Prescription.from_matrix(M)
    sysd = eng.system(pres)
    k = 128
    yax = np.linspace(-14, 14, k).astype(np.float32); xax = np.linspace(-14, 14, k).astype(np.float32)
    axes = np.concatenate([yax, xax])
    N, S = k * k, pres.rows - 1
    xv = np.empty((S, N), dtype=np.float32); yv = np.empty((S, N), dtype=np.float32)
    st = np.empty(N, dtype=np.int32)
    out = _capi.ort_grid_out_f32()
    out.xv, out.yv, out.ld, out.status = xv.ctypes.data, yv.ctypes.data, N, st.ctypes.data
    b = _capi.make_bundles([dict(system=0, stop=0, U=0.1, V=0.0, yaxis_off=0, xaxis_off=k)])
    _capi.check(eng.ctx.lib.ort_trace_grid_f32(eng.ctx.h, sysd.h, 1, b, axes.ctypes.data, axes.size, k, k,
                                               C.byref(out), 0))
    L = oc.lib()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    R, t, n = f(M[:, 0]), f(M[:, 1]), f(M[:, 2])
    oxv = np.empty((S, N), dtype=np.float32); oyv = np.empty((S, N), dtype=np.float32)
    ost = np.empty(N, dtype=np.int32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    L.orc_trace_skew_grid_f32(pres.rows, fp(R), fp(t), fp(n), None, None, 0, k, fp(yax), k, fp(xax),
                              np.float32(math.tan(0.1)), np.float32(0.0), fp(oxv), fp(oyv), N,
                              ost.ctypes.data_as(C.POINTER(C.c_int32)), 1)
    # the IEEE policy reproduces the float loop bit for bit as well (no libm call on this path: the slope
    # tan(U) is taken on the host in binary64 and rounded once, on both sides)
    assert np.array_equal(st, ost)
    assert np.array_equal(xv, oxv, equal_nan=True) and np.array_equal(yv, oyv, equal_nan=True)


def _snell_residual_torch(R, t, n, u, v, X, Y):
    """max over rays and surfaces of |n1 d1 x N - n2 d2 x N| for a spherical / flat prescription: d the unit chords
    between consecutive hit points (X, Y: [S][rays] on the device), the launch direction (slopes u = dy/dz, v = dx/dz)
    in front of the first surface — the chord from the launch plane is too short to define one near the axis —, N the
    unit normal of the sphere at the hit.  Nothing of the reference's formulas enters."""
    import torch
    S = X.shape[0]
    zv = np.concatenate([[0.0], np.cumsum(t[1:S])])

    def point(i):                                   # hit on surface i (1-based) and its unit normal
        x, y = X[i - 1], Y[i - 1]
        if not math.isfinite(R[i]):
            z = torch.zeros_like(x); nr = torch.stack([z, z, torch.ones_like(x)])
        else:
            c = 1.0 / R[i]
            root = torch.sqrt(1.0 - c * c * (x * x + y * y))
            z = c * (x * x + y * y) / (1.0 + root)
            nr = torch.stack([-c * x / root, -c * y / root, torch.ones_like(x)])
            nr = nr / nr.norm(dim=0)
        return torch.stack([x, y, zv[i - 1] + z]), nr

    def unit(d):                                    # every ray travels towards +z (see tests/test_oracle_physics.py)
        return d * (torch.sign(d[2]) / d.norm(dim=0))

    cur, nr = point(1)
    d1 = torch.tensor([v, u, 1.0], dtype=X.dtype, device=X.device)[:, None].expand(3, X.shape[1])
    d1 = d1 / d1.norm(dim=0)
    worst = 0.0
    for i in range(1, S):
        nxt, nr_next = point(i + 1)
        d2 = unit(nxt - cur)
        res = n[i - 1] * torch.linalg.cross(d1, nr, dim=0) - n[i] * torch.linalg.cross(d2, nr, dim=0)
        worst = max(worst, float(torch.nan_to_num(res.norm(dim=0)).max()))
        d1, cur, nr = d2, nxt, nr_next
    return worst


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_device_pointer_path_full_size_properties(oracle_engine, policy):
    """BASELINE config 2 at FULL size (3 fields x 3 index columns x 1024 x 1024, S = 12)
    through the device-pointer ABI (torch tensors).  Too large for the oracle, so check
    size-independent properties: (1) x -> -x mirror symmetry, exact; (2) a strided sample of
    rays re-traced by the oracle from explicit lists, bit-exact; (3) status histogram sanity;
    (4) every one of the 9.4e6 rays obeys the vector law of refraction at every surface to 1e-11."""
    import ctypes as C
    import torch
    from opticalraytracing_jl_amd import _capi
    eng = ort.default_engine()
    k = 1024
    pres, bundles, axes = _dg_bundles(oracle_engine, k)
    nb, rpb = len(bundles), k * k
    N, S = nb * rpb, pres.rows - 1
    dev = torch.device("cuda:0")
    d_axes = torch.from_numpy(axes).to(dev)
    xv = torch.empty((S, N), dtype=torch.float64, device=dev)
    yv = torch.empty((S, N), dtype=torch.float64, device=dev)
    st = torch.empty(N, dtype=torch.int32, device=dev)
    xf = torch.empty(N, dtype=torch.float64, device=dev); yf = torch.empty_like(xf)
    xs = torch.empty_like(xf); ys = torch.empty_like(xf)
    out = _capi.ort_grid_out_f64()
    out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
    out.xf, out.yf, out.xs, out.ys, out.status = xf.data_ptr(), yf.data_ptr(), xs.data_ptr(), ys.data_ptr(), st.data_ptr()
    sysd = eng.system(pres)
    barr = _capi.make_bundles(bundles)
    torch.cuda.synchronize()
    fast = policy == "fast"
    _capi.check(eng.ctx.lib.ort_trace_grid_f64(eng.ctx.h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                               C.byref(out), _capi.ORT_DEVICE_PTRS | (_capi.ORT_FAST_MATH if fast else 0)))
    eng.ctx.synchronize()
    # (0) summary == last / stop rows of the history
    assert torch.equal(xf, xv[-1]) or torch.equal(torch.nan_to_num(xf), torch.nan_to_num(xv[-1]))
    stop = bundles[0]["stop"]
    assert torch.equal(torch.nan_to_num(ys), torch.nan_to_num(yv[stop - 1]))
    # (1) mirror symmetry in x (V = 0): x(iy, ix) == -x(iy, k-1-ix), y equal — exact
    X = xv.view(S, nb, k, k); Y = yv.view(S, nb, k, k)
    assert torch.equal(torch.nan_to_num(X), torch.nan_to_num(-X.flip(-1)))
    assert torch.equal(torch.nan_to_num(Y), torch.nan_to_num(Y.flip(-1)))
    # (2) strided sample against the oracle
    idx = np.arange(0, N, 40009)
    sxv = xv[:, torch.from_numpy(idx).to(dev)].cpu().numpy(); syv = yv[:, torch.from_numpy(idx).to(dev)].cpu().numpy()
    sst = st[torch.from_numpy(idx).to(dev)].cpu().numpy()
    for b in range(nb):
        sel = (idx // rpb) == b
        j = idx[sel] % rpb
        bd = bundles[b]
        yy = axes[bd["yaxis_off"] + j // k]; xx = axes[bd["xaxis_off"] + j % k]
        sub = Prescription(pres.R[bd["system"]], pres.t[bd["system"]], pres.n[bd["system"]])
        u = np.full(j.size, math.tan(bd["U"])); v = np.zeros(j.size)
        ox, oy, os_ = oracle_engine.skew(sub, yy, xx, u, v, slopes=True, want_status=True)
        if fast:      # direction-cosine arithmetic: rounding-level differences only
            assert cm.rel_err(sxv[:, sel], ox, 1.0).max() <= 1e-12 and cm.rel_err(syv[:, sel], oy, 1.0).max() <= 1e-12
        else:         # the reference's operation sequence: bit-identical
            assert np.array_equal(sxv[:, sel], ox, equal_nan=True) and np.array_equal(syv[:, sel], oy, equal_nan=True)
        assert np.array_equal(sst[sel] & 0xffff, os_)
    # (3) every ray of this well-corrected system reaches the image; the square pupil's corners
    # fail the stop filter: the kept fraction is close to pi/4
    s = st.cpu().numpy()
    assert np.all((s & 0xffff) == S + 1)
    kept = 1.0 - np.count_nonzero(s >> 16) / N
    assert abs(kept - math.pi / 4) < 0.05
    # (4) first principles, EVERY ray of the launch (tests/test_oracle_physics.py does this to the oracle): consecutive
    # hit points obey the vector law of refraction at every surface — only the prescription's geometry is used
    worst = 0.0
    for b in range(nb):
        bd = bundles[b]
        sysi = bd["system"]
        worst = max(worst, _snell_residual_torch(pres.R[sysi], pres.t[sysi], pres.n[sysi], math.tan(bd["U"]), math.tan(bd["V"]),
                                                 xv[:, b * rpb:(b + 1) * rpb], yv[:, b * rpb:(b + 1) * rpb]))
    assert worst <= 1e-11, worst


def test_config4_zoom_sweep_many_systems(hip_engine, oracle_engine):
    """BASELINE config 4 shape at oracle size: 8 zoom positions x 5 index columns = 40 systems,
    5 fields each = 200 bundles of 24 x 24 rays; every bundle reads ITS system's table."""
    from opticalraytracing_jl_amd import api, workloads
    pres, bundles, axes = workloads.config4(api, 24, nzoom=8, engine=oracle_engine)
    assert pres.nsys == 40 and len(bundles) == 200
    g = hip_engine.grid(pres, bundles, axes, 24, 24)
    o = oracle_engine.grid(pres, bundles, axes, 24, 24)
    for key in ("xv", "yv", "xs", "ys"):
        assert np.array_equal(g[key], o[key], equal_nan=True), key
    assert np.array_equal(g["status"], o["status"])
    gf = hip_engine.full_trace_grid(pres, bundles[::17], axes, 24, 24)
    of = oracle_engine.full_trace_grid(pres, bundles[::17], axes, 24, 24)
    for a, b in zip(gf, of):
        assert a["count"] == b["count"] and np.array_equal(a["ex"], b["ex"])
        assert abs(a["rms"] - b["rms"]) <= TOL * b["rms"]


def test_config5_monte_carlo_instances(oracle_engine):
    """BASELINE config 5 shape: 500 perturbed instances (seed 12345) x 16 x 16 pupil on axis, one
    launch; Float64 bit-exact against the oracle and the Float32 build within 1e-4."""
    import ctypes as C
    from opticalraytracing_jl_amd import _capi, workloads
    eng = ort.default_engine()
    mats = workloads.config5(None, ninst=500)
    ninst, k = mats.shape[0], 16
    R = np.concatenate([mats[:, :, 0], np.full((ninst, 1), math.inf)], axis=1)
    t = np.concatenate([mats[:, :, 1], np.zeros((ninst, 1))], axis=1); t[:, -2] = 57.8
    n = np.concatenate([mats[:, :, 2], np.ones((ninst, 1))], axis=1)
    pres = Prescription(R, t, n)
    ax = ort.linrange(-12.0, 12.0, k)
    axes = np.concatenate([ax, ax])
    bundles = [dict(system=i, stop=6, U=0.0, V=0.0, a_stop=10.229, yaxis_off=0, xaxis_off=k) for i in range(ninst)]
    g = eng.grid(pres, bundles, axes, k, k)
    o = oracle_engine.grid(pres, bundles, axes, k, k)
    assert np.array_equal(g["xv"], o["xv"], equal_nan=True) and np.array_equal(g["yv"], o["yv"], equal_nan=True)
    assert np.array_equal(g["status"], o["status"])
    # Float32 build of the same launch
    N, S = ninst * k * k, pres.rows - 1
    xv = np.empty((S, N), dtype=np.float32); yv = np.empty((S, N), dtype=np.float32)
    out = _capi.ort_grid_out_f32()
    out.xv, out.yv, out.ld = xv.ctypes.data, yv.ctypes.data, N
    a32 = axes.astype(np.float32)
    sysd = eng.system(pres)
    _capi.check(eng.ctx.lib.ort_trace_grid_f32(eng.ctx.h, sysd.h, ninst, _capi.make_bundles(bundles), a32.ctypes.data,
                                               a32.size, k, k, C.byref(out), _capi.ORT_FAST_MATH))
    assert cm.rel_err(xv, o["xv"], 1.0).max() <= 1e-4 and cm.rel_err(yv, o["yv"], 1.0).max() <= 1e-4


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_config3_full_size_aspheric_properties(oracle_engine, policy):
    """BASELINE config 3 at FULL size: 3 x 3 bundles x 2048 x 2048 pupil (37.7 M rays, S = 12),
    4 aspheric surfaces, full_trace pipeline with stop-filter compaction, device pointers, in both policies.
    Properties: survivors ~ pi/4 of the pupil square; compacted first half equals the filtered
    summary trace in ray order (checked on one bundle against a second, independent launch);
    mirror halves exact; a strided sample of rays equals the oracle to 1e-12; FAST: the survivor count of every
    bundle equals the reference-sequence policy's (the stop filter is decided identically, PupilSampling.jl:132)."""
    import ctypes as C
    import torch
    from opticalraytracing_jl_amd import _capi, api, workloads
    eng = _engine(policy)
    pf = _capi.ORT_FAST_MATH if policy == "fast" else 0
    k = 2048
    pres, bundles, axes = workloads.config3(api, k, engine=oracle_engine)
    nb, rpb = len(bundles), k * k
    dev = torch.device("cuda:0")
    d_axes = torch.from_numpy(axes).to(dev)
    cap = 2 * rpb
    ex = torch.empty((nb, cap), dtype=torch.float64, device=dev); ey = torch.empty_like(ex)
    rho = torch.empty_like(ex); th = torch.empty_like(ex)
    cnt = torch.empty(nb, dtype=torch.int64, device=dev); rms = torch.empty(nb, dtype=torch.float64, device=dev)
    sysd = eng.system(pres)
    barr = _capi.make_bundles(bundles)
    torch.cuda.synchronize()
    _capi.check(eng.ctx.lib.ort_full_trace_f64(eng.ctx.h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                               ex.data_ptr(), ey.data_ptr(), rho.data_ptr(), th.data_ptr(),
                                               cnt.data_ptr(), rms.data_ptr(), _capi.ORT_DEVICE_PTRS | pf))
    eng.ctx.synchronize()
    c = cnt.cpu().numpy(); r = rms.cpu().numpy()
    assert np.all(c % 2 == 0) and np.all(np.abs(c / 2 / rpb - math.pi / 4) < 0.10)
    assert np.all(np.isfinite(r)) and np.all(r > 0)
    if policy == "fast":      # counts and RMS against the reference-sequence policy of the same call (statistics-only route)
        c2 = torch.empty_like(cnt); r2 = torch.empty_like(rms)
        _capi.check(eng.ctx.lib.ort_full_trace_f64(eng.ctx.h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                                   None, None, None, None, c2.data_ptr(), r2.data_ptr(), _capi.ORT_DEVICE_PTRS))
        eng.ctx.synchronize()
        assert torch.equal(c2, cnt) and float(((r2 - rms).abs() / r2).max()) <= 1e-10
    # bundle 4: summary trace + host filter == compacted first half, in order; mirror exact
    b = 4
    xf = torch.empty(rpb, dtype=torch.float64, device=dev); yf = torch.empty_like(xf)
    xs = torch.empty_like(xf); ys = torch.empty_like(xf); st = torch.empty(rpb, dtype=torch.int32, device=dev)
    out = _capi.ort_grid_out_f64()
    out.xf, out.yf, out.xs, out.ys, out.status = xf.data_ptr(), yf.data_ptr(), xs.data_ptr(), ys.data_ptr(), st.data_ptr()
    one = _capi.make_bundles([bundles[b]])
    _capi.check(eng.ctx.lib.ort_trace_grid_f64(eng.ctx.h, sysd.h, 1, one, d_axes.data_ptr(), axes.size, k, k,
                                               C.byref(out), _capi.ORT_DEVICE_PTRS | pf))
    eng.ctx.synchronize()
    keep = (st >> 16) == 0
    m = int(c[b]) // 2
    assert int(keep.sum()) == m
    assert torch.equal(ex[b, :m], xf[keep]) and torch.equal(ey[b, :m], yf[keep] - bundles[b]["hprime"])
    assert torch.equal(ex[b, m:2 * m], -ex[b, :m]) and torch.equal(ey[b, m:2 * m], ey[b, :m])
    assert torch.equal(rho[b, m:2 * m], rho[b, :m]) and float(rho[b, :m].max()) == 1.0
    # RMS from the returned vectors (float64 on the device) == returned RMS
    X, Y = ex[b, :2 * m], ey[b, :2 * m]
    sig = torch.sqrt((((X - X.mean()) ** 2).sum() + ((Y - Y.mean()) ** 2).sum()) / (2 * m))
    assert abs(float(sig) - r[b]) <= 1e-12 * r[b]
    # strided sample vs the oracle (aspheric rows: analytic vs complex-step derivative)
    idx = np.arange(0, rpb, 9973)
    bd = bundles[b]
    yy = axes[bd["yaxis_off"] + idx // k]; xx = axes[bd["xaxis_off"] + idx % k]
    sub = Prescription(pres.R[bd["system"]], pres.t[bd["system"]], pres.n[bd["system"]], pres.K[bd["system"]],
                       pres.coef[bd["system"]][None])
    u = np.full(idx.size, math.tan(bd["U"])); v = np.zeros(idx.size)
    ox, oy, os_ = oracle_engine.skew(sub, yy, xx, u, v, slopes=True, want_status=True)
    ti = torch.from_numpy(idx).to(dev)
    assert cm.rel_err(xf[ti].cpu().numpy(), ox[-1], 1.0).max() <= 1e-12
    assert cm.rel_err(yf[ti].cpu().numpy(), oy[-1], 1.0).max() <= 1e-12
    assert np.array_equal(st[ti].cpu().numpy() & 0xffff, os_)


def test_device_aiming_matches_host_driven(hip_engine, oracle_engine):
    """ort_aim_f64 (one thread per (system, field) running the Newton loops on the device)
    == the same loops driven from the host over the device meridional kernel (identical device
    arithmetic -> identical iterates), and == the oracle-driven aiming to the Newton tolerance."""
    from opticalraytracing_jl_amd import api, workloads
    systems = [ort.solve(workloads.double_gauss(line, g), cm.DG_A, cm.DG_H, engine=oracle_engine)
               for line, g in ((0, 0.0), (1, 0.4), (2, -0.7))]
    M4, coef = cm.double_gauss_aspheric()
    systems.append(ort.solve(ort.Layout(M4[:, 0], M4[:, 1], M4[:, 2], M4[:, 3], [c for c in coef]), cm.DG_A, cm.DG_H,
                             engine=oracle_engine))
    fields = (0.0, 0.7, 1.0)
    dev = ort.full_trace_aim_batch(systems, fields, engine=hip_engine)
    fdev = ort.full_trace_aim_batch(systems, fields, engine=ort.HipEngine(0, fast_math=True))
    for si, s in enumerate(systems):
        for fi, H in enumerate(fields):
            host = ort.full_trace_aim(s.layout, s, H, engine=hip_engine)
            orc = ort.full_trace_aim(s.layout, s, H, engine=oracle_engine)
            d = dev[si][fi]
            # U, y_EP: same device arithmetic on both routes (the reference's meridional sequence) -> identical.  tan(U) is
            # ocml in the kernel and libm on the host route, so h' may differ in the last ulp and the edge-ray Newton
            # (stopped at |residual| <= sqrt(eps)) may stop on a neighbouring iterate.
            assert d.U == host.U and d.y_EP == host.y_EP, (si, H)
            assert abs(d.hprime - host.hprime) <= 1e-14 * max(1.0, abs(host.hprime))
            # ORT_FAST_MATH: plain prescriptions are aimed through the trig-free trace (mer_plain_trace_to) — the same function
            # of the launch data to rounding, the same loops, iterates a few 1e-9 apart at most
            f = fdev[si][fi]
            assert abs(f.U - host.U) <= 1e-13 * max(1.0, abs(host.U)) and abs(f.y_EP - host.y_EP) <= 1e-9, (si, H)
            assert abs(f.y1 - host.y1) <= 1e-7 and abs(f.y2 - host.y2) <= 1e-7 and f.stop == host.stop, (si, H)
            for key in ("y1", "y2"):
                assert abs(getattr(d, key) - getattr(host, key)) <= 1e-7, (si, H, key)
            for key in ("U", "y_EP", "hprime"):
                assert abs(getattr(d, key) - getattr(orc, key)) <= 1e-9 * max(1.0, abs(getattr(orc, key))), (si, H, key)
            for key in ("y1", "y2"):
                assert abs(getattr(d, key) - getattr(orc, key)) <= 1e-7, (si, H, key)
            assert d.stop == host.stop and d.a_stop == host.a_stop and d.focus == host.focus


def test_edge_rule_only_moves_the_two_edge_rays(hip_engine, oracle_engine):
    """The edge-ray rule (an end point found outside the stop's edge steps to sqrt(eps) inside) is FITTED to the
    reference's published Tessar figure, not a restatement of its Optim.BFGS search (src/PupilSampling.jl:67-83,
    parity unpinned).  What it can change against a search left where it ends (ORT_AIM_EDGE_AS_FOUND): y1 / y2 by
    <= 2 sqrt(eps)-ish, hence only whether the x = 0 rays of the first and the last pupil row pass r > a_stop (:132).
    Checked on perturbed Double-Gauss instances, the Cooke triplet and the Tessar, three fields each; the traced grid is
    the reference-sequence policy and is also checked against the oracle's status."""
    from opticalraytracing_jl_amd import api, workloads
    mats = list(workloads.config5(None, ninst=24, seed=777)) + [cm.cooke(), cm.tessar()]
    aps = [(cm.DG_A, cm.DG_H)] * 24 + [(cm.COOKE_A, cm.COOKE_H), (cm.TESSAR_A, cm.TESSAR_H)]
    fields = (0.0, 0.7, 1.0)
    k, k2 = 32, 16
    moved = diff_rays = 0
    for M, (A, Hh) in zip(mats, aps):
        s = ort.solve(M, A, Hh, engine=oracle_engine)
        pf, lf, _ = api._as_layout(s.layout)
        pr, lr, _ = api._as_layout(api.reversed_layout(s.layout, s))
        specs = [dict(system=0, stop=s.stop, layout_fwd=lf, layout_rev=lr, H=H, y_marg=s.marginal.y[0], a_stop=s.a[s.stop - 1],
                      chief_y_end=s.chief.y[-1], chief_u_end=s.chief.u[-1], f=s.f) for H in fields]
        on = hip_engine.aim(pf, pr, specs)
        off = hip_engine.aim(pf, pr, specs, edge_as_found=True)
        foc = s.marginal.z[-1] - s.marginal.z[-2]
        pres = api.extended_prescription(s.layout, foc)
        masks = []
        for outs in (on, off):
            axes, bundles, o = [], [], 0
            for a in outs:
                axes += [ort.linrange(a["y1"], a["y2"], k), ort.linrange(0.0, a["y_EP"], k2)]
                bundles.append(dict(system=0, stop=s.stop, U=a["U"], V=0.0, a_stop=abs(s.a[s.stop - 1]), hprime=a["hprime"],
                                    yaxis_off=o, xaxis_off=o + k))
                o += k + k2
            axes = np.concatenate(axes)
            g = hip_engine.grid(pres, bundles, axes, k, k2, history=False)
            og = oracle_engine.grid(pres, bundles, axes, k, k2, history=False)
            assert np.array_equal(g["status"], og["status"])
            masks.append((g["status"] == pres.rows).reshape(len(fields), k, k2))        # every surface hit, not stopped
        for a, b in zip(on, off):
            for key in ("U", "y_EP", "hprime"):
                assert a[key] == b[key]                                                     # the rule touches y1, y2 only
            for key in ("y1", "y2"):
                assert abs(a[key] - b[key]) <= 1e-6
                moved += a[key] != b[key]
        d = masks[0] != masks[1]
        allowed = np.zeros_like(d); allowed[:, 0, 0] = True; allowed[:, k - 1, 0] = True
        assert not (d & ~allowed).any()
        assert (masks[0] | ~d).all()                       # where they differ, the rule is the one that keeps the ray
        diff_rays += int(d.sum())
    assert moved > 0 and diff_rays > 0                     # the rule did act on this set


def test_full_trace_batch_end_to_end(hip_engine, oracle_engine):
    """full_trace for (system x field) batches: aiming kernel + full_trace pipeline on the GPU
    == per-call full_trace through the oracle (aiming tolerance sqrt(eps) -> 1e-7 on errors)."""
    from opticalraytracing_jl_amd import workloads
    sg = [ort.solve(workloads.double_gauss(l), cm.DG_A, cm.DG_H, engine=hip_engine) for l in (0, 1)]
    so = [ort.solve(workloads.double_gauss(l), cm.DG_A, cm.DG_H, engine=oracle_engine) for l in (0, 1)]
    errs = ort.full_trace_batch(sg, (0.0, 1.0), 48, engine=hip_engine)
    for si in range(2):
        for fi, H in enumerate((0.0, 1.0)):
            ref = ort.full_trace(so[si], H, 48, engine=oracle_engine)
            got = errs[si][fi]
            assert len(got.x) == len(ref.x)
            assert np.abs(got.x - ref.x).max() <= 1e-7 and np.abs(got.y - ref.y).max() <= 1e-7
            assert abs(got.RMS - ref.RMS) <= 1e-7
    with pytest.raises(ort.DomainError):
        ort.full_trace_batch(sg, (1.2,), 16, engine=hip_engine)


def test_first_order_and_seidel_batch(hip_engine, oracle_engine):
    """ort_first_order_f64 / ort_aberrations_f64 over 300 perturbed Double-Gauss instances (config 5's
    Monte-Carlo), the Cooke triplet with its dispersion vector and the Tessar, against the C oracle
    oracle/ort_oracle.c::orc_solve_aberrations (pinned to Smith's tables on the CPU): first-order
    properties, Seidel sums, the ten per-surface vectors of `aberrations` (src/SeidelAberrations.jl:25-34)
    and the four columns of `incidences` (src/RayTracing.jl:338-353)."""
    from opticalraytracing_jl_amd import _capi, workloads
    from oracle import cpu
    mats = workloads.config5(None, ninst=300)
    res = hip_engine.first_order(mats[:, :, 0], mats[:, :, 1], mats[:, :, 2], cm.DG_A, cm.DG_H)
    full = hip_engine.aberrations(mats[:, :, 0], mats[:, :, 1], mats[:, :, 2], cm.DG_A, cm.DG_H)
    scal = ("f", "EBFD", "EFFD", "N", "FOV", "H", "EP_D", "EP_t", "XP_D", "XP_t", "PN", "W040", "W131", "W222", "W220",
            "W311", "W020", "W111", "W220P")

    def close(got, ref, what):
        assert np.all(np.abs(np.asarray(got) - np.asarray(ref)) <= 1e-12 * np.maximum(1.0, np.abs(ref))), what

    for i in (0, 7, 123, 299):
        o = cpu.solve_aberrations(mats[i], cm.DG_A, cm.DG_H)
        assert res[i]["stop"] == o["stop"] == full["stop"][i] and res[i]["k"] == o["k"]
        for key in scal:
            close(res[i][key], o[key], (i, key)); close(full[key][i], o[key], (i, key))
        close(res[i]["y_marg"], o["marginal_y"][0], i); close(res[i]["chief_y_end"], o["chief_y"][-1], i)
        close(res[i]["nu_end"], o["marginal_nu"][-1], i)
        for key in _capi.ORT_SURF_NAMES + _capi.ORT_INC_NAMES:
            close(full[key][i], o[key], (i, key))
    for surf, a, h, dn in ((cm.cooke(), cm.COOKE_A, cm.COOKE_H, cm.COOKE_DN), (cm.tessar(), cm.TESSAR_A, cm.TESSAR_H, None)):
        r = hip_engine.aberrations(surf[:, 0], surf[:, 1], surf[:, 2], a, h, dn=dn)
        o = cpu.solve_aberrations(surf, a, h, dn=dn)
        for key in scal:
            close(r[key][0], o[key], key)
        for key in _capi.ORT_SURF_NAMES + _capi.ORT_INC_NAMES:
            close(r[key][0], o[key], key)
    # the reference's documentation prints the third-order spot size of the Tessar (spot_diagram(W), H = 0): 0.12132
    # (tests/test_oracle_reference_vectors.py::test_tessar_polynomial_spot_diagram_figure) — from the DEVICE's W040 and n'u'
    t = cm.tessar()
    ra = hip_engine.aberrations(t[:, 0], t[:, 1], t[:, 2], cm.TESSAR_A, cm.TESSAR_H)
    fo = hip_engine.first_order(t[:, 0], t[:, 1], t[:, 2], cm.TESSAR_A, cm.TESSAR_H)[0]
    xx = np.linspace(-1.0, 1.0, 64)
    ee = 4.0 * ra["W040"][0] * xx ** 3 * 587.5618e-6 / fo["nu_end"]
    assert f"{math.sqrt(2.0 * np.sum((ee - ee.sum() / 64) ** 2) / 64):.5f}" == "0.12132"
    r = hip_engine.first_order(cm.cooke()[:, 0], cm.cooke()[:, 1], cm.cooke()[:, 2], cm.COOKE_A, cm.COOKE_H, dn=cm.COOKE_DN)[0]
    assert abs(r["f"] - 101.181) < 1e-3 and abs(r["EBFD"] - 77.405) < 1e-3 and r["stop"] == 5   # test/runtests.jl:53-60
    # a finite, non-zero last thickness: Lens() keeps the last row and the reference needs `rows` semi-diameters
    # (DimensionMismatch for rows-1) -> rejected instead of reading the next system's first aperture
    bad = cm.cooke(); bad[-1, 1] = 3.0
    with pytest.raises(_capi.OrtError) as e:
        hip_engine.first_order(bad[:, 0], bad[:, 1], bad[:, 2], cm.COOKE_A, cm.COOKE_H)
    assert e.value.code == -1 and "last thickness" in str(e.value)


def test_fan_tsa_sa_caustic_on_device(hip_engine, oracle_engine):
    """SURVEY §8f #4 on the device: `TSA` (src/SeidelAberrations.jl:116-137), the `SA` fit (:139-146) and the
    caustic ray set (ext/MakieExtension.jl:364-381) through ort_fan_f64, against the oracle engine's restatement
    of the same lines; the reference's own check `abs(SA(TSA(...)..., 9)[1] / W040 - 1) < 0.05`
    (test/runtests.jl:277-278) with W040 from the device Seidel kernel; and the batched entry fed by the device
    aiming kernel (y_EP, XP_t) for 40 perturbed Double-Gauss instances in one launch."""
    from opticalraytracing_jl_amd import analysis as an, batch, workloads
    for surf, a, h in ((cm.cooke(), cm.COOKE_A, cm.COOKE_H), (cm.tessar(), cm.TESSAR_A, cm.TESSAR_H)):
        sg = ort.solve(surf.copy(), a, h, engine=hip_engine)
        so = ort.solve(surf.copy(), a, h, engine=oracle_engine)
        yg, eg = an.TSA(surf, sg, engine=hip_engine)
        yo, eo = an.TSA(surf, so, engine=oracle_engine)
        assert yg.shape == yo.shape == (ort.api.K_RAYS,)
        assert cm.rel_err(yg, yo, 1.0).max() <= TOL and cm.rel_err(eg, eo, 1e-3).max() <= 1e-8    # eps ~ 1e-2 mm: aiming atol 1.5e-8
        cg, co = an.caustic_rays(surf, sg, 24, engine=hip_engine), an.caustic_rays(surf, so, 24, engine=oracle_engine)
        assert cg["to_paraxial_plane"] == co["to_paraxial_plane"] and abs(cg["zf"] - co["zf"]) <= 1e-8
        for key in ("y0", "yf", "y_surf", "z_surf"):
            assert np.abs(cg[key] - co[key]).max() <= 1e-8, key
    surf = cm.cooke()
    sg = ort.solve(surf.copy(), cm.COOKE_A, cm.COOKE_H, engine=hip_engine)
    # the reference's own check (test/runtests.jl:277-278): abs(B1 / W040 - 1) < 0.05 with ITS constant W040 = -0.186575,
    # Smith's third-order transverse spherical sum in mm (test/runtests.jl:164-184) — B1, the cubic coefficient of the
    # fit to the device's TSA fan, is a transverse quantity in mm too
    B1 = an.SA(*an.TSA(surf, sg, engine=hip_engine), 9)[0]
    assert abs(B1 / -0.186575 - 1) < 0.05, B1
    # ... and the device's Seidel sum says the same in waves: TSC = 4 W040 lambda / n'u'
    W040 = hip_engine.aberrations(surf[:, 0], surf[:, 1], surf[:, 2], cm.COOKE_A, cm.COOKE_H)["W040"][0]
    assert abs(4 * W040 * 587.5618e-6 / sg.marginal.nu[-1] / -0.186575 - 1) < 1e-3
    # batched: device aiming (y_EP, XP_t) -> one fan launch over 40 instances, vs the oracle fan per instance
    mats = workloads.config5(None, ninst=40)
    fo = batch.first_order_arrays(hip_engine, mats, cm.DG_A, cm.DG_H)
    aims = batch.aim_instances(mats, cm.DG_A, cm.DG_H, (0.0,), engine=hip_engine)
    pres = Prescription(mats[:, :, 0], mats[:, :, 1], mats[:, :, 2])
    specs = [dict(system=i, layout_mode=0, y_marg=float(aims["y_EP"][i, 0]), XP_t=float(aims["XP_t"][i, 0]), BFD=float(fo["BFD"][i]))
             for i in range(40)]
    yg, eg = hip_engine.fan(pres, specs, 32)
    yo, eo = oracle_engine.fan(pres, specs, 32)
    assert cm.rel_err(yg, yo, 1.0).max() <= TOL and np.abs(eg - eo).max() <= 1e-10
    # XP_t of the device aiming == the host-driven trace_chief_ray (RayTracing.jl:294)
    s0 = ort.solve(mats[0].copy(), cm.DG_A, cm.DG_H, engine=oracle_engine)
    rc = ort.trace_chief_ray(mats[0], s0, engine=oracle_engine)
    assert abs((rc.z[-1] - rc.z[-2]) - aims["XP_t"][0, 0]) <= 1e-6 * abs(aims["XP_t"][0, 0])


def test_native_rccl_allgather_single_rank():
    """ort_comm_* / ort_allgather_hits_f64 (librccl loaded lazily): a 1-rank communicator on the one
    GPU of the test box — the all-gather must reproduce the slabs (multi-rank correctness of the
    shard order is covered by the gloo test and is RCCL's contract)."""
    import torch
    from opticalraytracing_jl_amd import dist as odist
    eng = ort.default_engine()
    comm = odist.RcclComm(eng, 1, 0, odist.RcclComm.unique_id())
    xf = torch.randn(100003, dtype=torch.float64, device="cuda:0")
    yf = torch.randn(100003, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    gx, gy = comm.allgather_hits(xf, yf)
    assert torch.equal(gx, xf) and torch.equal(gy, yf)
    comm.close()


def test_context_destroyed_ahead_of_its_communicator():
    """A garbage collector may destroy a context before the communicator created on it: ort_ctx_destroy
    drains the communicator's stream and detaches it, the communicator's entry points then fail with a message instead of
    reaching into freed memory, and ort_comm_destroy still releases it."""
    import torch
    from opticalraytracing_jl_amd import _capi, dist as odist
    eng = ort.HipEngine(0)
    comm = odist.RcclComm(eng, 1, 0, odist.RcclComm.unique_id())
    hits = torch.randn((2, 50001), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    g = comm.allgather_hits_packed(hits, wait=False)             # a collective still in flight on the communicator's stream
    lib = eng.ctx.lib
    eng._systems.clear()
    eng.ctx.close()                                               # the context goes first
    assert torch.equal(g[0], hits)                                # ... after draining the communicator's stream
    rc = lib.ort_comm_synchronize(comm.h)
    assert rc == -1 and b"null context" in lib.ort_last_error()
    assert lib.ort_comm_size(comm.h) == 1                         # plain queries still answer
    assert lib.ort_comm_destroy(comm.h) == 0
    comm.h = None


def test_error_codes(hip_engine):
    """Error behaviour at the boundary: bad arguments -> ORT_EINVAL with a message, |H| > 1 ->
    ORT_EDOMAIN (the reference's DomainError, src/PupilSampling.jl:88-89); no exceptions cross the ABI."""
    from opticalraytracing_jl_amd import _capi
    pres = Prescription.from_matrix(cm.cooke())
    with pytest.raises(_capi.OrtError) as e:
        hip_engine.skew(pres, [1.0], [0.0], [0.0], [0.0], isys=3)
    assert e.value.code == -1 and "out of range" in str(e.value)
    with pytest.raises(_capi.OrtError) as e:
        hip_engine.grid(pres, [dict(system=0, stop=99, U=0.0, V=0.0, yaxis_off=0, xaxis_off=4)], np.zeros(8), 4, 4)
    assert e.value.code == -1
    with pytest.raises(_capi.OrtError) as e:
        hip_engine.grid(pres, [dict(system=0, stop=0, U=0.0, V=0.0, yaxis_off=0, xaxis_off=6)], np.zeros(8), 4, 4)
    assert e.value.code == -1 and "axis" in str(e.value)
    s = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=hip_engine)
    spec = dict(system=0, stop=s.stop, H=1.5, y_marg=s.marginal.y[0], a_stop=s.a[s.stop - 1],
                chief_y_end=s.chief.y[-1], chief_u_end=s.chief.u[-1], f=s.f)
    with pytest.raises(_capi.OrtError) as e:
        hip_engine.aim(pres, pres, [spec])
    assert e.value.code == _capi.ORT_EDOMAIN
    big = np.zeros((70, 3)); big[:, 2] = 1.0
    with pytest.raises(_capi.OrtError):
        hip_engine.skew(Prescription.from_matrix(big), [0.0], [0.0], [0.0], [0.0])      # rows > ORT_MAX_ROWS
    # maximum sizes: a bundle of 2^32 rays is refused before anything is allocated (ray indices inside a bundle are 31-bit)
    import ctypes as C
    axes = np.zeros(131072)
    out = _capi.ort_grid_out_f64()
    st = np.zeros(4, dtype=np.int32)
    out.status = st.ctypes.data                                   # never written: the size check comes first
    barr = _capi.make_bundles([dict(system=0, stop=5, U=0.0, V=0.0, yaxis_off=0, xaxis_off=65536)])
    rc = hip_engine.ctx.lib.ort_trace_grid_f64(hip_engine.ctx.h, hip_engine.system(pres).h, 1, barr, _capi.ptr(axes), axes.size,
                                               65536, 65536, C.byref(out), 0)
    assert rc == -1 and b"too large" in hip_engine.ctx.lib.ort_last_error()


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


# FAST policy (tests below; measured tables: scripts/fast_attribution.py).  The FAST forms differ from the reference
# sequence by rounding only.  Wherever the two could DECIDE differently — a ray within 1e-9 (normalised) of a miss / TIR /
# equator / stop-edge branch, or where the reference's formulas stop being the geometry (far-cap hits, backward
# directions, polynomial rows outside their conic) — the kernel retraces the wave with the reference sequence itself
# (ort_device.hpp, `odd`).  So: status identical on EVERY ray, no allowance; coordinates compared on EVERY ray, no
# exclusion zone.  A rounding difference is amplified by the conditioning of the ray's path, which the oracle measures on
# itself (sens = largest relative change of any coordinate under a 1e-13 relative perturbation of the launch data): the
# bar is 1e-10, or 100 x that response for the rare ill-conditioned ray (counted and bounded below).
FAST_TOL = 1e-10          # BASELINE north_star: 1e-10 relative
FAST_AMP = 100.0          # ... or 100 x the oracle's own response to the 1e-13 perturbation, whichever is larger


def _deviation(ax, ay, bx, by):
    s = np.maximum(1.0, np.maximum(np.nanmax(np.abs(bx), axis=0, initial=0.0), np.nanmax(np.abs(by), axis=0, initial=0.0)))
    d = np.maximum(np.nanmax(np.abs(ax - bx), axis=0, initial=0.0), np.nanmax(np.abs(ay - by), axis=0, initial=0.0)) / s
    pat = (np.isnan(ax) != np.isnan(bx)).any(axis=0) | (np.isnan(ay) != np.isnan(by)).any(axis=0)
    return np.where(pat, np.inf, d)


def _fast_attribution(fast, oracle_engine, pres, y, x, u, v, tag, worst=None):
    """One prescription: FAST vs oracle on every ray.  Asserts (1) status identical on every ray, (2) NaN patterns
    identical and coordinates within max(FAST_TOL, FAST_AMP * sens) on every ray — far-cap hits, post-TIR paths, near-
    boundary rays and huge coordinates included.  Returns (rays, rays that needed the amplified bar, far-cap rays)."""
    ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    d = 1e-13
    px, py = oracle_engine.skew(pres, y * (1 + d), x * (1 - d), u * (1 + d), v * (1 - d), slopes=True)
    sens = _deviation(px, py, ox, oy)
    fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
    mg = oracle_engine.skew_margins(pres, y, x, u, v)
    err = _deviation(fx, fy, ox, oy)
    flip = fs != os_
    assert not flip.any(), (tag, "status", int(flip.sum()), mg[flip][:3])
    bad = ~(err <= np.maximum(FAST_TOL, FAST_AMP * sens))
    assert not bad.any(), (tag, "coordinates", int(bad.sum()), float(err[bad].max()), float(sens[bad].max()), mg[bad][:3])
    if worst is not None:                                         # the caller's record of the largest finite deviation seen
        worst[0] = max(worst[0], float(np.max(err[np.isfinite(err)], initial=0.0)))
    return y.size, int((err > FAST_TOL).sum()), int((mg[:, 3] > 0).sum())


def test_polynomial_forms_on_device(hip_engine, oracle_engine):
    """Every polynomial arm of the FAST policy on the device, and the build each system is launched with: BASELINE config
    3's rows (conic + 4-term even form: the even-asphere build), a 10th-order even asphere (6-term even form, same build),
    odd coefficients (general forms of <= 8 and <= 12 coefficients: the full build), an even asphere beside a vertex-form
    sphere (|R| > 1e3: full build) and a plane carrying a polynomial (Schmidt-like; sag = 0 WITHOUT p(y),
    src/PupilSampling.jl:12, tilt = p' only, :18).  Status identical on every ray, coordinates <= 1e-11 (both policies:
    the reference takes p' by a complex step, src/RayTracing.jl:103, the device analytically)."""
    fast = ort.HipEngine(0, fast_math=True)
    M4, coef = cm.double_gauss_aspheric()
    ext = np.vstack([M4, [math.inf, 0.0, 1.0, 0.0]]); ext[-2, 1] = 57.8
    rng = np.random.default_rng(3)
    m = 6000
    y = rng.uniform(-14, 14, m); x = rng.uniform(-14, 14, m)
    u = np.tan(rng.uniform(-0.1, 0.1, m)); v = np.tan(rng.uniform(-0.1, 0.1, m))
    c10 = np.zeros((coef.shape[0], 11)); c10[:, :7] = coef; c10[1, 8] = 3e-14; c10[5, 10] = -2e-16
    c_odd = coef.copy(); c_odd[7, 3] = 4e-6; c_odd[1, 5] = -3e-9
    c_odd12 = np.zeros((coef.shape[0], 12)); c_odd12[:, :7] = c_odd; c_odd12[11, 11] = 1e-18; c_odd12[5, 9] = 2e-15
    weak = ext.copy(); weak[3, 0] = 2500.0
    S = np.array([[math.inf, 0.0, 1.0, 0.0], [math.inf, 4.0, 1.52, 0.0], [-80.0, 30.0, 1.0, 0.0], [math.inf, 0.0, 1.0, 0.0]])
    cs = np.zeros((4, 7)); cs[1, 2] = 1e-4; cs[1, 4] = -3e-7
    cases = [("config3", ext, coef), ("even6", ext, c10), ("odd8", ext, c_odd), ("odd12", ext, c_odd12), ("weak", weak, coef)]
    for tag, M, c in cases:
        c = np.vstack([c, np.zeros((1, c.shape[1]))])
        pres = Prescription(M[:, 0], M[:, 1], M[:, 2], M[:, 3], c[None])
        ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        assert (os_ == M.shape[0]).mean() > 0.3, tag                  # a good share of the bundle reaches the image
        for eng in (hip_engine, fast):
            gx, gy, gs = eng.skew(pres, y, x, u, v, slopes=True, want_status=True)
            assert np.array_equal(gs, os_), tag
            assert np.array_equal(np.isnan(gx), np.isnan(ox)) and np.array_equal(np.isnan(gy), np.isnan(oy)), tag
            assert max(cm.rel_err(gx, ox, 1.0).max(), cm.rel_err(gy, oy, 1.0).max()) <= 1e-11, (tag, eng is fast)
    pres = Prescription(S[:, 0], S[:, 1], S[:, 2], S[:, 3], cs[None])
    ys, xs = y * 0.5, x * 0.5
    ox, oy, os_ = oracle_engine.skew(pres, ys, xs, u * 0.5, v * 0.5, slopes=True, want_status=True)
    for eng in (hip_engine, fast):
        gx, gy, gs = eng.skew(pres, ys, xs, u * 0.5, v * 0.5, slopes=True, want_status=True)
        assert np.array_equal(gs, os_)
        assert max(cm.rel_err(gx, ox, 1.0).max(), cm.rel_err(gy, oy, 1.0).max()) <= 1e-11


def test_random_systems_property(hip_engine, oracle_engine):
    """120 random prescriptions (2-14 rows; flat rows, both curvature signs, conics, polynomial
    terms, glass/air sequences that TIR and miss) x 1500 random skew rays each: the IEEE policy is
    BIT-IDENTICAL to the oracle on every ray incl. NaN patterns and status; the FAST policy has the same status on
    every ray and every ray's coordinates within the bar (see _fast_attribution)."""
    rng = np.random.default_rng(2024)
    fast = ort.HipEngine(0, fast_math=True)
    ntot = nill = nfar = 0
    for case in range(120):
        rows = int(rng.integers(2, 15))
        aspheric = (True, "even", False)[case % 3]
        R, t, n, K, coef = _random_system(rng, rows, aspheric)
        pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
        m = 1500
        y = rng.uniform(-6, 6, m); x = rng.uniform(-6, 6, m)
        u = np.tan(rng.uniform(-0.1, 0.1, m)); v = np.tan(rng.uniform(-0.1, 0.1, m))
        ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        assert np.array_equal(gs, os_), case
        if aspheric:      # analytic vs complex-step p'
            assert cm.rel_err(gx, ox, 1.0).max() <= 1e-11 and cm.rel_err(gy, oy, 1.0).max() <= 1e-11, case
        else:
            assert np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True), case
        a, b, c = _fast_attribution(fast, oracle_engine, pres, y, x, u, v, case)
        ntot += a; nill += b; nfar += c
    cm.report(f"test_random_systems_property (FAST): rays {ntot}, past 1e-10 (amplified bar) {nill}, far-cap rays {nfar}")
    assert nill <= 1e-4 * ntot, (nill, ntot)             # rays beyond 1e-10 (ill-conditioned paths, within 100 x sens): rare


def test_random_deep_systems_property(hip_engine, oracle_engine):
    """The random-prescription property test at DEPTH: 60 systems of 15 .. 63 rows (the suites above draw 2 .. 14) x 1000 rays.
    Reference-sequence policy: status identical on every ray; spherical / conic systems bit-identical; polynomial systems (analytic
    against complex-step p') within 1e-10 on every ray that stays inside 1e3 mm — random coefficients sized for a +-6 mm bundle
    explode once a ray is tens of mm off axis (metres of "sag"; coordinates of 1e4 .. 1e9 mm mean nothing and are compared for their
    NaN pattern only).  FAST policy (the reference sequence past ORT_FAST_MAX_SURFACES): status identical on every ray, the rays
    inside 1e3 mm within max(1e-10, 100 x their conditioning)."""
    rng = np.random.default_rng(4096)
    fast = ort.HipEngine(0, fast_math=True)
    ntot = nill = nwild = 0
    worst_poly = 0.0
    for case in range(60):
        rows = int(rng.integers(15, 64))
        aspheric = (True, "even", False)[case % 3]
        R, t, n, K, coef = _random_system(rng, rows, aspheric)
        pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
        m = 1000
        y = rng.uniform(-6, 6, m); x = rng.uniform(-6, 6, m)
        u = np.tan(rng.uniform(-0.1, 0.1, m)); v = np.tan(rng.uniform(-0.1, 0.1, m))
        ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        assert np.array_equal(gs, os_), (case, rows)
        sane = (np.nanmax(np.abs(ox), axis=0, initial=0.0) < 1e3) & (np.nanmax(np.abs(oy), axis=0, initial=0.0) < 1e3)
        nwild += int((~sane).sum())
        if aspheric:
            assert np.array_equal(np.isnan(gx), np.isnan(ox)) and np.array_equal(np.isnan(gy), np.isnan(oy)), (case, rows)
            if sane.any():
                dp = max(cm.rel_err(gx[:, sane], ox[:, sane], 1.0).max(), cm.rel_err(gy[:, sane], oy[:, sane], 1.0).max())
                worst_poly = max(worst_poly, float(dp))
                assert dp <= 1e-10, (case, rows, dp)
        else:
            assert np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True), (case, rows)
        fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
        assert np.array_equal(fs, os_), (case, rows, "FAST status")
        if rows - 1 > 48:
            assert np.array_equal(fx, gx, equal_nan=True) and np.array_equal(fy, gy, equal_nan=True), (case, rows)
        elif sane.any():
            d13 = 1e-13
            px, py = oracle_engine.skew(pres, y * (1 + d13), x * (1 - d13), u * (1 + d13), v * (1 - d13), slopes=True)
            sens = _deviation(px, py, ox, oy)
            err = _deviation(fx, fy, ox, oy)
            bad = sane & ~(err <= np.maximum(FAST_TOL, FAST_AMP * sens))
            assert not bad.any(), (case, rows, int(bad.sum()), float(err[bad].max()), float(sens[bad].max()))
            nill += int((sane & (err > FAST_TOL)).sum())
        ntot += m
    cm.report(f"test_random_deep_systems_property (15 .. 63 rows): rays {ntot}, beyond 1e3 mm {nwild}, FAST rays past 1e-10 (amplified bar) {nill}, "
              f"polynomial systems (reference-sequence policy, analytic vs complex-step p') worst {worst_poly:.2e}")
    assert nill <= 2e-3 * ntot, (nill, ntot)


def test_fast_policy_baseline_fixtures_need_no_amplified_bar(oracle_engine):
    """The amplified bar of _fast_attribution (100 x the oracle's own response to a 1e-13 perturbation) exists for the
    ill-conditioned rays of the adversarial random systems.  On the prescriptions of the BASELINE configs — Cooke triplet
    (config 1), Double-Gauss (configs 2, 4, 5), Double-Gauss with 4 aspheric surfaces (config 3) — it is never used:
    EVERY one of 20,001 random skew rays per system is within 1e-10 of the oracle (north_star's bar), status identical."""
    fast = ort.HipEngine(0, fast_math=True)
    M4, coef = cm.double_gauss_aspheric()
    ext4 = np.vstack([M4, [math.inf, 0.0, 1.0, 0.0]]); ext4[-2, 1] = 57.8
    c4 = np.vstack([coef, np.zeros((1, coef.shape[1]))])
    cases = [("cooke", Prescription.from_matrix(_ext(cm.cooke(), 77.40534796682427)), 14.7),
             ("double_gauss", Prescription.from_matrix(_ext(cm.double_gauss(), 57.8)), 29.0),
             ("double_gauss_aspheric", Prescription(ext4[:, 0], ext4[:, 1], ext4[:, 2], ext4[:, 3], c4[None]), 29.0)]
    for tag, pres, a1 in cases:
        y, x, U, V = _random_rays(20001, a1)
        worst = [0.0]
        n, nill, nfar = _fast_attribution(fast, oracle_engine, pres, y, x, np.tan(U), np.tan(V), tag, worst)
        cm.report(f"FAST policy on the {tag} fixture: rays {n}, past 1e-10 (amplified bar) {nill}, far-cap rays {nfar}, "
                  f"worst deviation {worst[0]:.2e}")
        assert nill == 0, (tag, nill)


def _cooke_relay(units, variant, pad):
    """Deep prescriptions for the parity tests: a chain of `units` Cooke triplets (test/runtests.jl:19-29 — 6 refracting
    surfaces and the flat stop plane each), alternately as they are and mirrored, a forward unit's focus being the next
    (mirrored) unit's front focus, so the beam is re-collimated after every pair and the rays stay inside the glass over
    tens of surfaces.  `pad` extra flat rows (a plane-parallel window, non-refracting planes) sit in the collimated spaces.
    variant: "sph" (spheres and planes), "even" (conic + even polynomial terms on one curved surface per unit: the
    even-asphere build), "mixed" (conic-only rows, even aspheres and one odd coefficient: the full build).
    Returns [rows][4] = [R t n K] and coef [rows][7]."""
    base = cm.cooke()[1:]                                             # 7 rows: R, t (gap behind), n (medium behind)
    bfl = 77.40534796682427
    rows = [[math.inf, 0.0, 1.0, 0.0]]
    coefs = [np.zeros(7)]
    flats = pad
    for uidx in range(units):
        fwd = uidx % 2 == 0
        last = uidx == units - 1
        if fwd:
            unit = [[r[0], r[1], r[2]] for r in base]
            unit[-1][1] = bfl if last else 2.0 * bfl                  # to the image plane | to the mirrored unit's first surface
        else:
            unit = []
            for i in range(len(base) - 1, -1, -1):                    # surfaces in reverse order, radii negated; medium and gap
                unit.append([-base[i][0], base[i - 1][1] if i > 0 else 0.0, base[i - 1][2] if i > 0 else 1.0])   # behind = those in front
            unit[-1][1] = 20.0                                        # collimated space behind a pair
        for j, r in enumerate(unit):
            K, c = 0.0, np.zeros(7)
            curved = math.isfinite(r[0])
            if variant != "sph" and curved and j == (0 if fwd else 6):
                K = -0.3; c[4] = 2e-7 * (1 if fwd else -1); c[6] = -3e-10 * (1 if fwd else -1)
            if variant == "mixed" and curved and j == 2:
                K = -0.5                                              # a conic without a polynomial
            if variant == "mixed" and uidx == 1 and j == 3:
                c[3] = 1e-6                                           # one odd coefficient: the general polynomial form
            rows.append([r[0], r[1], r[2], K]); coefs.append(c)
        if not fwd and flats > 0 and not last:                        # pad rows in the collimated space behind a pair
            take = min(flats, 3)
            flats -= take
            rows[-1][1] = 5.0
            pads = [[math.inf, 3.0, 1.5168, 0.0], [math.inf, 4.0, 1.0, 0.0], [math.inf, 2.0, 1.0, 0.0]][:take]
            if take == 1:
                pads = [[math.inf, 2.0, 1.0, 0.0]]                    # a single non-refracting plane
            for q in pads:
                rows.append(q); coefs.append(np.zeros(7))
    assert flats == 0, "not enough collimated spaces for the pad rows"
    rows.append([math.inf, 0.0, 1.0, 0.0]); coefs.append(np.zeros(7))  # image plane
    return np.array(rows), np.array(coefs)


@pytest.mark.parametrize("rows,units,pad", [(24, 3, 1), (40, 5, 3), (49, 6, 5), (50, 6, 6), (63, 8, 5), (64, 8, 6)])
def test_deep_prescriptions_both_policies(hip_engine, oracle_engine, rows, units, pad):
    """Prescriptions of 24, 40, 63 and 64 rows (ORT_MAX_ROWS: the kernel stages up to 63 records) — relay chains mixing
    spheres, planes, conics and even / odd aspheres — against the oracle, src/PupilSampling.jl:45-63 iterated up to 63
    times per ray.  Reference-sequence policy: status and NaN patterns identical, coordinates BIT-identical (polynomial
    rows: <= 1e-11, analytic against complex-step p').  FAST policy: through _fast_attribution unchanged — status
    identical on every ray, every ray within the bar; the worst deviation is reported.  The fast forms' rounding differences
    grow with depth (profiles/r04_fast_depth.log), so beyond ORT_FAST_MAX_SURFACES = 48 loop iterations (49 rows: the deepest FAST
    case here; 50 rows: the first past it) an ORT_FAST_MATH call is traced with the reference sequence: identical to the default engine's, bit for bit."""
    fast = ort.HipEngine(0, fast_math=True)
    rng = np.random.default_rng(rows)
    m = 3000
    w = np.where(np.arange(m) % 2 == 0, 1.0, 3.2)                 # every other ray from a 3.2 x wider box: a few of those miss or
    y = rng.uniform(-5, 5, m) * w; x = rng.uniform(-5, 5, m) * w  # are totally reflected tens of rows deep (the status index)
    u = np.tan(rng.uniform(-0.01, 0.01, m) * w); v = np.tan(rng.uniform(-0.01, 0.01, m) * w)
    for variant in ("sph", "even", "mixed"):
        M, coef = _cooke_relay(units, variant, pad)
        assert M.shape[0] == rows, (M.shape, rows)
        pres = Prescription(M[:, 0], M[:, 1], M[:, 2], None if variant == "sph" else M[:, 3], None if variant == "sph" else coef[None])
        ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        assert (os_ == rows).mean() > 0.3, (variant, float((os_ == rows).mean()))     # a good share reaches the last row
        gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
        assert np.array_equal(gs, os_), variant
        if variant == "sph":
            assert np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True), variant
        else:
            assert np.array_equal(np.isnan(gx), np.isnan(ox)) and np.array_equal(np.isnan(gy), np.isnan(oy)), variant
            assert max(cm.rel_err(gx, ox, 1.0).max(), cm.rel_err(gy, oy, 1.0).max()) <= 1e-11, variant
        worst = [0.0]
        n, nill, nfar = _fast_attribution(fast, oracle_engine, pres, y, x, u, v, (rows, variant), worst)
        if rows - 1 > 48:                                         # past ORT_FAST_MAX_SURFACES: the reference sequence under either flag
            fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
            assert np.array_equal(fs, gs) and np.array_equal(fx, gx, equal_nan=True) and np.array_equal(fy, gy, equal_nan=True), (rows, variant)
        cm.report(f"deep prescription, {rows} rows ({variant}): FAST rays {n}, past 1e-10 (amplified bar) {nill}, far-cap rays {nfar}, "
                  f"worst deviation {worst[0]:.2e}; reached the last row {float((os_ == rows).mean()):.2f}")


def test_skew_list_f32(hip_engine, oracle_engine):
    """`ort_trace_skew_f32` (the Float32 build over an explicit ray list, every ray its own slopes) against the same loop in
    float on the CPU (orc_trace_skew_batch_f32): reference-sequence policy bit-identical, status included — spherical, conic
    and polynomial rows; an odd ray count and a prefix of the list (tail lanes); the FAST policy has the same status on every
    ray and coordinates within Float32 rounding of it."""
    fast = ort.HipEngine(0, fast_math=True)
    M4, coef = cm.double_gauss_aspheric()
    ext4 = np.vstack([M4, [math.inf, 0.0, 1.0, 0.0]]); ext4[-2, 1] = 57.8
    c4 = np.vstack([coef, np.zeros((1, coef.shape[1]))])
    cases = [("cooke", Prescription.from_matrix(_ext(cm.cooke(), 77.40534796682427)), 14.7),
             ("double_gauss", Prescription.from_matrix(_ext(cm.double_gauss(), 57.8)), 29.0),
             ("double_gauss_aspheric", Prescription(ext4[:, 0], ext4[:, 1], ext4[:, 2], ext4[:, 3], c4[None]), 29.0)]
    nfail = 0
    for tag, pres, a1 in cases:
        y, x, U, V = _random_rays(20001, a1, seed=11)
        u, v = np.tan(U), np.tan(V)
        ox, oy, os_ = oracle_engine.skew_f32(pres, y, x, u, v)
        assert (os_ == pres.rows).mean() > 0.2, tag
        nfail += int((os_ < pres.rows).sum())
        gx, gy, gs = hip_engine.skew_f32(pres, y, x, u, v)
        assert gx.dtype == np.float32 and np.array_equal(gs, os_), tag
        if pres.coef is None:
            assert np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True), tag
        else:                                                     # analytic p' (device) against the complex step in float (oracle)
            assert np.array_equal(np.isnan(gx), np.isnan(ox)) and np.array_equal(np.isnan(gy), np.isnan(oy)), tag
            assert max(cm.rel_err(gx, ox, 1.0).max(), cm.rel_err(gy, oy, 1.0).max()) <= 2e-5, tag
        px, py, ps = hip_engine.skew_f32(pres, y, x, u, v, nrays=12345)       # a prefix: the rest of the buffers untouched
        assert np.array_equal(px[:, :12345], gx[:, :12345], equal_nan=True) and np.isnan(px[:, 12345:]).all() and not ps[12345:].any(), tag
        fx, fy, fs = fast.skew_f32(pres, y, x, u, v)
        # FAST in Float32: near-branch rays (within 1e-4, normalised) retrace with the reference sequence; the others differ
        # from it by Float32 rounding amplified by the path — compared against the FLOAT64 oracle, where both Float32
        # policies sit equally close (a few 1e-5 mm on these systems)
        dx, dy, ds = oracle_engine.skew(pres, y.astype(np.float32), x.astype(np.float32), u.astype(np.float32), v.astype(np.float32),
                                        slopes=True, want_status=True)
        same = (fs == os_)
        assert same.mean() >= 0.9995, (tag, float(same.mean()))   # a status differs only where Float32 itself decides a branch differently
        ok = same & (os_ == pres.rows)
        e_fast = max(np.abs(fx[-1, ok] - dx[-1, ok]).max(), np.abs(fy[-1, ok] - dy[-1, ok]).max())
        e_ref = max(np.abs(gx[-1, ok] - dx[-1, ok]).max(), np.abs(gy[-1, ok] - dy[-1, ok]).max())
        cm.report(f"ort_trace_skew_f32 on {tag}: status identical to the Float32 reference sequence on {float(same.mean()):.5f} of rays (FAST); "
                  f"image-plane hits vs the Float64 oracle: FAST {e_fast:.2e} mm, reference sequence {e_ref:.2e} mm")
        assert e_fast <= max(4.0 * e_ref, 2e-3), (tag, e_fast, e_ref)
    assert nfail > 0                                              # misses / NaN statuses are exercised, not only clean paths


def test_fast_policy_far_cap_and_wide_bundles(oracle_engine):
    """The FAST policy on what a lens designer never traces but the reference defines: strongly curved rows
    (|R| 6.5-30 mm) under +-14 mm, +-0.2 rad bundles — thousands of FAR-CAP hits (beyond a sphere's equator,
    where the reference refracts with the vertex-side normal, src/PupilSampling.jl:16-19), TIR and miss
    sequences, directions refracted backward.  Same checks on every ray."""
    rng = np.random.default_rng(31337)
    fast = ort.HipEngine(0, fast_math=True)
    ntot = nill = nfar = 0
    for case in range(60):
        rows = int(rng.integers(3, 15))
        aspheric = (True, "even", False)[case % 3]
        R, t, n, K, coef = _random_system(rng, rows, aspheric)
        fin = np.isfinite(R); R[fin] = np.sign(R[fin]) * rng.uniform(6.5, 30.0, int(fin.sum()))
        pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
        m = 2000
        y = rng.uniform(-14, 14, m); x = rng.uniform(-14, 14, m)
        u = np.tan(rng.uniform(-0.2, 0.2, m)); v = np.tan(rng.uniform(-0.2, 0.2, m))
        a, b, c = _fast_attribution(fast, oracle_engine, pres, y, x, u, v, case)
        ntot += a; nill += b; nfar += c
    cm.report(f"test_fast_policy_far_cap_and_wide_bundles (FAST): rays {ntot}, past 1e-10 (amplified bar) {nill}, far-cap rays {nfar}")
    assert nfar >= 1000, nfar                      # the far-cap rule is exercised, not vacuous
    assert nill <= 2e-3 * ntot, (nill, ntot)


def test_random_systems_meridional_and_paraxial(hip_engine, oracle_engine):
    """Random prescriptions through the meridional (plain-matrix and Layout{Aspheric} dispatch, Q16)
    and paraxial (+clip) kernels: NaN patterns identical, values within 1e-10 (device trig)."""
    rng = np.random.default_rng(77)
    for case in range(40):
        rows = int(rng.integers(2, 13))
        aspheric = case % 2 == 0
        R, t, n, K, coef = _random_system(rng, rows, aspheric)
        pres = Prescription(R, t, n, K if aspheric else None, coef[None] if aspheric else None)
        y = rng.uniform(-6, 6, 300); U = rng.uniform(-0.12, 0.12, 300)
        g = hip_engine.meridional(pres, y, U, layout_mode=aspheric)
        o = oracle_engine.meridional(pres, y, U, layout_mode=aspheric)
        # Base.asin's DomainError (src/RayTracing.jl:162) is reported, not thrown, by both engines; identically
        assert (hip_engine.last_domain_error is None) == (oracle_engine.last_domain_error is None), case
        for a, b in zip(g, o):
            assert np.array_equal(np.isnan(a), np.isnan(b)), case
            assert cm.rel_err(a, b, 1.0).max() <= TOL, case
        M = np.column_stack([R, t.copy(), n])
        L = ort.Lens(M)
        a_ap = rng.uniform(2.0, 9.0, L.M.shape[0])
        for clip in (False, True):
            gp = hip_engine.paraxial(L.M[:, 0], L.M[:, 1], y, U, a_ap, clip)
            op = oracle_engine.paraxial(L.M[:, 0], L.M[:, 1], y, U, a_ap, clip)
            assert np.array_equal(gp[0], op[0], equal_nan=True) and np.array_equal(gp[1], op[1], equal_nan=True), case
        assert np.array_equal(hip_engine.abcd(L.M[:, 0], L.M[:, 1]), oracle_engine.abcd(L.M[:, 0], L.M[:, 1]))


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_random_bundles_grid_and_full_trace(oracle_engine, policy):
    """Random multi-system batches, ragged grid shapes (odd nx, nx < 64, rays per bundle not a
    multiple of the tile), random stop rows and stop radii: grid summary/history and the
    full_trace pipeline (filter, ordered compaction, mirror, sigma) against the oracle.  FAST: status (stop-filter bit
    included) and survivor counts EQUAL the oracle's, coordinates <= 1e-10."""
    hip_engine = _engine(policy)
    rng = np.random.default_rng(99)
    for case in range(25):
        rows = int(rng.integers(4, 12))
        nsys = int(rng.integers(1, 5))
        Rs, ts, ns = [], [], []
        for _ in range(nsys):
            R, t, n, _, _ = _random_system(rng, rows, False)
            R[-1] = math.inf; t[-2] = rng.uniform(5.0, 30.0)          # image plane row
            Rs.append(R); ts.append(t); ns.append(n)
        pres = Prescription(np.array(Rs), np.array(ts), np.array(ns))
        ny, nx = int(rng.integers(3, 70)), int(rng.integers(3, 90))
        nb = int(rng.integers(1, 6))
        axes, bundles, off = [], [], 0
        for b in range(nb):
            a = rng.uniform(3.0, 7.0)
            axes += [ort.linrange(a, -a, ny), ort.linrange(0.0, a, nx)]
            stop = int(rng.integers(1, rows - 1))
            bundles.append(dict(system=int(rng.integers(0, nsys)), stop=stop, U=float(rng.uniform(-0.1, 0.1)), V=0.0,
                                a_stop=float(rng.uniform(2.0, 8.0)), hprime=float(rng.uniform(-3, 3)),
                                yaxis_off=off, xaxis_off=off + ny))
            off += ny + nx
        axes = np.concatenate(axes)
        g = hip_engine.grid(pres, bundles, axes, ny, nx)
        o = oracle_engine.grid(pres, bundles, axes, ny, nx)
        for key in ("xv", "yv", "xf", "yf", "xs", "ys"):
            _same(policy, g[key], o[key], (case, key))
        assert np.array_equal(g["status"], o["status"]), case
        gf = hip_engine.full_trace_grid(pres, bundles, axes, ny, nx, lookback=bool(case & 1))   # both compaction routes
        of = oracle_engine.full_trace_grid(pres, bundles, axes, ny, nx)
        for a, b in zip(gf, of):
            assert a["count"] == b["count"], case
            if b["count"]:
                _same(policy, a["ex"], b["ex"], case); _same(policy, a["ey"], b["ey"], case)
                assert cm.rel_err(a["rho"], b["rho"], 1e-3).max() <= TOL and cm.rel_err(a["theta"], b["theta"], 1e-3).max() <= TOL
                assert abs(a["rms"] - b["rms"]) <= TOL * max(b["rms"], 1e-6), case
            else:
                assert math.isnan(a["rms"])
        # the statistics-only route on the same ragged shapes (one-tile spans, a last tile of a few rays, bundles with no
        # survivor): counts equal, RMS within 1e-10 of the oracle's two-pass sigma; Float32: counts and RMS track the Float64 call
        gs = hip_engine.full_trace_grid(pres, bundles, axes, ny, nx, stats_only=True)
        g32 = hip_engine.full_trace_grid(pres, bundles, axes, ny, nx, stats_only=True, dtype=np.float32)
        for a, a32, b in zip(gs, g32, of):
            assert a["count"] == b["count"], case
            if b["count"]:
                assert abs(a["rms"] - b["rms"]) <= TOL * max(b["rms"], 1e-6), case
                assert abs(a32["count"] - b["count"]) <= max(4, 0.02 * b["count"]) and (a32["count"] == 0 or abs(a32["rms"] - b["rms"]) <= 2e-2 * max(b["rms"], 1e-3)), case
            else:
                assert math.isnan(a["rms"]) and a["count"] == 0


def test_tolerance_run_config5_pipeline(hip_engine, oracle_engine):
    """BASELINE config 5 end to end (Seidel + spot Monte-Carlo): 400 perturbed instances x 2 fields,
    three launches-worth of device work and no per-instance host code; spot-checked against the
    per-instance route (solve -> aberrations -> full_trace) through the oracle."""
    from opticalraytracing_jl_amd import analysis as an, batch, workloads
    mats = workloads.config5(None, ninst=400)
    res = batch.tolerance_run(mats, cm.DG_A, cm.DG_H, fields=(0.0, 1.0), k_rays=32, engine=hip_engine)
    assert res["rms"].shape == (400, 2) and np.all(np.isfinite(res["rms"])) and np.all(res["count"] > 0)
    for i in (0, 57, 399):
        s = ort.solve(mats[i].copy(), cm.DG_A, cm.DG_H, engine=oracle_engine)
        ab = an.aberrations(mats[i], s)
        assert abs(res["f"][i] - s.f) <= 1e-11 * abs(s.f) and res["stop"][i] == s.stop
        assert abs(res["W040"][i] - ab.W040) <= 1e-10 * max(1.0, abs(ab.W040))
        for fi, H in enumerate((0.0, 1.0)):
            e = ort.full_trace(s, H, 32, engine=oracle_engine)
            assert res["count"][i, fi] == len(e.x)
            assert abs(res["rms"][i, fi] - e.RMS) <= 1e-7
    # the perturbations move the spot size: the run resolves a distribution, not a constant
    assert res["rms"][:, 0].std() > 1e-5
    with pytest.raises(ort.DomainError):
        batch.tolerance_run(mats[:2], cm.DG_A, cm.DG_H, fields=(1.01,), engine=hip_engine)


def test_spot_batch_single_call_matches_staged_pipeline(hip_engine):
    """ort_spot_batch_f64 (solve -> aim -> axes -> trace -> statistics in one C call, every
    intermediate device-resident) against the staged host-driven route of `tolerance_run`: same
    kernels, so first-order results are bit-identical and the spot sizes differ only by the
    device-vs-host tan() of the field angle (<= 1 ulp on u)."""
    from opticalraytracing_jl_amd import batch, workloads
    mats = workloads.config5(None, ninst=300)
    fields = (0.0, 0.7, 1.0)
    a = batch.tolerance_run(mats, cm.DG_A, cm.DG_H, fields=fields, k_rays=32, engine=hip_engine)
    b = batch.spot_batch(mats, cm.DG_A, cm.DG_H, fields=fields, k_rays=32, engine=hip_engine)
    for k in ("f", "BFD", "EP_t", "W040", "W131", "W222", "y_marg", "chief_u_end"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["stop"], b["stop"])
    assert np.array_equal(a["count"], b["count"])
    assert np.all(np.abs(a["rms"] - b["rms"]) <= 1e-12 * np.maximum(a["rms"], 1e-3))
    with pytest.raises(ort.DomainError):
        batch.spot_batch(mats[:2], cm.DG_A, cm.DG_H, fields=(1.01,), engine=hip_engine)
    from opticalraytracing_jl_amd import _capi
    bad = mats[:2].copy(); bad[:, -1, 1] = 3.0
    with pytest.raises(_capi.OrtError):
        batch.spot_batch(bad, cm.DG_A, cm.DG_H, engine=hip_engine)


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_full_trace_f32_matches_its_own_summary_trace(oracle_engine, policy):
    """ort_full_trace_f32 (BASELINE config 5 names Float32): the error vectors are exactly the
    survivors of the Float32 summary trace (which test_f32_build_extension pins to the CPU float
    loop) in ray order, mirrored; RMS (accumulated in binary64) equals numpy's two-pass value on
    them; the statistics-only route agrees; and both sit within Float32 accuracy of the Float64 run."""
    import ctypes as C
    from opticalraytracing_jl_amd import _capi
    eng = ort.HipEngine(fast_math=(policy == "fast"))
    k = 96
    pres, bundles, axes = _dg_bundles(oracle_engine, k)
    bundles = bundles[:3]
    nb, rpb = len(bundles), k * k
    res = eng.full_trace_grid(pres, bundles, axes, k, k, dtype=np.float32)
    st_only = eng.full_trace_grid(pres, bundles, axes, k, k, dtype=np.float32, stats_only=True)
    ref64 = eng.full_trace_grid(pres, bundles, axes, k, k, stats_only=True)
    N = nb * rpb
    xf, yf = (np.empty(N, dtype=np.float32) for _ in range(2))
    st = np.empty(N, dtype=np.int32)
    out = _capi.ort_grid_out_f32()
    out.xf, out.yf, out.status = xf.ctypes.data, yf.ctypes.data, st.ctypes.data
    a32 = np.ascontiguousarray(axes, dtype=np.float32)
    _capi.check(eng.ctx.lib.ort_trace_grid_f32(eng.ctx.h, eng.system(pres).h, nb, _capi.make_bundles(bundles), a32.ctypes.data,
                                               a32.size, k, k, C.byref(out), eng.base_flags))
    for b, bd in enumerate(bundles):
        sl = slice(b * rpb, (b + 1) * rpb)
        keep = ((st[sl] & (1 << 16)) == 0) & ~np.isnan(xf[sl]) & ~np.isnan(yf[sl])
        ex = xf[sl][keep]; ey = yf[sl][keep] - np.float32(bd["hprime"])
        r = res[b]
        assert r["count"] == 2 * keep.sum() and r["ex"].dtype == np.float32
        assert np.array_equal(r["ex"], np.concatenate([ex, -ex])) and np.array_equal(r["ey"], np.concatenate([ey, ey]))
        e64x, e64y = r["ex"].astype(np.float64), r["ey"].astype(np.float64)
        rms = math.sqrt(np.mean((e64x - e64x.mean()) ** 2 + (e64y - e64y.mean()) ** 2))
        assert abs(r["rms"] - rms) <= 1e-12 * rms
        assert st_only[b]["count"] == r["count"] and abs(st_only[b]["rms"] - rms) <= 1e-9 * rms
        assert abs(r["count"] - ref64[b]["count"]) <= 0.002 * ref64[b]["count"]
        assert abs(r["rms"] - ref64[b]["rms"]) <= 2e-3 * ref64[b]["rms"]


def test_full_trace_batch_one_call(hip_engine, oracle_engine):
    """ort_full_trace_batch_f64 — full_trace(solve(surfaces, a, h′), H, k_rays) from the raw prescription
    in one C call — on the reference's own config-1 case (Cooke triplet, test/runtests.jl:19-35) and
    on perturbed Double-Gauss instances: counts identical to and RMS within 1e-10 of the statistics-only
    pipeline (same aiming kernels), vectors consistent with their own statistics, and the whole thing
    within the aiming tolerance (sqrt(eps), RayTracing.jl:1) of the host-driven oracle route."""
    from opticalraytracing_jl_amd import batch, workloads
    cases = [(cm.cooke()[None], cm.COOKE_A, cm.COOKE_H, (0.0, 1.0), 64),
             (workloads.config5(None, ninst=6), cm.DG_A, cm.DG_H, (0.0, 0.7, 1.0), 48)]
    for mats, a, hp, fields, k in cases:
        fo, res = batch.full_trace_systems(mats, a, hp, fields=fields, k_rays=k, engine=hip_engine)
        sb = batch.spot_batch(mats, a, hp, fields=fields, k_rays=k, engine=hip_engine)
        nf = len(fields)
        for b, r in enumerate(res):
            i, fi = divmod(b, nf)
            assert r["count"] == sb["count"][i, fi] and r["count"] > 0
            assert abs(r["rms"] - sb["rms"][i, fi]) <= 1e-10 * r["rms"]
            half = r["count"] // 2
            assert np.array_equal(r["ex"][half:], -r["ex"][:half]) and np.array_equal(r["ey"][half:], r["ey"][:half])
            rms = math.sqrt(np.mean((r["ex"] - r["ex"].mean()) ** 2 + (r["ey"] - r["ey"].mean()) ** 2))
            assert abs(r["rms"] - rms) <= 1e-12 * rms
            assert abs(r["rho"].max() - 1.0) <= 1e-15                 # r ./= maximum(r)  (PupilSampling.jl:142)
        for i in (0, mats.shape[0] - 1):
            s = ort.solve(mats[i].copy(), a, hp, engine=oracle_engine)
            assert fo["stop"][i] == s.stop and abs(fo["f"][i] - s.f) <= 1e-11 * abs(s.f)
            for fi, H in enumerate(fields):
                e = ort.full_trace(s, H, k, engine=oracle_engine)
                r = res[i * nf + fi]
                # the edge rays are aimed AT the stop rim to within sqrt(eps): the outermost grid rows sit on
                # it and may fall either side (reference behaviour too; edge-ray heights are parity-unpinned)
                assert abs(r["count"] - len(e.x)) <= 8
                tol = 1e-6 if r["count"] == len(e.x) else 5e-3
                assert abs(r["rms"] - e.RMS) <= tol * max(e.RMS, 1e-3)


def test_full_trace_layout_batch_aspheric(hip_engine, oracle_engine):
    """ort_full_trace_layout_batch_f64: the one-call pipeline on aspheric Layouts (conic constants +
    polynomial rows; the reversed system with K, p plainly reversed, RayTracing.jl:272-274) against the
    host-driven route `full_trace(solve(Layout), H, k)` through the oracle, and against the staged device
    route through the same library."""
    from opticalraytracing_jl_amd import batch, workloads
    fields, k = (0.0, 0.7, 1.0), 48
    M4s, coefs = zip(*(workloads.double_gauss_aspheric(line) for line in (0, 1, 2)))
    mats, coef = np.array(M4s), np.array(coefs)
    fo, res = batch.full_trace_systems(mats, cm.DG_A, cm.DG_H, fields=fields, k_rays=k, engine=hip_engine, coef=coef)
    for i in range(3):
        lay = ort.Layout(mats[i, :, 0], mats[i, :, 1], mats[i, :, 2], mats[i, :, 3], [c for c in coef[i]])
        so = ort.solve(lay, cm.DG_A, cm.DG_H, engine=oracle_engine)
        sg = ort.solve(lay, cm.DG_A, cm.DG_H, engine=hip_engine)
        assert fo["stop"][i] == so.stop and abs(fo["f"][i] - so.f) <= 1e-11 * abs(so.f)
        for fi, H in enumerate(fields):
            r = res[i * len(fields) + fi]
            eo = ort.full_trace(so, H, k, engine=oracle_engine)
            eg = ort.full_trace(sg, H, k, engine=hip_engine)
            # the outermost grid rows sit ON the stop rim (the edge rays are aimed at it to sqrt(eps)) and fall either
            # side with the last digits of the aiming; on this strongly aberrated system each such ray moves the RMS by
            # ~0.1 %.  So: total counts within a few rays, RMS within 1 %, and — the sharp check — the two sets of
            # transverse errors coincide (to 1e-5 mm) except for those few rim rays.
            assert r["count"] > 0
            key = lambda x, y: set(zip(np.round(np.asarray(x) / 1e-5).astype(np.int64), np.round(np.asarray(y) / 1e-5).astype(np.int64)))
            mine = key(r["ex"], r["ey"])
            for e in (eo, eg):
                assert abs(r["count"] - len(e.x)) <= 8 and abs(r["rms"] - e.RMS) <= 1e-2 * e.RMS, (i, H)
                theirs = key(e.x, e.y)
                assert len(mine ^ theirs) <= 24 and len(mine & theirs) >= 0.98 * min(len(mine), len(theirs)), (i, H)
            half = r["count"] // 2
            assert np.array_equal(r["ex"][half:], -r["ex"][:half])
    # the same call under the FAST policy (these rows take the even-asphere kernel build, one ray per lane on this size):
    # survivor counts identical, error vectors and RMS within 1e-10 of the reference-sequence policy's
    fast = ort.HipEngine(0, fast_math=True)
    fo_f, res_f = batch.full_trace_systems(mats, cm.DG_A, cm.DG_H, fields=fields, k_rays=k, engine=fast, coef=coef)
    assert np.array_equal(fo_f["stop"], fo["stop"]) and np.array_equal(fo_f["f"], fo["f"])
    for r, q in zip(res, res_f):
        assert q["count"] == r["count"]
        for key_ in ("ex", "ey", "rho", "theta"):
            assert cm.rel_err(q[key_], r[key_], 1.0).max() <= 1e-10, key_
        assert abs(q["rms"] - r["rms"]) <= 1e-10 * r["rms"]
    # the aspheric terms matter: the same prescriptions without them give different spots
    _, plain = batch.full_trace_systems(mats[:, :, :3], cm.DG_A, cm.DG_H, fields=fields, k_rays=k, engine=hip_engine)
    assert max(abs(p["rms"] - r["rms"]) / r["rms"] for p, r in zip(plain, res)) > 1e-2


def test_plain_c_caller_matches_host_mirror(hip_engine):
    """The drop-in boundary used from plain C (examples/cooke_full_trace.c, no Python in the process): same
    first-order numbers and spot sizes as the Python host mirror over the same library."""
    import re
    import subprocess
    from opticalraytracing_jl_amd import batch
    exe = cm.build_c_example()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    f = float(re.search(r"f = ([0-9.]+)", r.stdout).group(1))
    rows = re.findall(r"H = ([0-9.]+)  rays = (\d+)  RMS = ([0-9.]+)  \(recomputed from the vectors: ([0-9.]+)\)", r.stdout)
    assert len(rows) == 2
    fo, res = batch.full_trace_systems(cm.cooke()[None], cm.COOKE_A, cm.COOKE_H, (0.0, 1.0), 64,
                                       engine=ort.HipEngine())           # IEEE policy, as the C program's flags = 0
    assert abs(f - fo["f"][0]) < 1e-6
    for (H, cnt, rms, rms2), ref in zip(rows, res):
        assert int(cnt) == ref["count"] and abs(float(rms) - ref["rms"]) < 1e-9 and abs(float(rms) - float(rms2)) < 1e-9


def test_spot_batch_f32_tracks_f64(hip_engine):
    """ort_spot_batch_f32: solve + aiming in binary64, pupil trace in binary32 — counts and RMS within
    Float32 accuracy of the Float64 pipeline, first-order structs identical."""
    from opticalraytracing_jl_amd import batch, workloads
    mats = workloads.config5(None, ninst=200)
    a = batch.spot_batch(mats, cm.DG_A, cm.DG_H, fields=(0.0, 1.0), k_rays=64, engine=hip_engine)
    b = batch.spot_batch(mats, cm.DG_A, cm.DG_H, fields=(0.0, 1.0), k_rays=64, engine=hip_engine, dtype=np.float32)
    assert np.array_equal(a["f"], b["f"]) and np.array_equal(a["W040"], b["W040"])
    assert np.all(np.abs(a["count"] - b["count"]) <= 0.01 * a["count"])
    assert np.all(np.abs(a["rms"] - b["rms"]) <= 5e-3 * a["rms"])


def test_clear_aperture_extension(hip_engine, oracle_engine):
    """ort_system_set_apertures (extension, off by default): a ray leaving a surface's clear
    semi-diameter is flagged (bit 17 + index of the first such surface) and compacted away by the
    full_trace filter.  Checked against the oracle's per-surface history with the same predicate
    x*x + y*y > a*a; coordinates and the NaN status index must not change."""
    import dataclasses
    from opticalraytracing_jl_amd import _capi
    k = 64
    pres, bundles, axes = _dg_bundles(oracle_engine, k)
    S = pres.rows - 1
    ap = np.concatenate([0.93 * cm.DG_A, [math.inf]])           # rows-1 = S entries, image plane open
    assert ap.size == S
    pres_ap = dataclasses.replace(pres, apertures=ap)
    base = hip_engine.grid(pres, bundles, axes, k, k)
    g = hip_engine.grid(pres_ap, bundles, axes, k, k)
    o = oracle_engine.grid(pres, bundles, axes, k, k)
    for key in ("xv", "yv", "xf", "yf", "xs", "ys"):
        assert np.array_equal(g[key], base[key], equal_nan=True), key
    assert np.array_equal(g["status"] & 0x1ffff, base["status"])
    out = (o["xv"] * o["xv"] + o["yv"] * o["yv"]) > (ap * ap)[:, None]          # [S][N]
    vig = out.any(axis=0)
    first = np.where(vig, out.argmax(axis=0) + 1, 0)
    assert vig.any() and not vig.all()
    assert np.array_equal((g["status"] & _capi.ORT_STATUS_VIGNETTED) != 0, vig)
    assert np.array_equal(np.where(vig, ((g["status"] >> 20) & 0xff) + 1, 0), first)
    # full_trace drops them, in order, and the statistics-only route agrees
    ft = hip_engine.full_trace_grid(pres_ap, bundles, axes, k, k)
    so = hip_engine.full_trace_grid(pres_ap, bundles, axes, k, k, stats_only=True)
    rpb = k * k
    for b, bd in enumerate(bundles):
        sl = slice(b * rpb, (b + 1) * rpb)
        keep = ((g["status"][sl] & ((1 << 16) | (1 << 17))) == 0) & ~np.isnan(g["xf"][sl]) & ~np.isnan(g["yf"][sl])
        ex = g["xf"][sl][keep]; ey = g["yf"][sl][keep] - bd["hprime"]
        assert ft[b]["count"] == 2 * keep.sum() == so[b]["count"]
        assert np.array_equal(ft[b]["ex"], np.concatenate([ex, -ex])) and np.array_equal(ft[b]["ey"], np.concatenate([ey, ey]))
        assert abs(so[b]["rms"] - ft[b]["rms"]) <= 1e-10 * ft[b]["rms"]
    # clearing restores the reference behaviour
    sysd = hip_engine.system(pres_ap)
    sysd.set_apertures(None)
    g2 = hip_engine.grid(pres_ap, bundles, axes, k, k)
    assert np.array_equal(g2["status"], base["status"])
    sysd.set_apertures(ap)


def test_device_axes_match_host_range(hip_engine):
    """ort_make_axes_f64 == api.linrange_batch bit for bit (same double-double algorithm), incl. the
    dyadic tie (81/108) that separates an approximate lerp from the exactly rounded one."""
    import ctypes as C
    from opticalraytracing_jl_amd import _capi, api
    rng = np.random.default_rng(5)
    nb, ny, nx = 200, 109, 64
    ends = np.column_stack([rng.uniform(-50, 50, nb), rng.uniform(-50, 50, nb), np.zeros(nb), rng.uniform(1, 30, nb)])
    ends[0, :2] = (20.595904298688694, -44.811747227337484)
    axes = np.empty(nb * (ny + nx))
    _capi.check(hip_engine.ctx.lib.ort_make_axes_f64(hip_engine.ctx.h, nb, ny, nx, ends.ctypes.data, axes.ctypes.data, 0))
    axes = axes.reshape(nb, ny + nx)
    assert np.array_equal(axes[:, :ny], api.linrange_batch(ends[:, 0], ends[:, 1], ny))
    assert np.array_equal(axes[:, ny:], api.linrange_batch(ends[:, 2], ends[:, 3], nx))


def test_full_trace_stats_only_single_pass(hip_engine, oracle_engine):
    """Statistics-only full_trace (one pass: per-tile (n, mean, M2) merged with Chan's update, no ray-sized
    buffer) == the compacting pipeline and the oracle's two-pass sigma, incl. ragged tiles, several
    tiles per bundle, an offset spot (|mean| >> sigma) and an empty bundle."""
    pres, bundles, axes = _dg_bundles(oracle_engine, 130)                    # 33 tiles per bundle, last one ragged
    full = hip_engine.full_trace_grid(pres, bundles, axes, 130, 130)
    stat = hip_engine.full_trace_grid(pres, bundles, axes, 130, 130, stats_only=True)
    orc = oracle_engine.full_trace_grid(pres, bundles, axes, 130, 130)
    for f, s, o in zip(full, stat, orc):
        assert s["count"] == f["count"] == o["count"]
        assert abs(s["rms"] - o["rms"]) <= 1e-12 * o["rms"] and abs(f["rms"] - o["rms"]) <= 1e-12 * o["rms"]
    # offset spot: h' far from the actual image height -> mean ey ~ 5 mm with sigma ~ 0.02 mm
    off = [dict(b, hprime=b["hprime"] - 5.0) for b in bundles[:3]]
    s2 = hip_engine.full_trace_grid(pres, off, axes, 130, 130, stats_only=True)
    o2 = oracle_engine.full_trace_grid(pres, off, axes, 130, 130)
    for s, o in zip(s2, o2):
        assert abs(s["rms"] - o["rms"]) <= 1e-10 * o["rms"]
    # a stop radius nothing passes: count 0, RMS NaN (maximum(r) of an empty set throws in the reference)
    none = [dict(bundles[0], a_stop=-1.0)]
    s3 = hip_engine.full_trace_grid(pres, none, axes, 130, 130, stats_only=True)[0]
    assert s3["count"] == 0 and math.isnan(s3["rms"])


def test_image_hits_config4_sharded(hip_engine, oracle_engine):
    """BASELINE config 4 driver (batch.image_hits): every rank traces its contiguous slab of bundles;
    the slabs concatenated in rank order equal the single launch (what the all-gather reassembles),
    and a bundle equals the per-call route through the oracle."""
    import torch
    from opticalraytracing_jl_amd import batch, workloads
    mats = np.array([workloads.double_gauss(line, g) for g in (-1.0, 0.0, 1.0) for line in (0, 1)])
    fields = (0.0, 0.7, 1.0)
    whole = batch.image_hits(mats, cm.DG_A, cm.DG_H, fields, 24, engine=hip_engine)
    parts = [batch.image_hits(mats, cm.DG_A, cm.DG_H, fields, 24, engine=hip_engine, shard=(r, 4)) for r in range(4)]
    for j in range(3):
        cat = torch.cat([p[j] for p in parts])
        assert torch.equal(torch.nan_to_num(cat.double()), torch.nan_to_num(whole[j].double()))
    # bundle 4 = (system 1, field 0.7) against solve -> aim -> explicit trace through the oracle
    s = ort.solve(mats[1].copy(), cm.DG_A, cm.DG_H, engine=oracle_engine)
    aim = ort.full_trace_aim(s.layout, s, 0.7, engine=oracle_engine)
    pres = ort.extended_prescription(s.layout, aim.focus)
    yy = np.repeat(ort.linrange(aim.y1, aim.y2, 24), 24); xx = np.tile(ort.linrange(-aim.y_EP, aim.y_EP, 24), 24)
    ox, oy = oracle_engine.skew(pres, yy, xx, np.full(576, math.tan(aim.U)), np.zeros(576), slopes=True)
    assert np.abs(whole[0][4].cpu().numpy().ravel() - ox[-1]).max() <= 1e-6      # aiming tolerance sqrt(eps)
    assert np.abs(whole[1][4].cpu().numpy().ravel() - oy[-1]).max() <= 1e-6
    # Float32 hits (config 5's payload type): same aiming, Float32 trace — within Float32 accuracy of the above
    w32 = batch.image_hits(mats, cm.DG_A, cm.DG_H, fields, 24, engine=hip_engine, dtype=np.float32)
    assert w32[0].dtype == torch.float32 and w32[0].shape == whole[0].shape
    ok = ~torch.isnan(whole[0]) & ~torch.isnan(w32[0])
    assert float(ok.float().mean()) > 0.5
    assert float((w32[0][ok].double() - whole[0][ok]).abs().max()) <= 2e-4
    assert float((w32[1][ok].double() - whole[1][ok]).abs().max()) <= 2e-4


def test_image_hits_plan_row_shards_and_packed_slab(hip_engine):
    """ImageHitsPlan (what bench.py's multi-GPU step launches): pupil-ROW sharding for world sizes that do not
    divide the bundle count — slabs that start and end inside a bundle — concatenated in rank order equal the
    single launch bit for bit; the packed [2][n] slab is what ONE all-gather sends."""
    import torch
    from opticalraytracing_jl_amd import batch, dist as odist, workloads
    mats = np.array([workloads.double_gauss(line, g) for g in (-1.0, 1.0) for line in (0, 1, 2)])
    fields = (0.0, 0.7, 1.0)                                    # 18 bundles
    k = 20
    whole = batch.ImageHitsPlan(mats, cm.DG_A, cm.DG_H, fields, k, engine=hip_engine)
    wh = whole.new_hits(); whole.trace(wh); hip_engine.ctx.synchronize()
    assert wh.shape == (2, 18 * k * k)
    for world in (4, 7):                                        # 360 rows: 90 each; 51/52 rows -> uneven
        parts = []
        for r in range(world):
            plan = batch.ImageHitsPlan(mats, cm.DG_A, cm.DG_H, fields, k, engine=hip_engine, shard=(r, world), unit="row")
            lo, hi = odist.shard_bounds(18 * k, world)[r]
            assert plan.n_rays == (hi - lo) * k and len(plan.segs) <= 3
            h = plan.new_hits(); plan.trace(h); hip_engine.ctx.synchronize()
            parts.append(h)
        cat = torch.cat(parts, dim=1)
        assert torch.equal(torch.nan_to_num(cat), torch.nan_to_num(wh)), world


def test_native_rccl_packed_ragged_and_overlap_single_rank():
    """The native reassembly entry points on the one GPU of the test box (1-rank communicator): the packed
    single-collective all-gather on the communicator's own stream, ordered after the engine's stream
    (ort_comm_wait / _wait_lag / _synchronize), and the ragged gather (counts, then slabs).  Multi-rank order is
    covered by the gloo tests and is RCCL's contract."""
    import torch
    from opticalraytracing_jl_amd import dist as odist
    eng = ort.default_engine()
    comm = odist.RcclComm(eng, 1, 0, odist.RcclComm.unique_id())
    assert comm.nranks_seen == 1
    hits = torch.randn((2, 200003), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    g = comm.allgather_hits_packed(hits)
    assert g.shape == (1, 2, 200003) and torch.equal(g[0], hits)
    for i in range(5):                                          # ring of completion events
        comm.allgather_hits_packed(hits, g, wait=False)
        comm.wait_lag(1)
    comm.wait(); comm.synchronize()
    assert torch.equal(g[0], hits)
    vals = torch.randn(12345, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    out, counts = comm.allgather_ragged(vals, 20000)
    assert counts.tolist() == [12345] and torch.equal(out, vals)
    comm.close()


def test_device_buffers_through_the_c_abi(hip_engine, oracle_engine):
    """What julia/OpticalRayTracingHIP.jl's device-resident path does (`DeviceArray`, `trace_grid_device`,
    `full_trace_rms`), call for call through ctypes with no torch tensor involved: ort_device_malloc / _upload,
    a grid trace with ORT_DEVICE_PTRS that leaves the history on the GPU, ort_device_download of ONE row of it,
    and the statistics-only full_trace (16 bytes out)."""
    import ctypes as C
    from opticalraytracing_jl_amd import _capi
    lib, h = hip_engine.ctx.lib, hip_engine.ctx.h
    system = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=oracle_engine)
    aim = ort.full_trace_aim(system.layout, system, 0.7, engine=oracle_engine)
    pres = ort.extended_prescription(system.layout, aim.focus)
    k = 64
    axes = np.concatenate([ort.linrange(aim.y1, aim.y2, k), ort.linrange(0.0, aim.y_EP, k // 2)])
    b = dict(system=0, stop=aim.stop, U=aim.U, V=0.0, a_stop=aim.a_stop, hprime=aim.hprime, yaxis_off=0, xaxis_off=k)
    N, S = k * (k // 2), pres.rows - 1
    ptrs = []

    def dmalloc(nbytes):
        p = C.c_void_p()
        _capi.check(lib.ort_device_malloc(h, nbytes, C.byref(p)))
        ptrs.append(p)
        return p

    d_axes = dmalloc(axes.nbytes); _capi.check(lib.ort_device_upload(h, d_axes, axes.ctypes.data, axes.nbytes))
    d_xv, d_yv = dmalloc(8 * S * N), dmalloc(8 * S * N)
    out = _capi.ort_grid_out_f64(); out.xv, out.yv, out.ld = d_xv.value, d_yv.value, N
    sysd = hip_engine.system(pres)
    _capi.check(lib.ort_trace_grid_f64(h, sysd.h, 1, _capi.make_bundles([b]), d_axes, axes.size, k, k // 2, C.byref(out),
                                       hip_engine.base_flags | _capi.ORT_DEVICE_PTRS))
    last = np.empty(N)                                             # the image-plane row only: 16 KB, not the history
    _capi.check(lib.ort_device_download(h, last.ctypes.data, d_yv.value + 8 * (S - 1) * N, last.nbytes))
    o = oracle_engine.grid(pres, [b], axes, k, k // 2)
    assert np.array_equal(last, o["yv"][-1], equal_nan=True)
    cnt = np.zeros(1, dtype=np.int64); rms = np.zeros(1)
    _capi.check(lib.ort_full_trace_f64(h, sysd.h, 1, _capi.make_bundles([b]), axes.ctypes.data, axes.size, k, k // 2,
                                       None, None, None, None, cnt.ctypes.data, rms.ctypes.data, hip_engine.base_flags))
    of = oracle_engine.full_trace_grid(pres, [b], axes, k, k // 2)[0]
    assert int(cnt[0]) == of["count"] and abs(rms[0] - of["rms"]) <= TOL * of["rms"]
    for p in ptrs:
        _capi.check(lib.ort_device_free(h, p))


def test_config5_full_size_properties(hip_engine, oracle_engine):
    """BASELINE config 5 at FULL size (10^4 perturbed Double-Gauss instances x 2 fields x 256 x 128 half pupil =
    6.6e8 rays, Float32 trace, ONE C call `ort_spot_batch_f32`): size-independent properties —
    (1) a permutation of the instances permutes every output with it (no cross-instance state, a checksum of
    checksums); (2) a 200-instance slice of the full call equals the same 200 instances run alone, bit for bit;
    (3) the Float32 spot statistics track the Float64 call on that slice (counts within 2e-4, RMS within 1e-3
    relative); (4) the first-order structs equal ort_first_order_f64's; and one instance against the per-call route
    through the oracle (solve -> aim -> full_trace)."""
    from opticalraytracing_jl_amd import batch, workloads
    ninst = 10 ** 4
    mats = workloads.config5(None, ninst=ninst)
    fields = (0.0, 1.0)
    full = batch.spot_batch(mats, cm.DG_A, cm.DG_H, fields, 256, engine=hip_engine, dtype=np.float32)
    assert full["rms"].shape == (ninst, 2) and np.isfinite(full["rms"]).all() and (full["count"] > 0).all()
    nominal = ort.full_trace(ort.solve(cm.double_gauss(0), cm.DG_A, cm.DG_H, engine=oracle_engine), 0.0, 256, engine=oracle_engine)
    assert (np.abs(full["rms"][:, 0] / nominal.RMS - 1.0) < 0.5).all()          # perturbations of 1e-3: same spot to tens of percent
    perm = np.random.default_rng(5).permutation(ninst)
    shuf = batch.spot_batch(mats[perm], cm.DG_A, cm.DG_H, fields, 256, engine=hip_engine, dtype=np.float32)
    for key in ("rms", "count", "f", "W040", "stop"):
        assert np.array_equal(shuf[key], full[key][perm]), key
    sl = slice(4321, 4521)
    part = batch.spot_batch(mats[sl], cm.DG_A, cm.DG_H, fields, 256, engine=hip_engine, dtype=np.float32)
    assert np.array_equal(part["rms"], full["rms"][sl]) and np.array_equal(part["count"], full["count"][sl])
    p64 = batch.spot_batch(mats[sl], cm.DG_A, cm.DG_H, fields, 256, engine=hip_engine)
    # Float32 hits are good to ~3.5e-5 mm; these spots are ~0.02 mm RMS, so the statistics agree to a few 1e-4
    assert np.abs(part["count"] / p64["count"] - 1.0).max() <= 2e-4 and np.abs(part["rms"] / p64["rms"] - 1.0).max() <= 1e-3
    fo = batch.first_order_arrays(hip_engine, mats[sl], cm.DG_A, cm.DG_H)
    for key in ("f", "EBFD", "W040", "W131", "stop"):
        assert np.array_equal(fo[key], p64[key]), key
    i = 4400
    s = ort.solve(mats[i].copy(), cm.DG_A, cm.DG_H, engine=oracle_engine)
    e = ort.full_trace(s, 1.0, 256, engine=oracle_engine)
    assert len(e.x) == p64["count"][i - 4321, 1] and abs(e.RMS - p64["rms"][i - 4321, 1]) <= 1e-6 * e.RMS   # aiming atol sqrt(eps)


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_config4_full_size_properties(oracle_engine, policy):
    """BASELINE config 4 at FULL size (32 zoom positions x 5 index columns x 5 fields x 512^2 full pupil =
    2.1e8 rays, summary trace into the packed [2][n] hit slab that ONE all-gather sends): size-independent
    properties — (1) eight row-level shards concatenated in rank order equal the single launch bit for bit (what
    the reassembly relies on, src/PupilSampling.jl:123,134-137); (2) the pupil's x axis is symmetric about the
    meridional plane, so hits mirror: x_f(-x) = -x_f(x), y_f(-x) = y_f(x) to 1e-11, NaN pattern exactly; (3) a strided sample of 4000
    rays against the per-call route through the oracle (explicit pupil coordinates from the plan's own aimed
    axes): bit-identical in the reference-sequence policy, <= 1e-12 in the fast one."""
    import torch
    from opticalraytracing_jl_amd import batch, dist as odist, workloads
    eng = ort.HipEngine(0, fast_math=(policy == "fast"))
    mats = np.array([workloads.double_gauss(line, -1.5 + 3.0 * z / 31) for z in range(32) for line in (0, 1, 2, 1, 2)])
    fields = (0.0, 0.5, 0.7, 0.85, 1.0)
    k, nb = 512, 160 * 5
    whole = batch.ImageHitsPlan(mats, workloads.DG_A, workloads.DG_H, fields, k, engine=eng)
    wh = whole.new_hits(); st = torch.empty(whole.n_rays, dtype=torch.int32, device=whole.dev)
    whole.trace(wh, st); eng.ctx.synchronize()
    assert wh.shape == (2, nb * k * k)
    # (1) 8 ranks, pupil-row shards (800 * 512 rows / 8): slabs start and end on bundle borders here; 7 does not divide
    for world in (8, 7):
        off = 0
        for r in range(world):
            plan = batch.ImageHitsPlan(mats, workloads.DG_A, workloads.DG_H, fields, k, engine=eng, shard=(r, world), unit="row")
            h = plan.new_hits(); plan.trace(h); eng.ctx.synchronize()
            assert torch.equal(torch.nan_to_num(h), torch.nan_to_num(wh[:, off:off + plan.n_rays])), (world, r)
            off += plan.n_rays
            del h, plan
        assert off == whole.n_rays
    # (2) mirror symmetry in x, every bundle
    g = wh.view(2, nb, k, k)
    nanx = torch.isnan(g[0])
    assert torch.equal(nanx, nanx.flip(-1)) and float(nanx.float().mean()) < 0.5
    for c, sign in ((0, 1.0), (1, -1.0)):          # x_f antisymmetric, y_f symmetric (the axis values mirror to the last bit or two)
        worst_sym = 0.0
        for b0 in range(0, nb, 100):
            blk = torch.nan_to_num(g[c, b0:b0 + 100])
            worst_sym = max(worst_sym, float((blk + sign * blk.flip(-1)).abs().max()))
        assert worst_sym <= 1e-11, (c, worst_sym)
    # (3) strided sample through the oracle, bundle by bundle (explicit rays: the plan's axes, its aimed field angle)
    axes = whole.d_axes.cpu().numpy().reshape(nb, 2, k)
    rng = np.random.default_rng(44)
    worst, nsample = 0.0, 0
    for b in rng.choice(nb, 40, replace=False):
        iy = rng.integers(0, k, 100); ix = rng.integers(0, k, 100)
        yy, xx = axes[b, 0, iy], axes[b, 1, ix]
        ox, oy = oracle_engine.skew(whole.ext, yy, xx, np.full(100, math.tan(whole.aim_U[b])), np.zeros(100),
                                    isys=int(whole.inst[b]), slopes=True)
        hx = g[0, b][iy, ix].cpu().numpy(); hy = g[1, b][iy, ix].cpu().numpy()
        if policy == "ieee":
            assert np.array_equal(hx, ox[-1], equal_nan=True) and np.array_equal(hy, oy[-1], equal_nan=True), b
        else:
            assert np.array_equal(np.isnan(hx), np.isnan(ox[-1])), b
            ok = ~np.isnan(hx)
            worst = max(worst, float(np.abs(hx[ok] - ox[-1][ok]).max()), float(np.abs(hy[ok] - oy[-1][ok]).max()))
        nsample += 100
    assert nsample == 4000 and worst <= 1e-12


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_tessar_spot_figure_of_the_reference_docs(oracle_engine, policy):
    """The one reference-held number that pins the skew loop OFF the meridional plane to its printed digits: the title
    of docs/src/assets/images/real_spot_diagram.png, "RMS Spot Size = 0.11975" for `full_trace(system, 0.0)` on the
    Tessar of docs/setup.jl (tests/test_oracle_reference_vectors.py::test_tessar_real_spot_diagram_figure explains the
    two edge rays).  Here the whole chain runs on the device, twice: (1) the reference's call sequence through the HIP
    engine (solve, aiming driven from the host, then grid, trace, stop filter, compaction, mirror and sigma in
    `ort_full_trace_f64`); (2) ONE C call, `ort_spot_batch_f64`: first-order solve, aiming (`k_aim`), trace and
    statistics all on the device."""
    from opticalraytracing_jl_amd import batch
    eng = ort.HipEngine(0, fast_math=(policy == "fast"))
    system = ort.solve(cm.tessar(), cm.TESSAR_A, cm.TESSAR_H, engine=eng)
    e = ort.full_trace(system, 0.0, engine=eng)
    assert f"{e.RMS:.5f}" == "0.11975" and len(e.x) == 2 * 1560
    assert abs(np.abs(e.x).max() - 0.37) < 0.01
    sb = batch.spot_batch(cm.tessar()[None], cm.TESSAR_A, cm.TESSAR_H, (0.0,), 64, engine=eng)
    assert f"{sb['rms'][0, 0]:.5f}" == "0.11975" and sb["count"][0, 0] == 2 * 1560


def test_empty_single_and_all_dropped_inputs(hip_engine, oracle_engine):
    """The edges of the input space: an empty ray list (nothing to do, ORT_OK), a single ray, a 1 x 1 pupil grid, and a
    bundle of which NO ray passes the stop filter — the reference then fails in `maximum(r)` over an empty collection
    (src/PupilSampling.jl:142); the C ABI returns count 0 (and the host mirror raises, as the reference does)."""
    pres = Prescription.from_matrix(_ext(cm.cooke(), 77.40534796682427))
    S = pres.rows - 1
    e = np.zeros(0)
    gx, gy, gs = hip_engine.skew(pres, e, e, e, e, slopes=True, want_status=True)
    assert gx.shape == (S, 0) and gy.shape == (S, 0) and gs.shape == (0,)
    one = [np.array([3.0]), np.array([-2.0]), np.array([0.05]), np.array([-0.02])]
    gx, gy, gs = hip_engine.skew(pres, *one, slopes=True, want_status=True)
    ox, oy, os_ = oracle_engine.skew(pres, *one, slopes=True, want_status=True)
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gs, os_) and gx.shape == (S, 1)
    # 1 x 1 pupil grid
    b = dict(system=0, stop=5, U=0.1, V=0.0, a_stop=10.3, hprime=0.0, yaxis_off=0, xaxis_off=1)
    axes = np.array([2.0, 1.0])
    g = hip_engine.grid(pres, [b], axes, 1, 1)
    o = oracle_engine.grid(pres, [b], axes, 1, 1)
    assert np.array_equal(g["xv"], o["xv"]) and np.array_equal(g["yv"], o["yv"]) and np.array_equal(g["status"], o["status"])
    # every ray outside the stop: a 4 x 4 grid far off axis
    axes = np.concatenate([np.linspace(13.0, 14.0, 4), np.linspace(13.0, 14.0, 4)])
    b = dict(system=0, stop=5, U=0.0, V=0.0, a_stop=10.3, hprime=0.0, yaxis_off=0, xaxis_off=4)
    for route in (False, True):
        r = hip_engine.full_trace_grid(pres, [b], axes, 4, 4, lookback=route)[0]
        assert r["count"] == 0 and len(r["ex"]) == 0
    ro = oracle_engine.full_trace_grid(pres, [b], axes, 4, 4)[0]
    assert ro["count"] == 0
    sysm = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=hip_engine)
    aim = ort.full_trace_aim(sysm.layout, sysm, 0.0, engine=hip_engine)
    aim.y1, aim.y2, aim.y_EP = 14.0, 13.0, 14.0                 # a pupil box that misses the stop altogether
    aim.a_stop = 0.5
    with pytest.raises(ValueError):
        ort.full_trace_grid(sysm.layout, aim, 8, engine=hip_engine)


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_history_beyond_2_31_elements(oracle_engine, policy):
    """Maximum sizes: a history of 12 x 1.9e8 = 2.27e9 elements per array (36 GB for x and y together — the 288 GB of
    the card hold eight of these), so that row offsets i * ld + ray pass 2^31 and 2^32 bytes many times over: 20 bundles
    (the 9 of config 2 and 11 of them again) x 3072^2 rays, S = 12.  Checked: the last history row equals the summary
    output; the second copy of a bundle equals the first bit for bit (same descriptors, 1.7e9 elements apart); x mirror
    symmetry of the LAST bundle's last rows; a strided sample incl. the very last ray through the oracle."""
    import ctypes as C
    import torch
    from opticalraytracing_jl_amd import _capi
    eng = ort.default_engine()
    torch.cuda.empty_cache()                                    # 37 GB below: hand back what earlier tests' tensors cached
    k = 3072
    pres, bundles, axes = _dg_bundles(oracle_engine, k)
    bundles = (bundles + bundles + bundles)[:20]
    nb, rpb = len(bundles), k * k
    N, S = nb * rpb, pres.rows - 1
    assert S * N > 2 ** 31
    dev = torch.device("cuda:0")
    d_axes = torch.from_numpy(axes).to(dev)
    xv = torch.empty((S, N), dtype=torch.float64, device=dev)
    yv = torch.empty((S, N), dtype=torch.float64, device=dev)
    xf = torch.empty(N, dtype=torch.float64, device=dev); yf = torch.empty_like(xf)
    st = torch.empty(N, dtype=torch.int32, device=dev)
    out = _capi.ort_grid_out_f64()
    out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
    out.xf, out.yf, out.status = xf.data_ptr(), yf.data_ptr(), st.data_ptr()
    sysd = eng.system(pres)
    barr = _capi.make_bundles(bundles)
    torch.cuda.synchronize()
    fast = policy == "fast"
    _capi.check(eng.ctx.lib.ort_trace_grid_f64(eng.ctx.h, sysd.h, nb, barr, d_axes.data_ptr(), axes.size, k, k,
                                               C.byref(out), _capi.ORT_DEVICE_PTRS | (_capi.ORT_FAST_MATH if fast else 0)))
    eng.ctx.synchronize()
    assert torch.equal(torch.nan_to_num(xf), torch.nan_to_num(xv[-1])) and torch.equal(torch.nan_to_num(yf), torch.nan_to_num(yv[-1]))
    del xf, yf
    for b in (9, 19):                                            # bundle b repeats bundle b - 9
        a0, a1 = (b - 9) * rpb, b * rpb
        assert torch.equal(torch.nan_to_num(xv[:, a1:a1 + rpb]), torch.nan_to_num(xv[:, a0:a0 + rpb]))
        assert torch.equal(torch.nan_to_num(yv[:, a1:a1 + rpb]), torch.nan_to_num(yv[:, a0:a0 + rpb]))
    last = xv[S - 2:, (nb - 1) * rpb:].view(2, k, k)
    assert torch.equal(torch.nan_to_num(last), torch.nan_to_num(-last.flip(-1)))
    idx = np.unique(np.concatenate([np.arange(0, N, 7000003), [N - 1, N - k, (nb - 1) * rpb]]))
    ti = torch.from_numpy(idx).to(dev)
    sxv = xv[:, ti].cpu().numpy(); syv = yv[:, ti].cpu().numpy(); sst = st[ti].cpu().numpy()
    for b in range(nb):
        sel = (idx // rpb) == b
        if not sel.any():
            continue
        j = idx[sel] % rpb
        bd = bundles[b]
        yy = axes[bd["yaxis_off"] + j // k]; xx = axes[bd["xaxis_off"] + j % k]
        sub = Prescription(pres.R[bd["system"]], pres.t[bd["system"]], pres.n[bd["system"]])
        ox, oy, os_ = oracle_engine.skew(sub, yy, xx, np.full(j.size, math.tan(bd["U"])), np.zeros(j.size), slopes=True, want_status=True)
        if fast:
            assert cm.rel_err(sxv[:, sel], ox, 1.0).max() <= 1e-12 and cm.rel_err(syv[:, sel], oy, 1.0).max() <= 1e-12
        else:
            assert np.array_equal(sxv[:, sel], ox, equal_nan=True) and np.array_equal(syv[:, sel], oy, equal_nan=True)
        assert np.array_equal(sst[sel] & 0xffff, os_)


def test_nan_and_inf_inputs(hip_engine, oracle_engine):
    """The domain's nulls: NaN and +-Inf in the launch data (heights, slopes) — the reference just computes with them.
    The reference-sequence policy reproduces the oracle's outputs bit for bit (NaN patterns, Inf, status); the fast
    policy agrees on every NaN pattern and status and on the finite values."""
    pres = Prescription.from_matrix(_ext(cm.cooke(), 77.40534796682427))
    specials = [math.nan, math.inf, -math.inf, 0.0, -0.0, 1e308, -1e308, 1e-320]
    y = [3.0]; x = [-2.0]; u = [0.05]; v = [-0.02]
    for val in specials:
        for slot in range(4):
            row = [3.0, -2.0, 0.05, -0.02]; row[slot] = val
            y.append(row[0]); x.append(row[1]); u.append(row[2]); v.append(row[3])
    y, x, u, v = (np.array(a) for a in (y, x, u, v))
    ox, oy, os_ = oracle_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    gx, gy, gs = hip_engine.skew(pres, y, x, u, v, slopes=True, want_status=True)
    assert np.array_equal(gs, os_) and np.array_equal(gx, ox, equal_nan=True) and np.array_equal(gy, oy, equal_nan=True)
    fast = ort.HipEngine(0, fast_math=True)
    fx, fy, fs = fast.skew(pres, y, x, u, v, slopes=True, want_status=True)
    assert np.array_equal(fs, os_)
    assert np.array_equal(np.isnan(fx), np.isnan(ox)) and np.array_equal(np.isnan(fy), np.isnan(oy))
    fin = np.isfinite(ox) & np.isfinite(oy)
    assert cm.rel_err(fx[fin], ox[fin], 1.0).max() <= 1e-12 and cm.rel_err(fy[fin], oy[fin], 1.0).max() <= 1e-12
    assert np.array_equal(np.isinf(fx), np.isinf(ox)) and np.array_equal(np.isinf(fy), np.isinf(oy))


def test_wavegrad_device_and_host(hip_engine, oracle_engine):
    """`wavegrad(eps, lambda)` (src/PupilSampling.jl:165-167) through `ort_wavegrad_f64`: on DEVICE-resident full_trace
    slabs (torch tensors, nothing copied) and on host arrays — both equal (eps * nu) / lambda elementwise, bit for bit."""
    import torch
    from opticalraytracing_jl_amd import _capi
    k = 50
    pres, bundles, axes = _dg_bundles(oracle_engine, k)
    nb, cap = len(bundles), 2 * k * k
    dev = torch.device("cuda:0")
    d_axes = torch.from_numpy(axes).to(dev)
    ex = torch.zeros((nb, cap), dtype=torch.float64, device=dev); ey = torch.zeros_like(ex)
    rho = torch.zeros_like(ex); th = torch.zeros_like(ex)
    cnt = torch.empty(nb, dtype=torch.int64, device=dev); rms = torch.empty(nb, dtype=torch.float64, device=dev)
    sysd = hip_engine.system(pres)
    torch.cuda.synchronize(dev)                                   # torch's fills (its stream) before the engine's launches (its own)
    _capi.check(hip_engine.ctx.lib.ort_full_trace_f64(hip_engine.ctx.h, sysd.h, nb, _capi.make_bundles(bundles), d_axes.data_ptr(), axes.size,
                                                      k, k, ex.data_ptr(), ey.data_ptr(), rho.data_ptr(), th.data_ptr(), cnt.data_ptr(),
                                                      rms.data_ptr(), _capi.ORT_DEVICE_PTRS))
    nu = torch.linspace(-0.21, -0.17, nb, dtype=torch.float64, device=dev)
    lam = 587.5618e-6
    hip_engine.ctx.synchronize(); torch.cuda.synchronize(dev)     # ... and the other way round
    gx, gy = hip_engine.wavegrad(ex, ey, cnt, nu, lam)
    hip_engine.ctx.synchronize()
    c = cnt.cpu().numpy()
    for b in range(nb):
        m = int(c[b])
        assert m > 0
        want_x = (ex[b, :m].cpu().numpy() * float(nu[b])) / lam
        want_y = (ey[b, :m].cpu().numpy() * float(nu[b])) / lam
        assert np.array_equal(gx[b, :m].cpu().numpy(), want_x) and np.array_equal(gy[b, :m].cpu().numpy(), want_y)
    hx, hy = hip_engine.wavegrad(ex.cpu().numpy(), ey.cpu().numpy(), c, nu.cpu().numpy(), lam)
    for b in range(nb):
        m = int(c[b])
        assert np.array_equal(hx[b, :m], gx[b, :m].cpu().numpy()) and np.array_equal(hy[b, :m], gy[b, :m].cpu().numpy())
    # and the host mirror's wavegrad of a RealRayError is the same two operations
    e = ort.full_trace(ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=hip_engine), 0.7, engine=hip_engine)
    wx, wy = ort.wavegrad(e)
    assert np.array_equal(wx, (e.x * e.nu) / ort.api.LAMBDA) and np.array_equal(wy, (e.y * e.nu) / ort.api.LAMBDA)


def test_lookback_fault_is_reported(oracle_engine):
    """The look-back route's guard (ort_kernels.hpp): when the context's ticket base and the device's ticket counter
    disagree — a host bookkeeping fault, forced here with the testing aid ort_ctx_test_skew_tickets — tiles would wait for
    predecessors that never run.  The call must FAIL (ORT_EHIP naming the look-back), not return ORT_OK with misplaced
    survivors; a device-pointer caller sees count = -1 and rms = NaN; the context works again once the base is right."""
    import torch
    from opticalraytracing_jl_amd import _capi
    eng = ort.HipEngine(0)
    pres, bundles, axes = _dg_bundles(oracle_engine, 40, fields=(0.0, 1.0), lines=(0,))
    good = eng.full_trace_grid(pres, bundles, axes, 40, 40, lookback=True)
    lib, h = eng.ctx.lib, eng.ctx.h
    _capi.check(lib.ort_ctx_test_skew_tickets(h, 1 << 40))
    with pytest.raises(_capi.OrtError) as e:
        eng.full_trace_grid(pres, bundles, axes, 40, 40, lookback=True)
    assert e.value.code == -3 and "hand-off fault" in str(e.value)
    # device-pointer (asynchronous) caller: poisoned results
    dev = torch.device("cuda:0")
    nb, cap = len(bundles), 2 * 40 * 40
    d_axes = torch.from_numpy(axes).to(dev)
    vec = [torch.zeros((nb, cap), dtype=torch.float64, device=dev) for _ in range(4)]
    cnt = torch.zeros(nb, dtype=torch.int64, device=dev); rms = torch.zeros(nb, dtype=torch.float64, device=dev)
    sysd = eng.system(pres)
    torch.cuda.synchronize(dev)                                   # torch's fills (its stream) before the engine's launches (its own)
    _capi.check(lib.ort_full_trace_f64(h, sysd.h, nb, _capi.make_bundles(bundles), d_axes.data_ptr(), axes.size, 40, 40,
                                       *(v.data_ptr() for v in vec), cnt.data_ptr(), rms.data_ptr(),
                                       _capi.ORT_DEVICE_PTRS | _capi.ORT_FT_LOOKBACK))
    eng.ctx.synchronize()
    assert (cnt.cpu().numpy() == -1).all() and np.isnan(rms.cpu().numpy()).all()
    # the one-call pipeline (host arrays in, error vectors back in ONE packed copy: its stages run with device pointers
    # internally) owes a host caller the same error code
    mats = cm.cooke()[None]
    R, t, n = (np.ascontiguousarray(mats[:, :, j]) for j in range(3))
    aa = np.ascontiguousarray(cm.COOKE_A[None]); hh = np.array([cm.COOKE_H]); ff = np.array([1.0])
    fo = (_capi.ort_first_order * 1)()
    bex, bey, brho, bth = (np.zeros((1, 2 * 64 * 32)) for _ in range(4))
    bc = np.zeros(1, dtype=np.int64); br = np.zeros(1)
    call = lambda: lib.ort_full_trace_batch_f64(h, 1, mats.shape[1], _capi.ptr(R), _capi.ptr(t), _capi.ptr(n), _capi.ptr(aa), _capi.ptr(hh),
                                                1, _capi.ptr(ff), 64, fo, _capi.ptr(bex), _capi.ptr(bey), _capi.ptr(brho), _capi.ptr(bth),
                                                _capi.ptr(bc), _capi.ptr(br), _capi.ORT_FT_LOOKBACK)
    assert call() == -3 and b"hand-off fault" in lib.ort_last_error()
    _capi.check(lib.ort_ctx_test_skew_tickets(h, -(1 << 40)))
    assert call() == 0 and bc[0] > 0 and np.isfinite(br[0])
    again = eng.full_trace_grid(pres, bundles, axes, 40, 40, lookback=True)
    for a, b in zip(again, good):
        assert a["count"] == b["count"] and np.array_equal(a["ex"], b["ex"]) and a["rms"] == b["rms"]


def test_fan_with_a_last_thickness(hip_engine, oracle_engine):
    """A prescription that does NOT end in image space (last thickness 3 mm): the fan's marginal ray takes its sag from
    the paraxial vertex (s_last, SeidelAberrations.jl:125-127, RayTracing.jl:93-95), every other ray sag(ray) = s_last -
    t[end] (:130,132); the caustic set (descending) re-traces the marginal ray like the others (MakieExtension.jl:369-371).
    With t[end] = 0 — every prescription of the reference's tests — the two rules coincide."""
    surf = cm.cooke(); surf[-1, 1] = 3.0
    pres = Prescription.from_matrix(surf)
    spec = [dict(system=0, layout_mode=0, y_marg=12.3, XP_t=-30.0, BFD=77.4)]
    for desc in (False, True):
        yg, eg = hip_engine.fan(pres, spec, 22, descending=desc)
        yo, eo = oracle_engine.fan(pres, spec, 22, descending=desc)
        assert cm.rel_err(yg, yo, 1.0).max() <= TOL and np.abs(eg - eo).max() <= 1e-10
    _, asc = oracle_engine.fan(pres, spec, 22)
    _, dsc = oracle_engine.fan(pres, spec, 22, descending=True)
    # same rays except the marginal one, whose focal-plane height differs by tan(u') * t[end]
    assert np.abs(asc[0, :-1] - dsc[0, :0:-1]).max() <= 1e-12 and abs(asc[0, -1] - dsc[0, 0]) > 1e-3


def test_small_problem_path_is_bit_identical_to_the_general_route(hip_engine):
    """Config 1 — the reference's own call, full_trace(system, H, 64) on the Cooke triplet — takes the small-problem
    route: ONE launch for solve + tables + aiming + bundle + axes (k_small_prepare), the grid trace, ONE launch for
    offsets + placement + sigma (k_ft_small_finish), one copy each way.  The general route (six setup launches, three
    finishing ones: ORT_NO_SMALL_PATH) runs the same device functions: first-order structs, counts, RMS and every error
    vector entry must be IDENTICAL, for spherical and aspheric prescriptions, vectors and statistics only."""
    import ctypes as C
    from opticalraytracing_jl_amd import _capi, workloads
    lib, h = hip_engine.ctx.lib, hip_engine.ctx.h
    M4, coef = workloads.double_gauss_aspheric(0)
    cases = [(cm.cooke()[None], None, None, cm.COOKE_A, cm.COOKE_H, (0.0, 0.7, 1.0), 64),
             (cm.tessar()[None], None, None, cm.TESSAR_A, cm.TESSAR_H, (0.0, 1.0), 64),
             (np.stack([workloads.double_gauss(l) for l in (0, 1, 2)]), None, None, cm.DG_A, cm.DG_H, (0.0, 1.0), 48),
             (M4[None, :, :3], M4[None, :, 3], coef[None], cm.DG_A, cm.DG_H, (0.0, 1.0), 64)]
    for mats, K, cf, a, hp, fields, k in cases:
        nsys, rows, _ = mats.shape
        R, t, n = (np.ascontiguousarray(mats[:, :, j]) for j in range(3))
        aa = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (nsys, rows - 1)))
        hh = np.full(nsys, float(hp)); ff = np.asarray(fields, dtype=np.float64)
        na, cap = nsys * len(fields), 2 * k * (k // 2)
        Kp = None if K is None else np.ascontiguousarray(K); cp = None if cf is None else np.ascontiguousarray(cf)
        res = {}
        for tag, extra in (("small", 0), ("general", _capi.ORT_NO_SMALL_PATH)):
            fo = (_capi.ort_first_order * nsys)()
            ex, ey, rho, th = (np.zeros((na, cap)) for _ in range(4))
            cnt = np.zeros(na, dtype=np.int64); rms = np.zeros(na)
            _capi.check(lib.ort_full_trace_layout_batch_f64(h, nsys, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n), _capi.ptr(Kp),
                                                            _capi.ptr(cp), 0 if cp is None else cp.shape[2], _capi.ptr(aa), _capi.ptr(hh),
                                                            len(ff), _capi.ptr(ff), k, fo, _capi.ptr(ex), _capi.ptr(ey), _capi.ptr(rho),
                                                            _capi.ptr(th), _capi.ptr(cnt), _capi.ptr(rms), hip_engine.base_flags | extra))
            c2 = np.zeros(na, dtype=np.int64); r2 = np.zeros(na)
            if K is None:
                _capi.check(lib.ort_spot_batch_f64(h, nsys, rows, _capi.ptr(R), _capi.ptr(t), _capi.ptr(n), _capi.ptr(aa), _capi.ptr(hh),
                                                   len(ff), _capi.ptr(ff), k, None, _capi.ptr(c2), _capi.ptr(r2), hip_engine.base_flags | extra))
            res[tag] = (bytes(fo), ex, ey, rho, th, cnt, rms, c2, r2)
        s_, g_ = res["small"], res["general"]
        assert s_[0] == g_[0]                                                    # first-order structs, byte for byte
        assert np.array_equal(s_[5], g_[5]) and (s_[5] > 0).all()
        for j in (1, 2, 3, 4, 6, 7, 8):
            assert np.array_equal(s_[j], g_[j]), j


def test_four_host_threads_four_contexts():
    """include/ort.h: "one ctx per (host thread, GPU)", `ort_last_error` per thread.  Four host threads, each with its OWN
    context (own stream, own scratch, own look-back words and tickets) on GPU 0, run concurrently for a few hundred calls
    each: `ort_full_trace_f64` on the default route, with ORT_FT_LOOKBACK, with ORT_FT_FUSED and statistics-only (the walk route), and
    `ort_spot_batch_f64` — every result bit-identical to the serial run of the same call; then all four fail a call at the
    same moment with their own argument and each must read ITS message back."""
    import threading
    from opticalraytracing_jl_amd import _capi, batch, workloads
    k = 160                                                        # 25,600 rays per bundle: 50 tiles, walk route for the statistics
    pres, bundles, axes = _dg_bundles(ort.default_engine(), k, fields=(0.0, 1.0), lines=(0, 2))   # 4 bundles of the extended systems
    mats = workloads.config5(None, ninst=6)

    def work(eng):
        full = eng.full_trace_grid(pres, bundles, axes, k, k)
        look = eng.full_trace_grid(pres, bundles, axes, k, k, lookback=True)
        fuse = eng.full_trace_grid(pres, bundles, axes, k, k, fused=True)
        stat = eng.full_trace_grid(pres, bundles, axes, k, k, stats_only=True)
        sb = batch.spot_batch(mats, cm.DG_A, cm.DG_H, (0.0, 1.0), 64, engine=eng)
        key = []
        for r in full + look + fuse:
            key += [r["ex"].tobytes(), r["ey"].tobytes(), r["rho"].tobytes(), r["theta"].tobytes(), r["count"], r["rms"]]
        key += [(r["count"], r["rms"]) for r in stat]
        key += [sb["rms"].tobytes(), sb["count"].tobytes(), sb["W040"].tobytes()]
        return key

    serial = work(ort.HipEngine(0))
    assert serial[4] > 0 and all(r[0] > 0 for r in serial if isinstance(r, tuple))
    nthreads, iters = 4, 60                                         # 4 x 60 x 4 = 960 concurrent C calls
    gate = threading.Barrier(nthreads)
    fails, msgs = [], [None] * nthreads

    def body(i):
        try:
            eng = ort.HipEngine(0)
            gate.wait()
            for it in range(iters):
                if work(eng) != serial:
                    fails.append((i, it, "result differs from the serial run")); return
            # per-thread error text: everyone fails at once with a different system index, then reads its own message
            gate.wait()
            one = np.zeros(1); xv = np.zeros((pres.rows - 1, 1))
            rc = eng.ctx.lib.ort_trace_skew_f64(eng.ctx.h, eng.system(pres).h, 1000 + i, 1, _capi.ptr(one), _capi.ptr(one), _capi.ptr(one),
                                                _capi.ptr(one), _capi.ptr(xv), _capi.ptr(xv.copy()), 1, None, 0)
            gate.wait()                                             # every thread has failed before anyone reads
            msgs[i] = (rc, eng.ctx.lib.ort_last_error().decode())
        except Exception as exc:                                    # noqa: BLE001 — reported by the assertion below
            fails.append((i, -1, f"{type(exc).__name__}: {exc}"))
            try:
                gate.abort()
            except Exception:                                       # noqa: BLE001
                pass

    ts = [threading.Thread(target=body, args=(i,)) for i in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a thread is stuck"
    assert not fails, fails[:3]
    for i, (rc, msg) in enumerate(msgs):
        assert rc == -1 and f"system index {1000 + i} " in msg, (i, rc, msg)
    cm.report(f"thread safety: {nthreads} host threads x {iters} rounds x 5 calls (full_trace default / look-back / fused / statistics-only, "
              f"spot_batch), own context each, all bit-identical to the serial run; ort_last_error per thread")


def test_f32_summary_walk_is_launch_shape_only():
    """Float32 summary-mode grid launches let a workgroup walk several consecutive tiles of a bundle (k_trace, SWALK: the
    table staged once per workgroup).  What a tile writes must not depend on it: 40 bundles x 716 x 716 rays (1,002 tiles each, a
    ragged last tile; 40,080 tiles -> two tiles per workgroup) in summary mode equal, bit for bit, the same launch with the
    history also requested (one tile per workgroup), in both policies."""
    import ctypes as C
    import torch
    from opticalraytracing_jl_amd import _capi
    dev = torch.device("cuda:0")
    k, nb = 716, 40
    M = _ext(cm.double_gauss(), 57.8)
    pres = Prescription.from_matrix(M)
    rng = np.random.default_rng(9)
    axes = np.concatenate([np.concatenate([np.linspace(-13 - b * 0.01, 13, k), np.linspace(-12, 12 + b * 0.01, k)]) for b in range(nb)]).astype(np.float32)
    bundles = [dict(system=0, stop=6, U=float(rng.uniform(-0.1, 0.1)), V=float(rng.uniform(-0.05, 0.05)), a_stop=10.0,
                    yaxis_off=2 * k * b, xaxis_off=2 * k * b + k) for b in range(nb)]
    N, S = nb * k * k, pres.rows - 1
    d_axes = torch.from_numpy(axes).to(dev)
    for policy in ("ieee", "fast"):
        eng = _engine(policy)
        sysd = eng.system(pres)
        res = []
        for hist in (False, True):
            bufs = [torch.full((N,), float("nan"), dtype=torch.float32, device=dev) for _ in range(4)]
            st = torch.zeros(N, dtype=torch.int32, device=dev)
            out = _capi.ort_grid_out_f32()
            out.xf, out.yf, out.xs, out.ys = (b.data_ptr() for b in bufs)
            out.status = st.data_ptr()
            if hist:
                xv = torch.empty((S, N), dtype=torch.float32, device=dev); yv = torch.empty_like(xv)
                out.xv, out.yv, out.ld = xv.data_ptr(), yv.data_ptr(), N
            torch.cuda.synchronize(dev)                           # the fills above ran on torch's stream, the trace runs on the engine's
            _capi.check(eng.ctx.lib.ort_trace_grid_f32(eng.ctx.h, sysd.h, nb, _capi.make_bundles(bundles), d_axes.data_ptr(), axes.size, k, k,
                                                       C.byref(out), eng.base_flags | _capi.ORT_DEVICE_PTRS))
            eng.ctx.synchronize()
            res.append([b.cpu().numpy() for b in bufs] + [st.cpu().numpy()])
            if hist:
                assert np.array_equal(xv[-1].cpu().numpy(), res[-1][0], equal_nan=True)
                del xv, yv
        for a, b in zip(*res):
            assert np.array_equal(a, b, equal_nan=True), policy
        assert np.isfinite(res[0][0]).mean() > 0.5 and (res[0][4] & (1 << 16)).any()


@pytest.mark.parametrize("rows,units,pad", [(40, 5, 3), (64, 8, 6)])
def test_deep_prescriptions_other_kernels(hip_engine, oracle_engine, rows, units, pad):
    """The kernels beside the skew trace on prescriptions of 40 and 64 rows (ORT_MAX_ROWS; tables of up to 63 loop iterations
    in LDS): meridional trace (plain and Layout dispatch), paraxial y-nu with clip and the ABCD product (bit-exact), the
    batched first-order solve + Seidel sums against the C oracle."""
    from opticalraytracing_jl_amd import _capi, batch
    from oracle import cpu
    rng = np.random.default_rng(100 + rows)
    for variant in ("sph", "mixed"):
        M, coef = _cooke_relay(units, variant, pad)
        asph = variant != "sph"
        pres = Prescription(M[:, 0], M[:, 1], M[:, 2], M[:, 3] if asph else None, coef[None] if asph else None)
        y = rng.uniform(-5, 5, 400); U = rng.uniform(-0.01, 0.01, 400)
        g = hip_engine.meridional(pres, y, U, layout_mode=asph)
        o = oracle_engine.meridional(pres, y, U, layout_mode=asph)
        assert (hip_engine.last_domain_error is None) == (oracle_engine.last_domain_error is None)
        for a, b in zip(g, o):
            assert np.array_equal(np.isnan(a), np.isnan(b)), (rows, variant)
            assert cm.rel_err(a, b, 1.0).max() <= TOL, (rows, variant, cm.rel_err(a, b, 1.0).max())
        assert np.isfinite(o[0][-1]).mean() > 0.9
    M, _ = _cooke_relay(units, "sph", pad)
    L = ort.Lens(M[:, :3].copy())
    k = L.M.shape[0]
    assert k >= rows - 2
    yy = rng.uniform(-5, 5, 300); ww = rng.uniform(-0.01, 0.01, 300)
    a_ap = rng.uniform(4.0, 12.0, k)
    for clip in (False, True):
        gp = hip_engine.paraxial(L.M[:, 0], L.M[:, 1], yy, ww, a_ap, clip)
        op = oracle_engine.paraxial(L.M[:, 0], L.M[:, 1], yy, ww, a_ap, clip)
        assert np.array_equal(gp[0], op[0], equal_nan=True) and np.array_equal(gp[1], op[1], equal_nan=True)
    assert np.array_equal(hip_engine.abcd(L.M[:, 0], L.M[:, 1]), oracle_engine.abcd(L.M[:, 0], L.M[:, 1]))
    # first-order solve + Seidel sums of the deep system (semi-diameters: the Cooke triplet's, unit after unit)
    Ms = M[:-1, :3].copy()                                       # without the image row: ends in image space (t[end] = 0)
    Ms[-1, 1] = 0.0
    nrow = Ms.shape[0]
    a = np.resize(np.concatenate([cm.COOKE_A, cm.COOKE_A[::-1]]), nrow - 1).astype(np.float64)
    a[np.isinf(Ms[1:, 0]) & (a > 12.0)] = 12.0
    fo = batch.first_order_arrays(hip_engine, Ms[None], a, cm.COOKE_H)
    ref = cpu.solve_aberrations(Ms, a, cm.COOKE_H)
    for key in ("f", "EBFD", "W040", "W131", "W222", "W311", "H"):     # (8 units: the chain ends collimated, f = -inf on both sides)
        assert fo[key][0] == ref[key] or abs(fo[key][0] - ref[key]) <= 1e-12 * max(1.0, abs(ref[key])), (key, fo[key][0], ref[key])
    assert int(fo["stop"][0]) == int(ref["stop"])


@pytest.mark.parametrize("policy", ["ieee", "fast"])
def test_full_trace_fused_route_is_bit_identical(policy):
    """ORT_FT_FUSED (include/ort.h): the second pass of full_trace (tile offsets, placement of both halves, squared deviations)
    runs inside the trace launch — workgroup i traces tile i and places tile i - (tiles per bundle + margin).  Same device
    functions as the default route: every output bit-identical — on a launch small enough that the trailing workgroups do all
    the placing (4 bundles of 50 tiles), on one where placements run beside traces (config 3's systems at 768^2: 9 bundles of
    1,152 tiles, lag 3,200), in Float32, with one bundle (falls back to the default route) and on repeated calls of different
    shapes through one context (the ready words are told apart by the epoch, the counters return to zero)."""
    from opticalraytracing_jl_amd import api, workloads
    eng = _engine(policy)

    def same(a, b):
        assert len(a) == len(b)
        for ra, rb in zip(a, b):
            assert ra["count"] == rb["count"] and ra["count"] > 0
            assert np.float64(ra["rms"]).tobytes() == np.float64(rb["rms"]).tobytes()
            for key in ("ex", "ey", "rho", "theta"):
                assert ra[key].tobytes() == rb[key].tobytes(), key

    pres, bundles, axes = _dg_bundles(ort.default_engine(), 160, fields=(0.0, 1.0), lines=(0, 2))
    ref_small = eng.full_trace_grid(pres, bundles, axes, 160, 160)
    # a second launch of the same shape with other contents: the same workspace slots, offsets and aggregates hold other
    # values on alternating calls — a line kept by a cache from the call before would show
    presb, bundlesb, axesb = _dg_bundles(ort.default_engine(), 160, fields=(0.7, 0.3), lines=(1, 0))
    ref_smallb = eng.full_trace_grid(presb, bundlesb, axesb, 160, 160)
    assert any(x["count"] != y["count"] for x, y in zip(ref_small, ref_smallb))
    for _ in range(6):
        same(eng.full_trace_grid(pres, bundles, axes, 160, 160, fused=True), ref_small)
        same(eng.full_trace_grid(presb, bundlesb, axesb, 160, 160, fused=True), ref_smallb)
    pres3, bundles3, axes3 = workloads.config3(api, 768, engine=ort.default_engine())
    ref_big = eng.full_trace_grid(pres3, bundles3, axes3, 768, 768)
    for _ in range(2):
        same(eng.full_trace_grid(pres, bundles, axes, 160, 160, fused=True), ref_small)
        same(eng.full_trace_grid(pres3, bundles3, axes3, 768, 768, fused=True), ref_big)
    same(eng.full_trace_grid(pres, bundles[:1], axes, 160, 160, fused=True), ref_small[:1])
    ref32 = eng.full_trace_grid(pres, bundles, axes, 160, 160, dtype=np.float32)
    same(eng.full_trace_grid(pres, bundles, axes, 160, 160, dtype=np.float32, fused=True), ref32)
    # through the one-call pipelines (solve -> aiming -> axes -> full_trace on the device): 3 prescriptions x 2 fields, 512 x 256
    # half pupils (256 tiles per bundle), the plain and the Layout (conic + polynomial) entry points
    from opticalraytracing_jl_amd import _capi, batch
    mats = workloads.config5(None, ninst=3)
    r0 = batch.full_trace_systems(mats, cm.DG_A, cm.DG_H, (0.0, 1.0), 512, engine=eng)[1]
    r1 = batch.full_trace_systems(mats, cm.DG_A, cm.DG_H, (0.0, 1.0), 512, engine=eng, flags=_capi.ORT_FT_FUSED)[1]
    same(r1, r0)
    M4, coef = cm.double_gauss_aspheric()
    l0 = batch.full_trace_systems(M4[None], cm.DG_A, cm.DG_H, (0.0, 0.7, 1.0), 512, engine=eng, coef=np.asarray(coef)[None])[1]
    l1 = batch.full_trace_systems(M4[None], cm.DG_A, cm.DG_H, (0.0, 0.7, 1.0), 512, engine=eng, coef=np.asarray(coef)[None],
                                  flags=_capi.ORT_FT_FUSED)[1]
    same(l1, l0)
    # ... and as an engine-wide choice: every pipeline of the engine takes the route
    engf = ort.HipEngine(0, fast_math=(policy == "fast"), fused_full_trace=True)
    same(batch.full_trace_systems(mats, cm.DG_A, cm.DG_H, (0.0, 1.0), 512, engine=engf)[1], r0)
    same(engf.full_trace_grid(pres3, bundles3, axes3, 768, 768), ref_big)


def test_fused_route_fault_is_reported():
    """ORT_FT_FUSED: a workgroup that waits for a bundle's offsets beyond its poll cap raises the fault word; the call returns
    ORT_EHIP to a host caller and count = -1 / rms = NaN to a device-pointer caller, never misplaced survivors.  Forced with the
    testing aid ort_ctx_test_fused_no_scan (no workgroup runs the scans); the same context then runs the same call correctly."""
    import torch
    from opticalraytracing_jl_amd import _capi
    eng = ort.HipEngine(0)
    lib, h = eng.ctx.lib, eng.ctx.h
    pres, bundles, axes = _dg_bundles(ort.default_engine(), 160, fields=(0.0, 1.0), lines=(0, 2))
    good = eng.full_trace_grid(pres, bundles, axes, 160, 160)
    _capi.check(lib.ort_ctx_test_fused_no_scan(h, 1))
    with pytest.raises(_capi.OrtError, match="hand-off fault"):
        eng.full_trace_grid(pres, bundles, axes, 160, 160, fused=True)
    dev = torch.device("cuda", 0)
    nb, rpb = len(bundles), 160 * 160
    d_axes = torch.from_numpy(axes).to(dev)
    vec = [torch.zeros((nb, 2 * rpb), dtype=torch.float64, device=dev) for _ in range(4)]
    cnt = torch.zeros(nb, dtype=torch.int64, device=dev); rms = torch.zeros(nb, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)
    _capi.check(lib.ort_full_trace_f64(h, eng.system(pres).h, nb, _capi.make_bundles(bundles), d_axes.data_ptr(), axes.size, 160, 160,
                                       *(v.data_ptr() for v in vec), cnt.data_ptr(), rms.data_ptr(),
                                       _capi.ORT_DEVICE_PTRS | _capi.ORT_FT_FUSED))
    eng.ctx.synchronize()
    assert (cnt.cpu().numpy() == -1).all() and np.isnan(rms.cpu().numpy()).all()
    _capi.check(lib.ort_ctx_test_fused_no_scan(h, 0))
    again = eng.full_trace_grid(pres, bundles, axes, 160, 160, fused=True)
    for a, b in zip(again, good):
        assert a["count"] == b["count"] and a["rms"] == b["rms"]
        for key in ("ex", "ey", "rho", "theta"):
            assert a[key].tobytes() == b[key].tobytes()
