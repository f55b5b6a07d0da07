"""Independent 50-digit restatement of the reference hot path (mpmath).

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/ort_oracle.c (vector form, arbitrary
precision) so that the two restatements check each other; tests/golden/make_golden.py turns
its outputs into committed fixtures.  The values are restatement-derived, NOT produced by the
Julia reference (no Julia runtime exists in the build container).
Citations are into /root/reference/.
"""
from __future__ import annotations

from mpmath import mp, mpf, mpc, sqrt, tan, sin, cos, asin, atan, atan2, hypot, isnan, isinf, nan, inf, pi

mp.dps = 50

EPS = mpf(2) ** -26          # sqrt(eps(Float64)), src/RayTracing.jl:1


def _m(v):
    if hasattr(v, "tolist"):
        v = v.tolist()
    if isinstance(v, (list, tuple)):
        return [_m(a) for a in v]
    v = float(v)
    if v != v:
        return nan
    if v in (float("inf"), float("-inf")):
        return inf if v > 0 else -inf
    return mpf(v)


def _sign(R):
    return mpf(1) if R > 0 else (mpf(-1) if R < 0 else R)


def _finite(R):
    return not (isinf(R) or isnan(R))


def _poly(c, y):
    """p(y) as a power series; None = `zero` (src/Types.jl:21-27)."""
    if c is None:
        return mpf(0) if not isinstance(y, mpc) else mpc(0)
    acc = c[-1]
    for a in reversed(c[:-1]):
        acc = acc * y + a
    return acc


def dp_dy(c, y):
    """src/RayTracing.jl:103 — complex step."""
    if c is None:
        return mpf(0)
    return _poly(c, mpc(y, EPS)).imag / EPS


def sag3(y, x, u, v, R, K, c):
    """src/PupilSampling.jl:1-14"""
    if _finite(R):
        beta = R - y * u - x * v
        r2 = x ** 2 + y ** 2
        D = beta ** 2 - r2 * (1 + K + u ** 2 + v ** 2)
        if D >= 0:
            return r2 / (beta + _sign(R) * sqrt(D)) + _poly(c, y)
        return nan
    return mpf(0)


def tilt3(y, x, R, K, c):
    """src/PupilSampling.jl:16-19"""
    if isinf(R):
        D = inf
        return [(_sign(R) * a / D if not isnan(a) else nan) + dp_dy(c, a) for a in (x, y)]
    D = R ** 2 - (x ** 2 + y ** 2) * (1 + K)
    if D < 0:
        return [nan, nan]
    return [_sign(R) * a / sqrt(D) + dp_dy(c, a) for a in (x, y)]


def trace_skew(R, t, n, K, coef, y, x, U, V, slopes=False):
    """src/PupilSampling.jl:34-65.  coef: list (per row) of coefficient lists or None."""
    R, t, n = _m(R), _m(t), _m(n)
    rows = len(R)
    K = [mpf(0)] * rows if K is None else _m(K)
    coef = [None] * rows if coef is None else [None if (c is None or not any(float(a) != 0 for a in c)) else _m(list(c)) for c in coef]
    y, x = _m(y), _m(x)
    u, v = (_m(U), _m(V)) if slopes else (tan(_m(U)), tan(_m(V)))
    ts = list(t)
    k = [v, u, mpf(1)]
    nrm = sqrt(sum(a * a for a in k))
    k = [a / nrm for a in k]
    xv, yv = [], []
    bad = False
    for i in range(rows - 1):
        y = y + u * ts[i]
        x = x + v * ts[i]
        Rs, Ks, ps = R[i + 1], K[i + 1], coef[i + 1]
        s = nan if bad else sag3(y, x, u, v, Rs, Ks, ps)
        if isnan(s):
            bad = True
        y = y + s * u
        x = x + s * v
        ts[i] += s
        ts[i + 1] -= s
        if not bad:
            m = tilt3(y, x, Rs, Ks, ps) + [mpf(-1)]
            if not any(isnan(a) for a in m):
                nm = sqrt(sum(a * a for a in m))
                m = [a / nm for a in m]
                eta = n[i] / n[i + 1]                       # refract! :21-32
                g = -sum(a * b for a, b in zip(k, m))
                D = 1 - eta ** 2 * (1 - g ** 2)
                if D >= 0:
                    k = [eta * a + (eta * g - sqrt(D)) * b for a, b in zip(k, m)]
            u = k[1] / k[2]
            v = k[0] / k[2]
        xv.append(nan if bad else x)
        yv.append(nan if bad else y)
    return xv, yv


def sag2(y, U, R, K, c):
    """src/RayTracing.jl:75-88"""
    if _finite(R):
        beta = R - y * tan(U)
        y2 = y ** 2
        D = beta ** 2 - y2 * ((1 / cos(U)) ** 2 + K)
        if D >= 0:
            return y2 / (beta + _sign(R) * sqrt(D)) + _poly(c, y)
        return nan
    return mpf(0)


def trace_meridional(R, t, n, K, coef, layout_mode, y, U):
    """src/RayTracing.jl:145-169 -> (y[rows], U[rows], ts[rows])"""
    R, t, n = _m(R), _m(t), _m(n)
    rows = len(R)
    K = [mpf(0)] * rows if K is None else _m(K)
    coef = [None] * rows if coef is None else [None if (c is None or not any(float(a) != 0 for a in c)) else _m(list(c)) for c in coef]
    y, U = _m(y), _m(U)
    ts = list(t)
    ys, Us = [y], [U]
    for i in range(rows - 1):
        y = y + tan(U) * ts[i]
        Rs, Ks, ps = R[i + 1], K[i + 1], coef[i + 1]
        s = sag2(y, U, Rs, Ks, ps)
        y = y + s * tan(U)
        ts[i] += s
        ts[i + 1] -= s
        if isnan(y):
            theta = nan
        elif Ks == 0 and ps is None and not layout_mode:
            theta = asin(y / Rs) if _finite(Rs) else mpf(0)
        else:
            tl = (mpf(0) if isinf(Rs) else _sign(Rs) * y / sqrt(Rs ** 2 - y ** 2 * (1 + Ks))) + dp_dy(ps, y)
            theta = atan(tl)
        if isnan(theta) or isnan(U):
            U = nan
        else:
            sip = n[i] * sin(U + theta) / n[i + 1]
            U = asin(sip) - theta if abs(sip) <= 1 else nan
        ys.append(y)
        Us.append(U)
    return ys, Us, ts


def trace_paraxial(tau, phi, y, w):
    """src/RayTracing.jl:127-143 without clip."""
    tau, phi, y, w = _m(tau), _m(phi), _m(y), _m(w)
    ys, ws = [y], [w]
    for tq, ph in zip(tau, phi):
        y = y + w * tq if _finite(tq) else y
        w = w - y * ph
        ys.append(y)
        ws.append(w)
    return ys, ws


def abcd(tau, phi):
    """src/TransferMatrix.jl:4"""
    tau, phi = _m(tau), _m(phi)
    acc = None
    for tq, ph in reversed(list(zip(tau, phi))):
        Mi = [[mpf(1), tq], [-ph, 1 - tq * ph]]
        if acc is None:
            acc = Mi
        else:
            acc = [[sum(acc[i][k] * Mi[k][j] for k in range(2)) for j in range(2)] for i in range(2)]
    return acc


def full_trace_grid(R, t, n, K, coef, yaxis, xaxis, U, V, stop, a_stop, hprime):
    """src/PupilSampling.jl:121-146,169-173 for a System (not RayBasis)."""
    ex, ey, r, th = [], [], [], []
    for yi in yaxis:
        for xi in xaxis:
            xv, yv = trace_skew(R, t, n, K, coef, yi, xi, U, V)
            xf, yf = xv[-1], yv[-1]
            ri = hypot(xv[stop - 1], yv[stop - 1]) if not (isnan(xv[stop - 1]) or isnan(yv[stop - 1])) else nan
            if (not isnan(ri) and ri > a_stop) or isnan(xf) or isnan(yf):
                continue
            th.append(atan2(yv[stop - 1], xv[stop - 1]))
            ey.append(yf - _m(hprime))
            ex.append(xf)
            r.append(ri)
    m = len(ex)
    rmax = max(r)
    ey = ey + ey
    ex = ex + [-a for a in ex]
    rho = [a / rmax for a in r] * 2
    th = th + [pi - a for a in th]
    nn = len(ex)
    mux, muy = sum(ex) / nn, sum(ey) / nn
    rms = sqrt((sum((a - mux) ** 2 for a in ex) + sum((a - muy) ** 2 for a in ey)) / nn)
    return ex, ey, rho, th, rms, m
