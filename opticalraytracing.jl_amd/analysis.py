"""Consumers of the traced rays that the reference keeps next to the hot path (SURVEY §8f
rows 3-4): Seidel sums, the meridional fan `TSA` with its least-squares fit `SA`, and the caustic ray set.

These are O(rows) host arithmetic per system in the reference too (no per-ray loop), so they
stay host code here; every *trace* they need (paraxial, meridional fan) goes through the GPU
engine.  The batched, one-thread-per-system device form of the first-order solve + Seidel sums
for Monte-Carlo runs is `HipEngine.first_order` (csrc `k_first_order`).
Citations are into /root/reference/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import api
from .api import (DomainError, K_RAYS, LAMBDA, Layout, RealRay, System, surface_ray, surface_to_focus,
                  trace_chief_ray, trace_marginal_ray, transfer_real)


@dataclass
class Aberration:                                                      # Types.jl:143-167
    W040: float
    W131: float
    W222: float
    W220: float
    W311: float
    W020: float
    W111: float
    W220P: float
    W220M: float
    W220T: float
    spherical: np.ndarray
    coma: np.ndarray
    astigmatism: np.ndarray
    sagittal: np.ndarray
    distortion: np.ndarray
    axial: np.ndarray
    lateral: np.ndarray
    petzval: np.ndarray
    medial: np.ndarray
    tangential: np.ndarray
    lam: float
    field_sign: int
    system: object

    def __call__(self, rho, theta, H):                                 # SeidelAberrations.jl:61-76
        H = abs(H)
        if not H <= 1.0:
            raise DomainError(f"DomainError with {H}: Domain: |H| ≤ 1.0")
        if not 0.0 <= rho <= 1.0:
            raise DomainError(f"DomainError with {rho}: Domain: 0.0 ≤ ρ ≤ 1.0")
        H *= self.field_sign
        c = math.cos(theta)
        return (self.W040 * rho ** 4 + self.W131 * H * rho ** 3 * c + self.W222 * H ** 2 * rho ** 2 * c ** 2 +
                self.W220 * H ** 2 * rho ** 2 + self.W311 * H ** 3 * rho * c + self.W020 * rho ** 2 +
                self.W111 * H * rho * c)


def aberrations(surfaces, system=None, lam: float = LAMBDA, dn=None) -> Aberration:
    """Third-order wavefront coefficients from Seidel sums (SeidelAberrations.jl:6-59)."""
    if isinstance(surfaces, System) and system is None:                # :55-59
        system = surfaces
        surfaces = system.layout
    M = np.asarray(surfaces.M if isinstance(surfaces, Layout) else surfaces, dtype=np.float64)
    if dn is None:
        dn = np.zeros(M.shape[0])
    dn = np.asarray(dn, dtype=np.float64)
    marginal, chief, H = system.marginal, system.chief, system.H
    R = M[1:, 0]
    n = marginal.n
    y = surface_ray(marginal.y)
    yb = surface_ray(chief.y)
    nu, nub, u = marginal.nu, chief.nu, marginal.u
    j = np.arange(len(R))

    def delta(x, yv, i):                                               # :4
        return x[i + 1] / yv[i + 1] - x[i] / yv[i]

    A = np.array([nu[i] + n[i] * y[i] / R[i] for i in j])
    Ab = np.array([(H + A[i] * yb[i]) / y[i] for i in j])
    yD = np.array([y[i] * delta(u, n, i) for i in j])
    yd = np.array([y[i] * delta(dn, n, i) for i in j])
    Dn2 = np.array([(1 / n[i + 1]) ** 2 - (1 / n[i]) ** 2 for i in j])
    P = np.array([((1 / n[i + 1]) - (1 / n[i])) / R[i] for i in j])
    yj = y[: len(j)]
    ybj = yb[: len(j)]
    spherical = -A ** 2 * yD / (8 * lam)
    coma = -A * Ab * yD / (2 * lam)
    astigmatism = -Ab ** 2 * yD / (2 * lam)
    petzval = -H ** 2 * P / (4 * lam)
    sagittal = petzval + astigmatism / 2
    distortion = -Ab * (Ab ** 2 * yj * Dn2 - (H + Ab * yj) * ybj * P) / (2 * lam)
    axial = A * yd / (2 * lam)
    lateral = Ab * yd / lam
    medial = petzval + astigmatism
    tangential = petzval + 1.5 * astigmatism
    W040, W131, W222, W311 = spherical.sum(), coma.sum(), astigmatism.sum(), distortion.sum()
    W220P = petzval.sum()
    return Aberration(W040, W131, W222, W220P + 0.5 * W222, W311, axial.sum(), lateral.sum(), W220P,
                      W220P + W222, W220P + 1.5 * W222, spherical, coma, astigmatism, sagittal, distortion,
                      axial, lateral, petzval, medial, tangential, lam, int(np.sign(chief.y[-1])), system)


def _fan_spec(surfaces, system, engine):
    """What ort_fan_f64 needs of the aimed real rays: y_m = real_marginal.y[1] (SeidelAberrations.jl:121),
    XP_t = real_chief.z[end] - real_chief.z[end-1] (:120), and the prescription as the engine sees it."""
    rm = trace_marginal_ray(surfaces, system, engine=engine)
    rc = trace_chief_ray(surfaces, system, engine=engine)
    pres, layout_mode, _ = api._as_layout(surfaces)
    return rm, dict(system=0, layout_mode=layout_mode, y_marg=float(rm.y[0]), XP_t=float(rc.z[-1] - rc.z[-2])), pres


def TSA(surfaces, system=None, k_rays: int = K_RAYS, engine=None):
    """Transverse spherical aberration fan (SeidelAberrations.jl:116-137): the k_rays meridional rays, their
    extension to the exit pupil and to the paraxial focal plane in ONE launch of the device fan kernel
    (`ort_fan_f64`); the aiming of the real marginal and chief rays is the reference's serial FD-Newton."""
    if isinstance(surfaces, System) and system is None:
        system = surfaces
        surfaces = system.layout
    pm = system.marginal
    _, spec, pres = _fan_spec(surfaces, system, engine)
    spec["BFD"] = float(pm.z[-1] - pm.z[-2])                           # paraxial_BFD  :124
    y_XP, eps = api._eng(engine).fan(pres, [spec], k_rays)
    return y_XP[0], eps[0]


def caustic_rays(surfaces, system=None, k_rays: int = K_RAYS, engine=None):
    """The ray set behind the reference's caustic plot (ext/MakieExtension.jl:353-398, the numbers only):
    k_rays meridional rays y = range(y_marginal, y_marginal / k_rays, k_rays) at U = 0, each extended to the
    paraxial image plane (negative LSA, :373-377) or to the marginal focus (:378-381): one launch of the fan
    kernel for the end points, one of the meridional kernel for the polylines.  Returns dict: y0 [k],
    z_surf / y_surf [k][rows-1] (polyline through the surfaces, :386-387), zf (scalar), yf [k] (image-space
    end point, :391-392)."""
    if isinstance(surfaces, System) and system is None:
        system = surfaces
        surfaces = system.layout
    z = system.marginal.z
    real_marginal, spec, pres = _fan_spec(surfaces, system, engine)
    paraxial_BFD = z[-1] - z[-2]                                                    # :365
    marginal_focus = real_marginal.z[-1]                                            # :366
    marginal_BFD = marginal_focus - z[-2]                                           # :368
    to_paraxial = abs(marginal_focus) < abs(z[-1])                                  # :373
    zf = z[-1] if to_paraxial else marginal_focus
    spec["BFD"] = float(paraxial_BFD if to_paraxial else marginal_BFD)
    eng = api._eng(engine)
    _, yf = eng.fan(pres, [spec], k_rays, descending=True)                          # :377,381
    y = api.linrange(real_marginal.y[0], real_marginal.y[0] / k_rays, k_rays)       # :364
    rays = api.raytrace(surfaces, y, 0.0, RealRay, engine=engine)
    return {"y0": np.array([r.y[0] for r in rays]), "z_surf": np.array([r.z[:-1] for r in rays]),
            "y_surf": np.array([r.y[1:] for r in rays]), "zf": float(zf), "yf": yf[0],
            "to_paraxial_plane": bool(to_paraxial)}


def SA(y, eps, degree: int):                                           # SeidelAberrations.jl:139-146
    if degree % 2 == 0 or degree < 3:
        raise DomainError(f"DomainError with {degree}: Required: isodd(degree) && degree ≥ 3")
    y = np.asarray(y, dtype=np.float64)
    yp = y / y.max()
    A = np.column_stack([yp ** k for k in range(3, degree + 1, 2)])
    return np.linalg.lstsq(A, np.asarray(eps, dtype=np.float64), rcond=None)[0]
