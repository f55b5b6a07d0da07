"""Synthetic workloads of BASELINE.json's configs (inputs only — prescriptions and bundle
descriptors; no compute happens here).

The reference ships no Double-Gauss, so one is authored here: 10 refracting
surfaces in 6 elements (f/3-style double Gauss, curved cemented interfaces) with a flat stop
plane between the two halves: object row + 10 surfaces + stop = 12 rows; full_trace appends
the image row -> 13 rows, S = 12 loop iterations per ray.  "Wavelengths" are index columns
(the reference has no wavelength axis, Q21): d / F / C lines.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

INF = math.inf

_DG_GLASS = {            # nd, nF, nC
    "A": (1.60738, 1.61486, 1.60414),
    "B": (1.62041, 1.62756, 1.61727),
    "C": (1.60342, 1.61462, 1.59875),
}
_DG_ROWS = [             # R, t, medium after the surface
    (INF, 0.0, None),            # object space
    (54.153, 8.747, "A"),
    (152.522, 0.5, None),
    (35.951, 14.0, "B"),
    (420.0, 3.777, "C"),
    (22.270, 14.253, None),
    (INF, 12.428, None),         # stop plane
    (-25.685, 3.777, "C"),
    (-420.0, 10.834, "B"),
    (-36.980, 0.5, None),
    (196.417, 6.858, "B"),
    (-67.148, 0.0, None),
]
DG_A = np.array([29.225, 28.141, 24.296, 21.297, 14.919, 10.229, 13.188, 16.468, 18.930, 21.311, 21.646])
DG_H = 24.0
DG_FIELDS = (0.0, 0.7, 1.0)      # the field values the reference's tests use (test/runtests.jl:367-369)


# BASELINE config 1: the reference's own test prescription (Cooke triplet, test/runtests.jl:19-29: [R t n]; clear
# semi-diameters :31-33; image height :35) — data, as the reference's tests hold it
COOKE = np.array([[INF, 0.0, 1.0], [37.40, 5.90, 1.61272], [-341.48, 12.93, 1.0], [-42.65, 2.50, 1.64769],
                  [36.40, 2.00, 1.0], [INF, 9.85, 1.0], [204.52, 5.90, 1.61272], [-37.05, 0.0, 1.0]])
COOKE_A = np.array([14.7, 14.7, 10.8, 10.8, 10.3, 11.6, 11.6])
COOKE_H = 21.248


def double_gauss(line: int = 0, gap_shift: float = 0.0) -> np.ndarray:
    """rows x 3 surface matrix [R t n]; line 0/1/2 = d/F/C; gap_shift moves the two air gaps
    around the stop in opposite directions (the "zoom position" of BASELINE config 4)."""
    rows = []
    for R, t, g in _DG_ROWS:
        n = 1.0 if g is None else _DG_GLASS[g][line]
        rows.append([R, t, n])
    M = np.array(rows)
    M[5, 1] += gap_shift
    M[6, 1] -= gap_shift
    return M


def double_gauss_aspheric(line: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """BASELINE config 3: conic + even polynomial terms on 4 of the surfaces.
    Returns (rows x 4 [R t n K], rows x 7 coefficient table, p(y) = sum c_j y^j)."""
    M = double_gauss(line)
    K = np.zeros(M.shape[0])
    coef = np.zeros((M.shape[0], 7))
    for row, (k, a4, a6) in {1: (-0.35, 2.0e-8, -1.0e-11), 5: (0.25, -1.5e-7, 4.0e-10),
                             7: (0.25, 1.5e-7, -4.0e-10), 11: (-0.6, -2.0e-8, 1.0e-11)}.items():
        K[row] = k
        coef[row, 4] = a4
        coef[row, 6] = a6
    return np.column_stack([M, K]), coef


def square_pupil_bundles(api, systems: Sequence, k: int, fields: Sequence[float] = DG_FIELDS,
                         apertures=DG_A) -> Tuple[object, List[dict], np.ndarray]:
    """Bundles for a list of solved systems x fields, full square pupil k x k per bundle
    (y in [y2, y1], x in [-y_EP, y_EP], axes Julia-`range`-like).

    Returns (Prescription of the extended systems, bundle dicts, axes).  The pupil box comes
    from the paraxial solve (EP height and position, chief slope); real-ray aiming is the
    reference's serial host step and is not part of the timed path."""
    from .engine import Prescription
    Rs, ts, ns, Ks, Cs, bundles, axes = [], [], [], [], [], [], []
    off = 0
    for si, system in enumerate(systems):
        focus = system.marginal.z[-1] - system.marginal.z[-2]
        pres = api.extended_prescription(system.layout, focus)
        Rs.append(pres.R[0]); ts.append(pres.t[0]); ns.append(pres.n[0])
        Ks.append(pres.K[0] if pres.K is not None else np.zeros_like(pres.R[0]))
        Cs.append(None if pres.coef is None else pres.coef[0])
        y_EP = abs(system.marginal.y[0])
        Ub = math.atan(system.chief.u[0])
        for H in fields:
            U = H * Ub
            c = -math.tan(U) * system.EP.t
            axes += [api.linrange(c + y_EP, c - y_EP, k), api.linrange(-y_EP, y_EP, k)]
            bundles.append(dict(system=si, stop=system.stop, U=U, V=0.0,
                                a_stop=float(abs(apertures[system.stop - 1])),
                                hprime=math.tan(U) * system.f, yaxis_off=off, xaxis_off=off + k))
            off += 2 * k
    coef = None
    if any(c is not None for c in Cs):
        nc = max(c.shape[1] for c in Cs if c is not None)
        coef = np.zeros((len(Cs), len(Rs[0]), nc))
        for i, c in enumerate(Cs):
            if c is not None:
                coef[i, :, :c.shape[1]] = c
    pres = Prescription(np.array(Rs), np.array(ts), np.array(ns), np.array(Ks), coef)
    return pres, bundles, np.concatenate(axes)


def config2(api, k: int = 1024, engine=None, gap_shift: float = 0.0):
    """Double-Gauss, 3 fields x 3 index columns, k x k pupil, Float64 (BASELINE config 2)."""
    systems = [api.solve(double_gauss(line, gap_shift), DG_A, DG_H, engine=engine) for line in (0, 1, 2)]
    return square_pupil_bundles(api, systems, k)


def config3(api, k: int = 2048, engine=None):
    """Same with 4 aspheric surfaces (BASELINE config 3)."""
    systems = []
    for line in (0, 1, 2):
        M4, coef = double_gauss_aspheric(line)
        lay = api.Layout(M4[:, 0], M4[:, 1], M4[:, 2], M4[:, 3], [c for c in coef])
        systems.append(api.solve(lay, DG_A, DG_H, engine=engine))
    return square_pupil_bundles(api, systems, k)


def config4(api, k: int = 512, nzoom: int = 32, engine=None, fields=(0.0, 0.5, 0.7, 0.85, 1.0),
            lines=(0, 1, 2, 1, 2)):
    """Zoom-lens sweep (BASELINE config 4): nzoom positions of the two air gaps around the stop
    x 5 fields x 5 index columns (d, F, C and two repeats standing in for further lines),
    k x k pupil.  One system per (zoom, line); bundles ordered (zoom, line, field)."""
    systems = []
    for z in range(nzoom):
        gap = -1.5 + 3.0 * z / max(1, nzoom - 1)
        for line in lines:
            systems.append(api.solve(double_gauss(line, gap), DG_A, DG_H, engine=engine))
    return square_pupil_bundles(api, systems, k, fields=fields)


def config5(api, k: int = 256, ninst: int = 10 ** 4, engine=None, seed: int = 12345):
    """Monte-Carlo tolerance run (BASELINE config 5): ninst perturbed Double-Gauss instances
    (sigma_R/R = 1e-3, sigma_t = 10 um, sigma_n = 1e-4), one on-axis + one full-field
    bundle each is left to the caller; returns the perturbed surface matrices [ninst][rows][3]."""
    rng = np.random.default_rng(seed)
    base = double_gauss(0)
    rows = base.shape[0]
    out = np.repeat(base[None], ninst, axis=0)
    finite = np.isfinite(base[:, 0])
    out[:, finite, 0] *= 1.0 + 1e-3 * rng.standard_normal((ninst, int(finite.sum())))
    thick = base[:, 1] != 0.0
    out[:, thick, 1] += 0.010 * rng.standard_normal((ninst, int(thick.sum())))
    glass = base[:, 2] != 1.0
    out[:, glass, 2] += 1e-4 * rng.standard_normal((ninst, int(glass.sum())))
    return out
