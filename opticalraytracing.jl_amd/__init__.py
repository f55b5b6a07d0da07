"""opticalraytracing.jl_amd — MI355X-native batched ray-trace engine behind the
OpticalRayTracing.jl `raytrace` / `full_trace` / `TransferMatrix` API surface.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels (gfx950) + the C ABI (include/ort.h)
  _capi.py   ctypes binding of the C ABI
  engine.py  numpy-facing GPU engine
  api.py     host-side mirror of the reference API for this path
  analysis.py  Seidel sums, TSA / SA / caustic fan consumers (host arithmetic over device traces)
  workloads.py synthetic inputs of the BASELINE configs
  build.py   hipcc build of csrc/libort_hip.so
  dist.py    rank sharding + all-gather of image-plane hits (torch.distributed / RCCL)

The directory name carries a dot, so import it through the top-level shim module
`opticalraytracing_jl_amd` (repo root).
"""
from .api import (  # noqa: F401
    Aiming, Aspheric, Chief, DomainError, Layout, Lens, Marginal, ParaxialRay, Pupil, RayBasis,
    RealRay, RealRayError, RealRayT, Sagittal, Skew, Spherical, System, Tangential, TransferMatrix,
    VectorRealRay, compute_surfaces, extended_prescription, flatten, full_trace, full_trace_aim,
    full_trace_aim_batch, full_trace_batch, full_trace_grid, reversed_layout, incidences, linrange, linrange_batch, raytrace, reverse_transfer, sag, solve, surface_ray,
    surface_to_focus, trace_chief_ray, trace_marginal_ray, transfer, transfer_real, wavegrad, refract,
)
from .analysis import SA, TSA, Aberration, aberrations, caustic_rays  # noqa: F401
from .engine import HipEngine, Prescription, default_engine, set_default_engine  # noqa: F401

__version__ = "0.1.0"
