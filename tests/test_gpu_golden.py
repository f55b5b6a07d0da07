"""The HIP path against the COMMITTED golden fixtures (tests/golden/*.json: 50-digit mpmath restatement of the
reference loop, see test_oracle_golden.py for provenance) — the same checks the CPU oracle passes, run through
the C ABI on the GPU.  IEEE policy at the oracle's bar (1e-12); FAST policy at the north-star bar (1e-10)."""
import pytest

import opticalraytracing_jl_amd as ort
from tests import test_oracle_golden as g

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["ieee", "fast"])
def engine(request, monkeypatch):
    if request.param == "fast":
        monkeypatch.setattr(g, "TOL", 1e-10)
    return ort.HipEngine(0, fast_math=(request.param == "fast"))


def test_skew_goldens_on_gpu(engine):
    g.test_skew_goldens(engine)


def test_skew_anchor_ray_on_gpu(engine):
    g.test_skew_anchor_ray(engine)


def test_meridional_goldens_on_gpu(engine):
    g.test_meridional_goldens(engine)


def test_paraxial_and_abcd_goldens_on_gpu(engine):
    g.test_paraxial_and_abcd_goldens(engine)


def test_full_trace_golden_on_gpu(engine):
    g.test_full_trace_golden(engine)
