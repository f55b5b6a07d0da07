import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_engine():
    from oracle.cpu import OracleEngine
    return OracleEngine()


@pytest.fixture(scope="session")
def hip_engine():
    import opticalraytracing_jl_amd as ort
    return ort.default_engine()


def pytest_terminal_summary(terminalreporter):
    """Measured figures the parity tests record (rays past 1e-10 under the FAST policy, worst deviations): printed with -q too."""
    from tests import common
    if common.REPORT:
        terminalreporter.section("parity report")
        for line in common.REPORT:
            terminalreporter.write_line(line)
