"""Build-side guard (no GPU): the generated gfx950 code of the FAST trace kernels keeps its hot arms free of
register copies and spills (scripts/isa_lint.py).  One hipcc -S of the library, ~1.5 min."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_hot_arms_have_no_copies_or_spills():
    env = dict(os.environ)
    env["PATH"] = env.get("PATH", "") + os.pathsep + "/opt/rocm/bin"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_lint.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("-> ok") == 3, r.stdout
