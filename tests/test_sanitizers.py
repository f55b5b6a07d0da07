"""AddressSanitizer + UndefinedBehaviorSanitizer run of the two native pieces that execute in the CPU container: the C oracle
(oracle/ort_oracle.c — the parity checker) and the host emulation of the device's per-surface step functions
(tests/emu/emu_device.cpp — csrc/ort_device.hpp compiled for the host).  Both are rebuilt with -fsanitize=address,undefined
into build/sanitize/ and the golden-vector, reference-vector and emulation suites are re-run against those builds in a child
interpreter with the sanitizer runtimes preloaded; any report fails the test.  (GPU sanitizers are not available on the pool.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "sanitize")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None or _runtime("libasan.so") is None or _runtime("libubsan.so") is None,
                    reason="gcc sanitizer runtimes not installed")
def test_oracle_and_emulation_under_asan_ubsan():
    os.makedirs(OUT, exist_ok=True)
    orc = os.path.join(OUT, "libort_oracle_san.so")
    emu = os.path.join(OUT, "libemu_device_san.so")
    subprocess.run(["gcc", *SAN, "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-Wall", "-std=gnu11", "-shared",
                    "-o", orc, os.path.join(ROOT, "oracle", "ort_oracle.c"), "-lm"], check=True)
    subprocess.run(["g++", *SAN, "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "tests", "emu", "stub"),
                    "-o", emu, os.path.join(ROOT, "tests", "emu", "emu_device.cpp")], check=True)
    env = dict(os.environ)
    env.update(ORT_ORACLE_LIB=orc, ORT_EMU_LIB=emu,
               LD_PRELOAD=_runtime("libasan.so") + ":" + _runtime("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86",      # (the interpreter's own arenas are not ours to audit)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=87",
               PYTHONDONTWRITEBYTECODE="1")
    suites = ["tests/test_oracle_golden.py", "tests/test_oracle_reference_vectors.py", "tests/test_device_emulation.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", *suites, "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail
    assert " passed" in r.stdout
    # the sanitized builds were the ones loaded
    probe = subprocess.run([sys.executable, "-c", "from oracle import cpu; from tests import emu; cpu.lib(); emu.lib(); "
                            "print(open('/proc/self/maps').read().count('_san.so') > 0)"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr
