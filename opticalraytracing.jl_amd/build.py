"""Build csrc/libort_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

-ffp-contract=off: the default arithmetic policy reproduces the reference loop one IEEE
operation per reference operation (Julia never fuses a*b+c); the fast policy spells its FMAs.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libort_hip.so")
SOURCES = ["ort_hip.hip"]
DEPS = ["ort_hip.hip", "ort_kernels.hpp", "ort_device.hpp", os.path.join("..", "..", "include", "ort.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-fno-slp-vectorize",
         "-Wall"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    cmd = [hipcc(), *FLAGS, "-o", OUT, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
