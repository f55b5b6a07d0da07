"""The C-ABI library loads and exports every symbol include/ort.h declares; the ctypes table
matches the header; without a usable GPU the product fails loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import pytest

from opticalraytracing_jl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ort.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ort_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 20
    lib = _capi.load()
    for n in names:
        assert hasattr(lib, n), f"libort_hip.so does not export {n}"
    assert sorted(_capi.SIGNATURES) == names
    assert lib.ort_version() == 401


def test_no_torch_types_in_abi():
    raw = open(os.path.join(ROOT, "include", "ort.h")).read()
    assert 'extern "C"' in raw
    src = re.sub(r"/\*.*?\*/", "", raw, flags=re.S)           # declarations only, comments dropped
    assert "torch" not in src.lower() and "at::" not in src and "std::" not in src


def test_bundle_struct_layout():
    assert C.sizeof(_capi.ort_bundle) == 72
    assert _capi.ort_bundle.U.offset == 8 and _capi.ort_bundle.yaxis_off.offset == 56
    assert C.sizeof(_capi.ort_grid_out_f64) == 64


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load()
    h = C.c_void_p()
    rc = lib.ort_ctx_create(0, None, C.byref(h))
    assert rc == -3 and not h.value                      # ORT_EHIP
    assert b"no CPU fallback" in lib.ort_last_error() or b"HIP" in lib.ort_last_error()
    import opticalraytracing_jl_amd as ort
    with pytest.raises(_capi.OrtError):
        ort.HipEngine(0)
    with pytest.raises(_capi.OrtError):
        # default engine = GPU -> must raise, never fall back
        ort.solve([[float("inf"), 0, 1.0], [50.0, 3.0, 1.5], [-50.0, 0.0, 1.0]], [10.0, 10.0], 5.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "opticalraytracing.jl_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|libort_oracle|oracle/|oracle\.cpu|#include\s+\"[^\"]*oracle)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(txt), f"{f} reaches into oracle/"


def test_plain_c_caller_builds_and_fails_loudly_without_gpu():
    """examples/cooke_full_trace.c — a C program with nothing but include/ort.h — compiles and links against
    the library; without a GPU it must stop at ort_ctx_create with the no-fallback error, not compute."""
    import subprocess
    import torch
    from tests import common as cm
    exe = cm.build_c_example()
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is covered by the gpu suite")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr
