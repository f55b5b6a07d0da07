"""Static binding check of julia/OpticalRayTracingHIP.jl against include/ort.h (CPU only, no Julia needed).

The shim cannot run in this container (no Julia runtime), and a drifted `ccall` signature corrupts memory silently on
first use.  So every `ccall((:name, LIB), Ret, (ArgTypes...), ...)` of the shim is parsed and compared — name, arity,
and the C width / kind of the return and of every argument — with the prototype of the same name parsed from the
header, and with the ctypes table the Python mirror loads the library with (`_capi.SIGNATURES`); the Julia `struct`s
that mirror C structs are compared field by field (order, type, offset, total size) with the header's definitions
and with the ctypes classes.  Reference side: the shim extends the exports of src/OpticalRayTracing.jl:6-50.
"""
import ctypes as C
import os
import re

import pytest

from opticalraytracing_jl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "OpticalRayTracingHIP.jl")
HDR = os.path.join(ROOT, "include", "ort.h")

# kind = (class, bytes): what crosses the ABI in that slot
C_KINDS = {"int": ("i", 4), "int32_t": ("i", 4), "int64_t": ("i", 8), "unsigned": ("u", 4), "size_t": ("u", 8),
           "double": ("f", 8), "float": ("f", 4)}
JL_KINDS = {"Cint": ("i", 4), "Int32": ("i", 4), "Int64": ("i", 8), "Clonglong": ("i", 8), "UInt32": ("u", 4), "Cuint": ("u", 4),
            "Csize_t": ("u", 8), "UInt64": ("u", 8), "Float64": ("f", 8), "Cdouble": ("f", 8), "Float32": ("f", 4), "Cfloat": ("f", 4),
            "Cstring": ("p", 8)}
CT_KINDS = {C.c_int: ("i", 4), C.c_int64: ("i", 8), C.c_uint: ("u", 4), C.c_size_t: ("u", 8), C.c_double: ("f", 8),
            C.c_float: ("f", 4), C.c_void_p: ("p", 8), C.c_char_p: ("p", 8)}


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def c_kind(t):
    t = t.strip()
    if "*" in t:
        return ("p", 8)
    t = re.sub(r"\bconst\b", "", t).strip()
    t = t.split()[0] if t.split() and t.split()[0] in C_KINDS else t
    # drop a trailing parameter name
    toks = t.split()
    while toks and toks[-1] not in C_KINDS:
        toks.pop()
    assert toks, f"unknown C type {t!r}"
    return C_KINDS[toks[-1]]


def header_prototypes(src=None):
    src = _strip_c_comments(open(HDR).read() if src is None else src)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(ort_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argk = [] if args in ("", "void") else [c_kind(a) for a in args.split(",")]
        protos[name] = (c_kind(ret), argk)
    return protos


def header_structs(src=None):
    """{struct name: [(field, kind)]} of every `typedef struct name { ... } name;` in the header."""
    src = _strip_c_comments(open(HDR).read() if src is None else src)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(ort_[a-z0-9_]+)\s*\{(.*?)\}\s*\1\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            base = decl.split()[0] if not decl.startswith("const") else decl.split()[1]
            for nm in decl[decl.index(base) + len(base):].split(","):
                nm = nm.strip()
                fields.append((nm.lstrip("*").strip(), ("p", 8) if nm.startswith("*") or "*" in base else C_KINDS[base]))
        out[m.group(1)] = fields
    return out


def jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Ptr", "Ref"):
        return ("p", 8)
    assert t in JL_KINDS, f"unknown Julia type {t!r}"
    return JL_KINDS[t]


def _split_top(s):
    """Split on commas that are not inside braces / parentheses."""
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def _balanced(src, i):
    """src[i] == '(' -> index just past its matching ')'."""
    depth = 0
    for j in range(i, len(src)):
        if src[j] == "(":
            depth += 1
        elif src[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise AssertionError("unbalanced parentheses in the shim")


def julia_ccalls(src=None):
    """[(name, ret kind, [arg kinds], n values passed)] for every ccall of the shim."""
    src = open(JL).read() if src is None else src
    src = re.sub(r"#[^\n]*", "", src)                               # line comments (no '#' inside the shim's strings near ccalls)
    calls = []
    for m in re.finditer(r"ccall\(\(:(ort_[a-z0-9_]+),\s*LIB\)\s*,", src):
        start = src.index("(", m.start())                            # the ccall's own parenthesis
        body = src[start + 1:_balanced(src, start) - 1]
        parts = _split_top(body)                                     # [(:name, LIB), Ret, (ArgTypes...), values...]
        assert parts[0].startswith("(:"), parts[0]
        ret, argt = parts[1], parts[2]
        assert argt.startswith("(") and argt.endswith(")"), f"{m.group(1)}: argument tuple not found: {argt[:40]}"
        types = _split_top(argt[1:-1])
        calls.append((m.group(1), jl_kind(ret), [jl_kind(t) for t in types], len(parts) - 3))
    return calls


def julia_structs(src=None):
    """{name: [(field, julia type)]} of the immutable structs of the shim."""
    src = open(JL).read() if src is None else src
    out = {}
    for m in re.finditer(r"(?:^|\n)struct\s+([A-Za-z0-9_]+)[^\n]*\n(.*?)\nend", src, flags=re.S):
        fields = []
        for line in m.group(2).split("\n"):
            line = re.sub(r"#.*", "", line)
            for decl in line.split(";"):
                decl = decl.strip()
                if "::" in decl:
                    nm, ty = decl.split("::")
                    fields.append((nm.strip(), ty.strip()))
        out[m.group(1)] = fields
    return out


def _layout(kinds):
    """C layout of a struct of scalar fields: (offsets, size) with natural alignment."""
    off, offs, align = 0, [], 1
    for _, size in kinds:
        off = (off + size - 1) // size * size
        offs.append(off); off += size; align = max(align, size)
    return offs, (off + align - 1) // align * align


MIRRORS = {"OrtBundle": ("ort_bundle", _capi.ort_bundle), "GridOut64": ("ort_grid_out_f64", _capi.ort_grid_out_f64),
           "OrtAimIn": ("ort_aim_in", _capi.ort_aim_in), "OrtAimOut": ("ort_aim_out", _capi.ort_aim_out),
           "OrtFirstOrder": ("ort_first_order", _capi.ort_first_order)}


def check_shim(jl_src=None, hdr_src=None):
    """Raises AssertionError naming the first mismatch.  Returns (number of ccalls, number of structs) checked."""
    protos = header_prototypes(hdr_src)
    calls = julia_ccalls(jl_src)
    assert len(calls) >= 15
    for name, ret, args, nvals in calls:
        assert name in protos, f"{name}: not declared in include/ort.h"
        cret, cargs = protos[name]
        assert ret == cret, f"{name}: return type {ret} != header {cret}"
        assert len(args) == len(cargs), f"{name}: {len(args)} argument types, header has {len(cargs)}"
        assert nvals == len(args), f"{name}: {nvals} values passed for {len(args)} argument types"
        for i, (a, c) in enumerate(zip(args, cargs)):
            assert a == c, f"{name}: argument {i + 1} is {a} in the shim, {c} in the header"
        # and the table the Python mirror binds with says the same
        res, at = _capi.SIGNATURES[name]
        assert CT_KINDS.get(res, ("p", 8)) == cret, name
        assert [CT_KINDS.get(t, ("p", 8)) for t in at] == cargs, f"{name}: _capi.SIGNATURES disagrees with the header"
    cstructs = header_structs(hdr_src)
    nstruct = 0
    for jname, fields in julia_structs(jl_src).items():
        if jname not in MIRRORS:
            continue
        cname, ct = MIRRORS[jname]
        cf = cstructs[cname]
        assert [f for f, _ in fields] == [f for f, _ in cf], f"{jname}: field order {fields} != {cname} {cf}"
        kinds = [jl_kind(t) for _, t in fields]
        assert kinds == [k for _, k in cf], f"{jname}: field types differ from {cname}"
        offs, size = _layout(kinds)
        assert size == C.sizeof(ct), f"{jname}: {size} bytes, ctypes {cname} has {C.sizeof(ct)}"
        assert offs == [getattr(ct, f).offset for f, _ in cf], f"{jname}: field offsets differ from {cname}"
        nstruct += 1
    return len(calls), nstruct


def test_every_ccall_and_struct_matches_the_header():
    ncalls, nstruct = check_shim()
    assert ncalls >= 20 and nstruct >= 2


def test_header_prototypes_cover_the_ctypes_table():
    """The same comparison for every export, not only those the shim calls: header prototype == _capi.SIGNATURES."""
    protos = header_prototypes()
    assert sorted(protos) == sorted(_capi.SIGNATURES)
    for name, (cret, cargs) in protos.items():
        res, at = _capi.SIGNATURES[name]
        assert CT_KINDS.get(res, ("p", 8)) == cret, name
        assert [CT_KINDS.get(t, ("p", 8)) for t in at] == cargs, name
    for cname, ct in (("ort_bundle", _capi.ort_bundle), ("ort_aim_in", _capi.ort_aim_in), ("ort_aim_out", _capi.ort_aim_out),
                      ("ort_fan_in", _capi.ort_fan_in), ("ort_first_order", _capi.ort_first_order),
                      ("ort_grid_out_f64", _capi.ort_grid_out_f64), ("ort_grid_out_f32", _capi.ort_grid_out_f32)):
        cf = header_structs()[cname]
        assert [f for f, _ in cf] == [f for f, _ in ct._fields_], cname
        offs, size = _layout([k for _, k in cf])
        assert size == C.sizeof(ct) and offs == [getattr(ct, f).offset for f, _ in cf], cname


@pytest.mark.parametrize("old,new,what", [
    ("(Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},\n         Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, UInt32)",
     "(Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},\n         Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, UInt32)",
     "ort_trace_skew_f64: argument 4"),
    ("system::Int32; stop::Int32", "system::Int64; stop::Int32", "OrtBundle"),
    ("(Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, UInt32), ctx().h, 1, length(τ), τ, ϕ, M, 0))",
     "(Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, UInt32), ctx().h, 1, length(τ), τ, ϕ, M, 0))",
     "ort_abcd_f64"),
    ("xs::Ptr{Float64}; ys::Ptr{Float64}; status::Ptr{Int32}", "ys::Ptr{Float64}; xs::Ptr{Float64}; status::Ptr{Int32}", "GridOut64"),
])
def test_the_check_fails_when_the_shim_drifts(old, new, what):
    """Mutation check: one changed argument type / struct field in the .jl text must fail the comparison."""
    src = open(JL).read()
    assert old in src, "the mutation anchor moved: update this test"
    with pytest.raises(AssertionError) as e:
        check_shim(jl_src=src.replace(old, new, 1))
    assert what.split(":")[0] in str(e.value)
