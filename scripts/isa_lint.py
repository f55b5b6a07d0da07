#!/usr/bin/env python3
"""Build-side guard for the hot loop of the FAST trace kernels (runs here, no GPU): compiles the library to
gfx950 assembly and checks, for the history and the summary kernel (and the even-asphere build's statistics kernel),
  * the centre-form sphere arms (the blocks with exactly 4 v_rsq_f64 and no v_rcp_f64) carry no v_mov_b64,
  * no other block of the hot surface loop is a pure copy block (>= 10 v_mov_b64 in <= 20 instructions),
  * no scratch (spill) instruction in the centre-form sphere and flat arms of the hot surface loop (the blocks laid out
    before its first polynomial arm — 12 transcendental seeds per lane pair; the grouped general-form / conic /
    polynomial arms and the cold MATH_IEEE retrace may park a few registers).
The register coalescer's outcome is sensitive to the shape of the class dispatch in surface_step_n:
run this after touching it.   python scripts/isa_lint.py"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"history": "_ZN3ort7k_traceIdLi1ELi0ELb1ELb1ELb0ELi0ELi2EEEvNS_11TraceParamsIT_EE",
           "summary": "_ZN3ort7k_traceIdLi1ELi0ELb1ELb0ELb1ELi0ELi2EEEvNS_11TraceParamsIT_EE",
           # the even-asphere build (ARMS_EVEN), statistics-only full_trace (config 3): its sphere arms may end in the copy
           # of one ray's (x, y) for the stop capture, nothing more
           "even_stats": "_ZN3ort7k_traceIdLi1ELi2ELb1ELb0ELb0ELi4ELi2EEEvNS_11TraceParamsIT_EE"}   # (FT_WALK)
MOV_ALLOWANCE = {"even_stats": 4}      # the tile-walking kernel: the ray state lives across the tile loop, its sphere arms end in two merge copies per ray
with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "build") if os.path.isdir(os.path.join(ROOT, "build")) else None) as td:
    asm = os.path.join(td, "ort.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fno-slp-vectorize",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "opticalraytracing.jl_amd", "csrc", "ort_hip.hip"), "-o", asm], check=True,
                   stderr=subprocess.DEVNULL)
    txt = open(asm).read()
bad = 0
for tag, name in KERNELS.items():
    i = txt.index(name + ":"); j = txt.index(".Lfunc_end", i)
    blocks, cur, lab, hdr = [], [], "entry", None
    for l in txt[i:j].split("\n"):
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", t)
        if m:
            blocks.append((lab, hdr, cur)); cur = []; lab = m.group(1)
            h = re.search(r"Header=BB(\d+_\d+)", m.group(2))      # LLVM's loop annotation of the block
            hdr = ".LBB" + h.group(1) if h else (lab if "Loop Header" in m.group(2) else None)
        elif t and not t.startswith((".", ";")):
            cur.append(t.split(";")[0].strip())
    blocks.append((lab, hdr, cur))
    n_inst = sum(len(b) for _, _, b in blocks)
    arms = [(lab, h, b) for lab, h, b in blocks if sum(x.startswith("v_rsq_f64") for x in b) == 4 and not any(x.startswith("v_rcp_f64") for x in b)]
    # header of the loop the fast arms sit in = the hot surface loop (an arm that IS the loop's header block carries the annotation
    # on its following lines — nested loops of the tile-walking kernels —: take it from the arm that names its header)
    hot = next((h for _, h, _ in arms if h), None)
    for k_, (lab_, h_, b_) in enumerate(blocks):
        if lab_ == hot:
            blocks[k_] = (lab_, hot, b_)
    hot_blocks = [(lab, b) for lab, h, b in blocks if h == hot]
    seeds = lambda b: sum(x.startswith(("v_rsq_f64", "v_rcp_f64")) for x in b)
    first_poly = next((i for i, (_, b) in enumerate(hot_blocks) if seeds(b) >= 12), len(hot_blocks))
    scratch_hot = sum(1 for _, b in hot_blocks[:first_poly] for x in b if "scratch_" in x)
    scratch_all = sum(1 for _, _, b in blocks for x in b if "scratch_" in x)
    arm_movs = [sum(x.startswith("v_mov_b64") for x in b) for _, _, b in arms]
    arm_valu = [sum(x.startswith("v_") for x in b) for _, _, b in arms]
    copy_blocks = [lab for lab, b in hot_blocks if len(b) <= 20 and sum(x.startswith("v_mov_b64") for x in b) >= 10]
    # the general (grouped) arm keeps its own merge copies; they must not sit on the sphere / flat path:
    # heuristically, at most one pure copy block may remain in the hot loop.  Spills are tolerated only outside it
    # (the MATH_IEEE retrace of a wave that left the fast forms' domain is cold code).
    ok = scratch_hot == 0 and len(arms) >= 2 and all(m <= MOV_ALLOWANCE.get(tag, 0) for m in arm_movs[:2]) and len(copy_blocks) <= 1
    print(f"{tag}: {n_inst} instructions ({sum(len(b) for _, b in hot_blocks)} in the hot loop), scratch in the hot loop's centre-form sphere / flat arms {scratch_hot} "
          f"(kernel {scratch_all}), sphere arms VALU {arm_valu[:2]} with v_mov_b64 {arm_movs[:2]}, pure copy blocks {copy_blocks} "
          f"-> {'ok' if ok else 'REGRESSION'}")
    bad += not ok
sys.exit(1 if bad else 0)
