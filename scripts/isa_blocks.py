#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel of the gfx950 assembly (no GPU needed).

  python scripts/isa_blocks.py <asm.s or 'build'> <mangled-name substring> [--min N] [--dump LABEL]

'build' compiles csrc/ort_hip.hip to build/asm/ort.s first.  For every block with >= N instructions prints the
counts by class: fp64 arithmetic (fma / mul / add), transcendental seeds (v_rcp / v_rsq, 4 issue slots each),
moves, selects, compares, other VALU, SALU, LDS, VMEM, and the VALU issue slots (VALU + 3 x transcendental)."""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    out = os.path.join(ROOT, "build", "asm", "ort.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fno-slp-vectorize",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "opticalraytracing.jl_amd", "csrc", "ort_hip.hip"), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    return out


def classify(op):
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_rsq_f32")):
        return "trans"
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_fma_f32", "v_fmac_f32",
                      "v_mul_f32", "v_add_f32", "v_sub_f32", "v_pk_")):
        return "fp"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "mov"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def blocks_of(txt, name):
    i = txt.index(name)
    i = txt.rindex("\n", 0, i) + 1
    j = txt.index(".Lfunc_end", i)
    out, cur, lab, note = [], [], "entry", ""
    for l in txt[i:j].split("\n"):
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", t)
        if m:
            out.append((lab, note, cur)); cur = []; lab = m.group(1); note = m.group(2).strip()
        elif t and not t.startswith((".", ";")) and not t.endswith(":"):
            cur.append(t.split(";")[0].strip())
    out.append((lab, note, cur))
    return out


def main():
    src = sys.argv[1]
    if src == "build":
        src = build()
    name = sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 12
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    txt = open(src).read()
    full = re.search(r"^(\S*" + re.escape(name) + r"\S*):", txt, flags=re.M).group(1)
    bl = blocks_of(txt, full + ":")
    print(full)
    tot = Counter()
    for lab, note, ins in bl:
        c = Counter(classify(x.split()[0]) for x in ins)
        tot.update(c)
        if dump == lab:
            print("\n".join(ins))
        if len(ins) >= mn:
            valu = sum(c[k] for k in ("fp", "trans", "mov", "sel", "cmp", "valu"))
            print(f"{lab:12s} n={len(ins):4d} slots={valu + 3 * c['trans']:4d} fp={c['fp']:3d} trans={c['trans']:2d} mov={c['mov']:3d} sel={c['sel']:3d} "
                  f"cmp={c['cmp']:2d} valu={c['valu']:3d} salu={c['salu']:3d} wait={c['wait']:2d} lds={c['lds']:2d} vmem={c['vmem']:2d}  {note[:60]}")
    print("total", dict(tot))
    m = re.search(re.escape(full) + r".*?\.vgpr_count:\s*(\d+)", txt, flags=re.S)
    md = txt[txt.index(".amdhsa_kernel " + full):] if (".amdhsa_kernel " + full) in txt else ""
    for key in ("next_free_vgpr", "next_free_sgpr", "scratch", "group_segment_fixed_size"):
        mm = re.search(r"\.amdhsa_" + key + r"\w*\s+(\d+)", md)
        if mm:
            print(key, mm.group(1), end="  ")
    print()


if __name__ == "__main__":
    main()
