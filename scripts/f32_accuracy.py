#!/usr/bin/env python3
"""GPU box: how far are the Float32 image-plane hits (BASELINE config 5's trace, FAST policy) from the Float64 ones?
50 perturbed Double-Gauss instances x 2 fields x 96^2 pupil.  python scripts/f32_accuracy.py [--lib build/variants/...so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    os.environ["ORT_HIP_LIB"] = os.path.join(ROOT, sys.argv[sys.argv.index("--lib") + 1])
import numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads
fast = ort.HipEngine(0, fast_math=True); ieee = ort.HipEngine(0)
mats = workloads.config5(None, ninst=50)
k = 96
x64, y64, s64 = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, (0.0, 1.0), k, engine=ieee)
for tag, eng in (("f32 fast", fast), ("f32 reference sequence", ieee)):
    x32, y32, s32 = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, (0.0, 1.0), k, engine=eng, dtype=np.float32)
    ok = (s64 == s32) & ((s64 & 0xffff) == mats.shape[1] + 1)
    dx = (x32.double() - x64)[ok].abs(); dy = (y32.double() - y64)[ok].abs()
    d = (dx * dx + dy * dy).sqrt()
    print(f"{tag}: rays {int(ok.sum())} of {ok.numel()} with identical status; |hit32 - hit64| max {float(d.max()):.3e} mm, "
          f"rms {float((d * d).mean().sqrt()):.3e} mm, status differs on {int((s64 != s32).sum())}", flush=True)
a = batch.spot_batch(mats, workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=64, engine=fast)
b = batch.spot_batch(mats, workloads.DG_A, workloads.DG_H, fields=(0.0, 1.0), k_rays=64, engine=fast, dtype=np.float32)
print(f"spot RMS f32 vs f64: max relative difference {float(np.max(np.abs(a['rms'] - b['rms']) / a['rms'])):.3e}; "
      f"count differs on {int((a['count'] != b['count']).sum())} of {a['count'].size} bundles")
