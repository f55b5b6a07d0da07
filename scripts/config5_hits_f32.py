"""GPU box: BASELINE config 5 verbatim on ONE GPU — 10^4 perturbed Double-Gauss instances x 256 x 256 pupil,
Float32, image-plane hits only (8 B per ray: the all-gather payload), on-axis field; then one rank's share of
the 8-GPU run (1250 instances)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads
eng = ort.HipEngine(0, fast_math=True)
mats = workloads.config5(None, ninst=10000)
batch.image_hits(mats[:8], workloads.DG_A, workloads.DG_H, (0.0,), 64, engine=eng, dtype=np.float32)
for label, shard in (("all 10^4 instances on one GPU", None), ("rank 0 of 8 (1250 instances)", (0, 8))):
    for rep in range(3):
        t0 = time.perf_counter()
        xf, yf, st = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, (0.0,), 256, engine=eng, dtype=np.float32, shard=shard)
        dt = time.perf_counter() - t0
        rays = xf.numel()
        print(f"{label} rep {rep}: {rays:.3e} rays ({rays * 12:.3e} intersections) {dt * 1e3:.1f} ms wall -> "
              f"{rays * 12 / dt:.3e} intersections/s end to end; hits payload {rays * 8 / 1e9:.2f} GB ({xf.dtype})", flush=True)
        del xf, yf, st
# Float32 vs Float64 hits on a sample
a = batch.image_hits(mats[:16], workloads.DG_A, workloads.DG_H, (0.0,), 256, engine=eng, dtype=np.float32)
b = batch.image_hits(mats[:16], workloads.DG_A, workloads.DG_H, (0.0,), 256, engine=eng)
ok = ~torch.isnan(b[0]) & ~torch.isnan(a[0])
print("f32 vs f64 hits: max abs diff", float((a[0][ok].double() - b[0][ok]).abs().max()), "mm; status equal on",
      float((a[2] == b[2]).float().mean()))
