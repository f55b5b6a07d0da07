"""GPU box: BASELINE config 4 on ONE GPU — 32 zoom positions x 5 index columns (160 systems) x 5 fields x
512 x 512 pupil = 2.1e8 rays, image-plane hits only (the payload of the multi-GPU all-gather)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, workloads
eng = ort.HipEngine(0, fast_math=True)
mats = np.array([workloads.double_gauss(line, -1.5 + 3.0 * z / 31) for z in range(32) for line in (0, 1, 2, 1, 2)])
fields = (0.0, 0.5, 0.7, 0.85, 1.0)
batch.image_hits(mats[:4], workloads.DG_A, workloads.DG_H, fields, 64, engine=eng)
for rep in range(2):
    t0 = time.perf_counter()
    xf, yf, st = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, fields, 512, engine=eng)
    dt = time.perf_counter() - t0
    rays = xf.numel()
    kept = float(((st >> 16) == 0).float().mean())
    print(f"{mats.shape[0]} systems x {len(fields)} fields x 512^2 = {rays:.3e} rays ({rays*12:.3e} intersections): {dt*1e3:.1f} ms wall "
          f"-> {rays*12/dt:.3e} intersections/s end to end; hits payload {rays*16/1e9:.2f} GB; kept by the stop filter {kept:.3f}")
# 8-way shard consistency on one GPU: slabs concatenated in rank order == the single launch
parts = [batch.image_hits(mats, workloads.DG_A, workloads.DG_H, fields, 64, engine=eng, shard=(r, 8))[0] for r in range(8)]
whole = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, fields, 64, engine=eng)[0]
print("8 rank-ordered shards == single launch:", torch.equal(torch.nan_to_num(torch.cat(parts)), torch.nan_to_num(whole)))
