"""BASELINE config 4 as the north star words it: 32 zoom positions x 5 index columns x 5 fields x 512^2 pupil,
the bundles sharded across the ranks (one process per GPU, no data-path collective), then ONE all-gather of
the image-plane hit points (RCCL over xGMI with the nccl backend) so that every rank holds all 2.1e8 hits
in the single-GPU order.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29544 \\
      scripts/zoom_sweep_dist.py [--pupil 512] [--zoom 32] [--check]
ORT_BENCH_BACKEND=gloo rehearses the same code with every rank on cuda:0 and the collective on CPU tensors."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import batch, dist as odist, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--pupil", type=int, default=512)
ap.add_argument("--zoom", type=int, default=32)
ap.add_argument("--check", action="store_true", help="rank 0 re-traces everything alone and compares the gathered hits")
args = ap.parse_args()
rank, world, local = odist.env_rank_world()
backend = os.environ.get("ORT_BENCH_BACKEND", "nccl")
ndev = torch.cuda.device_count()
devi = local % ndev
torch.cuda.set_device(devi)
dev = torch.device("cuda", devi)
dist = odist.init_process_group(backend) if world > 1 else None
eng = ort.HipEngine(devi, fast_math=True)
mats = np.array([workloads.double_gauss(line, -1.5 + 3.0 * z / max(1, args.zoom - 1))
                 for z in range(args.zoom) for line in (0, 1, 2, 1, 2)])
fields = (0.0, 0.5, 0.7, 0.85, 1.0)
k = args.pupil
nb_total = mats.shape[0] * len(fields)
if nb_total % world:
    raise SystemExit(f"{nb_total} bundles do not split evenly over {world} ranks (equal slabs keep the gather dense)")
batch.image_hits(mats[:2], workloads.DG_A, workloads.DG_H, fields, 32, engine=eng)       # warm-up: allocations, first launches
for rep in range(3):
    torch.cuda.synchronize(dev)
    if dist: odist.barrier(devi)
    t0 = time.perf_counter()
    xf, yf, st = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, fields, k, engine=eng, shard=(rank, world))
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    if dist:
        cdev = dev if backend == "nccl" else torch.device("cpu")
        gx, gy = odist.allgather_hits(xf.reshape(-1).to(cdev), yf.reshape(-1).to(cdev))
        torch.cuda.synchronize(dev); odist.barrier(devi)
    else:
        gx, gy = xf.reshape(-1), yf.reshape(-1)
    t2 = time.perf_counter()
    tt = torch.tensor([t1 - t0, t2 - t1], dtype=torch.float64)
    if dist:
        tt = tt.to(cdev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); tt = tt.cpu()
    if rank == 0:
        rays = nb_total * k * k
        print(f"rep {rep}: {world} rank(s), {rays:.3e} rays ({rays * 12:.3e} intersections): trace {tt[0] * 1e3:.2f} ms "
              f"(max over ranks) + all-gather {tt[1] * 1e3:.2f} ms of {rays * 16 / 1e9:.2f} GB -> "
              f"{rays * 12 / float(tt.sum()):.3e} intersections/s with the hits reassembled on every rank", flush=True)
if args.check and rank == 0:
    wx, wy, _ = batch.image_hits(mats, workloads.DG_A, workloads.DG_H, fields, k, engine=eng)
    same = (torch.equal(torch.nan_to_num(gx.to(dev)), torch.nan_to_num(wx.reshape(-1))) and
            torch.equal(torch.nan_to_num(gy.to(dev)), torch.nan_to_num(wy.reshape(-1))))
    print("gathered hits == single-GPU trace, bit for bit:", same, flush=True)
    if not same:
        raise SystemExit(1)
if dist:
    odist.barrier(devi); dist.destroy_process_group()
