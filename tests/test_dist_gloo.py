"""world_size-2 gloo test of the N>1 path: contiguous rank-ordered shards of the bundle list,
traced independently (oracle engine on CPU), reassembled with ONE all-gather of image-plane
hits — the result equals the single-process trace in the reference's append order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opticalraytracing_jl_amd import dist as odist


def test_shard_bounds():
    assert odist.shard_bounds(9, 2) == [(0, 5), (5, 9)]
    assert odist.shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    b = odist.shard_bounds(10 ** 4, 8)
    assert b[0][0] == 0 and b[-1][1] == 10 ** 4 and all(b[i][1] == b[i + 1][0] for i in range(7))
    assert odist.shard(list(range(9)), 1, 2) == [5, 6, 7, 8]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, k, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import opticalraytracing_jl_amd as ort
        from opticalraytracing_jl_amd import api, workloads
        from oracle.cpu import OracleEngine
        eng = OracleEngine()
        systems = [ort.solve(workloads.double_gauss(line), workloads.DG_A, workloads.DG_H, engine=eng) for line in (0, 1)]
        pres, bundles, axes = workloads.square_pupil_bundles(api, systems, k, fields=(0.0, 1.0))
        mine = odist.shard(bundles, rank, world)                 # contiguous slab, rank order
        res = eng.grid(pres, mine, axes, k, k, history=False)
        xf, yf = torch.from_numpy(res["xf"]), torch.from_numpy(res["yf"])
        odist.barrier(None)                                      # gloo: plain barrier (nccl names its GPU)
        gx, gy = odist.allgather_hits(xf, yf)                    # ONE collective
        kept = torch.from_numpy(res["xf"][(res["status"] >> 16) == 0])
        rag = odist.allgather_ragged(kept)
        n, sx, sy, sxx, syy = odist.allreduce_moments(torch.tensor([float(xf.numel())]), xf.sum()[None], yf.sum()[None],
                                                      (xf * xf).sum()[None], (yf * yf).sum()[None])
        if rank == 0:
            full = eng.grid(pres, bundles, axes, k, k, history=False)
            ok = np.array_equal(gx.numpy(), full["xf"]) and np.array_equal(gy.numpy(), full["yf"])
            ok_rag = np.array_equal(rag.numpy(), full["xf"][(full["status"] >> 16) == 0])
            ok_mom = abs(float(sx) - full["xf"].sum()) < 1e-9 and int(n) == full["xf"].size
            out_q.put((ok, ok_rag, ok_mom))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_allgather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 12, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == (True, True, True)
