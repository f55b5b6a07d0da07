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


def test_row_segments_cover_every_slab_in_order():
    """dist.row_segments: any contiguous slab of the flattened (bundle, pupil row) list = at most three uniform
    pieces that tile it exactly, in order — for every world size, incl. slabs inside one bundle."""
    for nb, k in ((5, 6), (800, 512), (3, 7), (1, 64)):
        for world in (1, 2, 3, 7, 8, 16):
            seen = []
            for lo, hi in odist.shard_bounds(nb * k, world):
                segs = odist.row_segments(lo, hi, k)
                assert len(segs) <= 3
                for b0, nbs, r0, nrows in segs:
                    assert 0 <= r0 and r0 + nrows <= k and nrows > 0 and nbs >= 1 and (nbs == 1 or (r0 == 0 and nrows == k))
                    for b in range(b0, b0 + nbs):
                        seen += [b * k + r for r in range(r0, r0 + nrows)]
            assert seen == list(range(nb * k)), (nb, k, world)


def _worker_config4(rank, world, port, out_q):
    """BASELINE config 4's sharding at oracle size: (zoom position x index column) systems x fields x pupil rows,
    split (a) by bundle with an UNEVEN count (5 bundles over 2 ranks -> the ragged route) and (b) by pupil row
    (15 rows each: a slab that ends inside a bundle); each rank traces its slab with the CPU oracle, the hits are
    reassembled with the package's collectives and compared with the single-process trace."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import opticalraytracing_jl_amd as ort
        from opticalraytracing_jl_amd import api, workloads
        from oracle.cpu import OracleEngine
        eng = OracleEngine()
        k = 6
        systems = [ort.solve(workloads.double_gauss(line, gap), workloads.DG_A, workloads.DG_H, engine=eng)
                   for gap in (-0.5, 0.5) for line in (0, 1)][:3]
        pres, bundles, axes = workloads.square_pupil_bundles(api, systems, k, fields=(0.0, 1.0))
        bundles = bundles[:5]                                    # 5 bundles: not divisible by 2
        full = eng.grid(pres, bundles, axes, k, k, history=False)
        # (a) bundle-level slabs, uneven: allgather_hits detects it and takes the ragged route
        mine = odist.shard(bundles, rank, world)
        res = eng.grid(pres, mine, axes, k, k, history=False)
        gx, gy = odist.allgather_hits(torch.from_numpy(res["xf"]), torch.from_numpy(res["yf"]))
        ok_a = np.array_equal(gx.numpy(), full["xf"], equal_nan=True) and np.array_equal(gy.numpy(), full["yf"], equal_nan=True)
        # (b) row-level slabs
        lo, hi = odist.shard_bounds(len(bundles) * k, world)[rank]
        xs, ys = [], []
        for b0, nbs, r0, nrows in odist.row_segments(lo, hi, k):
            seg = [dict(bundles[b], yaxis_off=bundles[b]["yaxis_off"] + r0) for b in range(b0, b0 + nbs)]
            r = eng.grid(pres, seg, axes, nrows, k, history=False)
            xs.append(r["xf"]); ys.append(r["yf"])
        gx, gy = odist.allgather_hits(torch.from_numpy(np.concatenate(xs)), torch.from_numpy(np.concatenate(ys)))
        ok_b = np.array_equal(gx.numpy(), full["xf"], equal_nan=True) and np.array_equal(gy.numpy(), full["yf"], equal_nan=True)
        if rank == 0:
            out_q.put((ok_a, ok_b, int(gx.numel())))
    finally:
        dist.destroy_process_group()


def test_two_rank_config4_uneven_bundle_and_row_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_config4, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == (True, True, 5 * 36)


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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE — the way the driver calls `--gpus 1` — starts its
    two ranks itself (a child `torch.distributed.run` on 127.0.0.1, never an exec), relays rank 0's JSON line and exits
    with the children's return code.  The ranks here only rendezvous over gloo on the CPU (--selftest-ranks): the launch
    path is what is under test, and it needs no GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-ranks"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"selftest": True, "world": 2, "rank_sum": 3, "spawned": True}
    # a failing rank's return code comes back through the launcher: --gpus 2 under a launcher that started ONE rank
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-ranks"],
                       capture_output=True, text=True, timeout=120, env=env1)
    assert r.returncode != 0 and "launcher started 1 rank" in (r.stderr + r.stdout)
