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


def _worker_config4(rank, world, port, out_q, nbundles=5):
    """BASELINE config 4's sharding at oracle size: (zoom position x index column) systems x fields x pupil rows,
    split (a) by bundle with an UNEVEN count (5 bundles over 2 ranks, 7 over 4 -> the ragged route) and (b) by pupil row
    (slabs that end inside a bundle); each rank traces its slab with the CPU oracle, the hits are
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
                   for gap in (-0.5, 0.5) for line in (0, 1)][:(nbundles + 1) // 2]
        pres, bundles, axes = workloads.square_pupil_bundles(api, systems, k, fields=(0.0, 1.0))
        bundles = bundles[:nbundles]                             # 5 bundles: not divisible by 2; 7: not by 4
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


def test_four_rank_config4_seven_bundles():
    """The same at world 4 with 7 bundles: bundle slabs of 2, 2, 2, 1 (ragged route) and row slabs of 11, 11, 10, 10 pupil
    rows, three of which start or end inside a bundle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_config4, args=(r, 4, port, q, 7)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert q.get(timeout=5) == (True, True, 7 * 36)


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
    # --gpus 8, the shape of the driver's scaling run: eight ranks rendezvous and agree
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--selftest-ranks"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"selftest": True, "world": 8, "rank_sum": 36, "spawned": True}
    # a failing rank's return code comes back through the launcher: --gpus 2 under a launcher that started ONE rank
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-ranks"],
                       capture_output=True, text=True, timeout=120, env=env1)
    assert r.returncode != 0 and "launcher started 1 rank" in (r.stderr + r.stdout)


def test_bench_launcher_deadline_kills_a_hung_rank_tree():
    """A rank stuck ahead of a collective keeps its peers inside it and the launcher's pipe open: `spawn_ranks` must not wait
    on that pipe.  With a 12 s deadline and rank 1 asleep, `bench.py --gpus 2` ends by itself with its own exit code (7),
    says why, and leaves no process of the tree behind."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ORT_BENCH_SELFTEST_HANG="1", ORT_BENCH_RANKS_TIMEOUT_S="12")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-ranks"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 7, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 60 and "did not finish within 12 s" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    time.sleep(1.0)
    left = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert "--selftest-ranks" not in left, left


def test_exchange_leg_is_surfaced_at_top_level():
    """The N > 1 line carries copies of the exchange leg's figures (extra.config4_allgather) at its top level — value, ms per
    step, the all-gather's implementation and rate, the ranks it spanned, verified — and a failed leg reads as failed."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    leg = {"value": 1.5e11, "ms_per_step": 16.8, "verified": True, "nranks_seen": 8,
           "allgather": {"impl": "ort_allgather_hits_packed_f64 (native RCCL)", "alone_GBps_per_rank": 210.0, "alone_ms": 15.9}}
    res = {}
    bench.surface_exchange_leg(res, leg)
    assert res["exchange_value"] == 1.5e11 and res["exchange_ms_per_step"] == 16.8 and res["exchange_nranks_seen"] == 8
    assert res["allgather_impl"].startswith("ort_allgather") and res["allgather_GBps_per_rank"] == 210.0
    assert res["exchange_verified"] is True and res["exchange_leg_ok"] is True
    bad = {}
    bench.surface_exchange_leg(bad, {"error": "RCCL rendezvous timeout"})
    assert bad["exchange_verified"] is False and bad["exchange_leg_ok"] is False and "timeout" in bad["exchange_error"]
    assert bad["exchange_value"] is None
    assert len({bench.EXIT_EXCHANGE_TIMEOUT, bench.EXIT_EXCHANGE_TIMEOUT_WITH_HEADLINE, bench.EXIT_EXCHANGE_FAILED, bench.EXIT_RANKS_TIMEOUT, 0}) == 5
