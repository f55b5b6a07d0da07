"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

The path shards over independent units — (system, field, index column, pupil row): no ray
reads another ray's state (src/PupilSampling.jl:34-65 is a pure function).  Units are split
into contiguous, equal slabs in rank order, so concatenating rank outputs in rank order
reproduces the reference's append order (PupilSampling.jl:134-137).  There is no data-path
collective; the only exchange is the reassembly of image-plane hit points (one all-gather of
equal-size dense slabs) or, when only spot statistics are wanted, of per-bundle moments.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_bounds(n_units: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous slabs [lo, hi) in rank order; the first n_units % world ranks get one extra."""
    base, extra = divmod(n_units, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def row_segments(ulo: int, uhi: int, k: int) -> List[Tuple[int, int, int, int]]:
    """A contiguous slab [ulo, uhi) of the flattened (bundle, pupil row) list, k rows per bundle, as at most
    three uniform pieces (first bundle, number of bundles, first row, rows per bundle): a partial first bundle,
    whole bundles, a partial last bundle — each one grid launch (the grid entry points take one ny per call).
    Traced in this order they yield the slab's rays in the reference's order (y outer, x inner,
    src/PupilSampling.jl:123)."""
    if uhi <= ulo:
        return []
    b0, r0 = divmod(ulo, k)
    b1, r1 = divmod(uhi, k)                # the slab ends before row r1 of bundle b1
    if b0 == b1:
        return [(b0, 1, r0, r1 - r0)]
    segs = []
    if r0 > 0:
        segs.append((b0, 1, r0, k - r0)); b0 += 1
    if b1 > b0:
        segs.append((b0, b1 - b0, 0, k))
    if r1 > 0:
        segs.append((b1, 1, 0, r1))
    return segs


def shard(seq: Sequence, rank: int, world: int) -> Sequence:
    lo, hi = shard_bounds(len(seq), world)[rank]
    return seq[lo:hi]


def init_process_group(backend: str = "nccl"):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend)
    return dist


def barrier(device=None):
    """dist.barrier() that names this rank's GPU for the nccl (RCCL) backend — the plain call guesses the
    device from the rank and warns; gloo takes no device."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl" and device is not None:
        dist.barrier(device_ids=[int(device)])
    else:
        dist.barrier()


def allgather_hits(xf, yf, group=None, check_equal: bool = True):
    """All-gather per-rank hit slabs (image-plane x, y) into rank-ordered tensors: ONE collective over a
    packed [2, n] buffer when every rank holds the same n (what `shard_bounds` gives for a divisible unit
    count).  Unequal slabs (n_units % world != 0) cannot go through `all_gather_into_tensor`; they are detected
    (a 16-byte MIN/MAX all-reduce; skip it with check_equal=False when the caller has sharded evenly) and
    routed through the ragged gather (counts, then padded slabs), which reproduces the same rank order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = xf.numel()
    if yf.numel() != n:
        raise ValueError("allgather_hits: xf and yf differ in length")
    if check_equal and world > 1:
        mm = torch.tensor([n, -n], dtype=torch.int64, device=xf.device)
        dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
        if int(mm[0]) != -int(mm[1]):
            return allgather_ragged(xf, group=group), allgather_ragged(yf, group=group)
    packed = torch.stack([xf.reshape(-1), yf.reshape(-1)])            # [2, n]
    out = torch.empty((world * 2, n), dtype=packed.dtype, device=packed.device)   # concat along dim 0
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    out = out.view(world, 2, n)
    return out[:, 0, :].reshape(-1), out[:, 1, :].reshape(-1)


def allgather_ragged(values, group=None):
    """All-gather 1-D tensors of different lengths (compacted survivors): counts first, then
    padded slabs; returns the rank-ordered concatenation."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([values.numel()], dtype=torch.int64, device=values.device)
    counts = torch.empty(world, dtype=torch.int64, device=values.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    m = int(counts.max().item())
    pad = torch.zeros(m, dtype=values.dtype, device=values.device)
    pad[:values.numel()] = values.reshape(-1)
    out = torch.empty((world * m,), dtype=values.dtype, device=values.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, m)
    return torch.cat([out[r, :int(counts[r].item())] for r in range(world)])


def allreduce_moments(count, sx, sy, sxx, syy, group=None):
    """Spot statistics without moving hits: all-reduce (n, Σx, Σy, Σx², Σy²) per bundle."""
    import torch
    import torch.distributed as dist
    buf = torch.stack([count.double(), sx.double(), sy.double(), sxx.double(), syy.double()])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[0], buf[1], buf[2], buf[3], buf[4]


class RcclComm:
    """Native reassembly through the C ABI (ort_comm_* / ort_allgather_hits_f64): what a non-Python
    host (the Julia shim) uses instead of torch.distributed.  `unique_id()` on rank 0, ship the
    128 bytes to every rank by any means, then construct on every rank."""

    def __init__(self, engine, nranks: int, rank: int, id_bytes: bytes):
        import ctypes as C
        from . import _capi
        self._capi, self._C = _capi, C
        self.engine, self.nranks, self.rank = engine, nranks, rank
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        h = C.c_void_p()
        _capi.check(engine.ctx.lib.ort_comm_create(engine.ctx.h, nranks, rank, buf, C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _capi
        buf = C.create_string_buffer(128)
        _capi.check(_capi.load().ort_comm_unique_id(buf))
        return buf.raw

    @property
    def nranks_seen(self) -> int:
        return int(self.engine.ctx.lib.ort_comm_size(self.h))

    def allgather_hits(self, xf, yf):
        """xf, yf: CUDA float64 tensors of equal length -> rank-ordered (gx, gy) of nranks*len."""
        import torch
        n = xf.numel()
        gx = torch.empty(self.nranks * n, dtype=torch.float64, device=xf.device)
        gy = torch.empty_like(gx)
        self._capi.check(self.engine.ctx.lib.ort_allgather_hits_f64(self.h, xf.data_ptr(), yf.data_ptr(), n,
                                                                    gx.data_ptr(), gy.data_ptr()))
        self.synchronize()
        return gx, gy

    def allgather_hits_packed(self, hits, gathered=None, wait: bool = True):
        """hits: CUDA tensor [2, n] (x row, y row — the trace wrote into it) -> [nranks, 2, n]: ONE ncclAllGather
        on the communicator's stream, after the work queued on the engine's stream.  wait=False returns at once;
        the caller overlaps further traces and calls wait() / synchronize() before reading `gathered`."""
        import torch
        n = hits.shape[-1]
        if gathered is None:
            gathered = torch.empty((self.nranks, 2, n), dtype=hits.dtype, device=hits.device)
        fn = self.engine.ctx.lib.ort_allgather_hits_packed_f64 if hits.dtype == torch.float64 else \
            self.engine.ctx.lib.ort_allgather_hits_packed_f32
        self._capi.check(fn(self.h, hits.data_ptr(), n, gathered.data_ptr()))
        if wait:
            self.synchronize()
        return gathered

    def allgather_ragged(self, values, capacity: int):
        """values: CUDA float64 tensor of this rank's length -> (rank-ordered concatenation [sum counts], counts)."""
        import numpy as np
        import torch
        out = torch.empty(int(capacity), dtype=torch.float64, device=values.device)
        counts = np.zeros(self.nranks, dtype=np.int64)
        self._capi.check(self.engine.ctx.lib.ort_allgather_ragged_f64(self.h, values.data_ptr(), values.numel(), out.data_ptr(),
                                                                      int(capacity), counts.ctypes.data))
        self.synchronize()
        return out[:int(counts.sum())], counts

    def wait(self):
        """engine stream waits for the collectives issued so far (no host block)."""
        self._capi.check(self.engine.ctx.lib.ort_comm_wait(self.h))

    def wait_lag(self, lag: int):
        """engine stream waits for the collective issued `lag` calls before the latest one."""
        self._capi.check(self.engine.ctx.lib.ort_comm_wait_lag(self.h, int(lag)))

    def synchronize(self):
        self._capi.check(self.engine.ctx.lib.ort_comm_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.engine.ctx.lib.ort_comm_destroy(self.h)
            self.h = None
