"""Placement of long-lived output buffers in HBM.

On MI355X the rate at which a streaming kernel writes a large buffer depends on WHERE the allocation came to lie: the history
kernel of BASELINE config 2 (two [12][9,437,184] Float64 arrays, 1.8 GB) runs at 0.71 to 0.85 of the HBM spec into eight
pairs of arrays allocated one after the other by ONE process on ONE GPU — stable for a given pair, the same kernel, the same
bytes (`scripts/history_placement_probe.py`, `profiles/r04_history_placement_probe.log`; the pure-store kernel of
`tools/store_alloc_probe.hip` shows the same: 5.8 to 6.7 TB/s).  A caller that keeps its output buffers for many launches
can therefore afford to allocate a few candidates, time its own launch into each and keep the best-placed one.
"""
from typing import Callable, List, Tuple


def best_placed(make: Callable[[], object], time_ms: Callable[[object], float], candidates: int = 6) -> Tuple[object, dict]:
    """`make()` allocates one set of device buffers (e.g. a tuple of torch tensors), `time_ms(bufs)` times the caller's launch
    into it (ms).  All `candidates` sets are alive together while they are timed (so that they are different places), the
    fastest is returned with a report {"candidates_ms": [...], "chosen": index}; the others are dropped by the caller's
    allocator once the references are gone."""
    if candidates <= 1:
        bufs = make()
        return bufs, {"candidates_ms": [], "chosen": 0, "note": "placement probing off"}
    sets: List[object] = [make() for _ in range(candidates)]
    ms = [float(time_ms(b)) for b in sets]
    best = min(range(candidates), key=lambda i: ms[i])
    chosen = sets[best]
    del sets
    return chosen, {"candidates_ms": ms, "chosen": best}
