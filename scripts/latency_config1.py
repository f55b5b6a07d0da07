"""GPU box: BASELINE config 1 — the reference's own call, full_trace(system, H, 64) on the Cooke triplet
(64 x 32 rays, 8 surfaces): end-to-end latency per call, GPU engine vs the C oracle on one host core."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import api
from oracle.cpu import OracleEngine
from tests import common as cm
hip, orc = ort.default_engine(), OracleEngine()
def bench(f, n=30):
    f(); f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return np.median(t) * 1e3
sg = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=hip)
so = ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=orc)
print("solve               : gpu %.3f ms   oracle %.3f ms" % (bench(lambda: ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=hip)),
                                                              bench(lambda: ort.solve(cm.cooke(), cm.COOKE_A, cm.COOKE_H, engine=orc))))
print("full_trace(H=1, 64) : gpu %.3f ms   oracle %.3f ms" % (bench(lambda: ort.full_trace(sg, 1.0, 64, engine=hip)),
                                                              bench(lambda: ort.full_trace(so, 1.0, 64, engine=orc))))
aim = ort.full_trace_aim(sg.layout, sg, 1.0, engine=hip)
print("  host-driven aiming: gpu %.3f ms" % bench(lambda: ort.full_trace_aim(sg.layout, sg, 1.0, engine=hip)))
print("  device aiming     : gpu %.3f ms" % bench(lambda: ort.full_trace_aim_batch([sg], [1.0], engine=hip)))
print("  grid stage only   : gpu %.3f ms   oracle %.3f ms" % (bench(lambda: ort.full_trace_grid(sg.layout, aim, 64, engine=hip)),
                                                              bench(lambda: ort.full_trace_grid(so.layout, aim, 64, engine=orc))))
systems = [sg] * 32
print("full_trace_batch 32 systems x 5 fields x 64x32: gpu %.3f ms total" % bench(lambda: ort.full_trace_batch(systems, [0, .25, .5, .75, 1.0], 64, engine=hip), 10))
e1 = ort.full_trace(sg, 1.0, 64, engine=hip); e2 = ort.full_trace(so, 1.0, 64, engine=orc)
print("RMS gpu %.9f oracle %.9f survivors %d %d" % (e1.RMS, e2.RMS, len(e1.x), len(e2.x)))
