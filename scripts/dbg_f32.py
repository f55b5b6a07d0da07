import ctypes as C, math, numpy as np
import opticalraytracing_jl_amd as ort
from opticalraytracing_jl_amd import _capi, Prescription
from oracle import cpu as oc
from tests import common as cm
eng = ort.default_engine()
M = np.vstack([cm.double_gauss(), [math.inf,0,1.0]]); M[-2,1]=57.8
pres = Prescription.from_matrix(M)
sysd = eng.system(pres)
k = 128
yax = np.linspace(-14, 14, k).astype(np.float32); xax = np.linspace(-14, 14, k).astype(np.float32)
axes = np.concatenate([yax, xax])
N, S = k * k, pres.rows - 1
L = oc.lib()
f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
R, t, n = f(M[:, 0]), f(M[:, 1]), f(M[:, 2])
oxv = np.empty((S, N), dtype=np.float32); oyv = np.empty((S, N), dtype=np.float32)
ost = np.empty(N, dtype=np.int32)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
L.orc_trace_skew_grid_f32(pres.rows, fp(R), fp(t), fp(n), None, None, 0, k, fp(yax), k, fp(xax),
                          np.float32(math.tan(0.1)), np.float32(0.0), fp(oxv), fp(oyv), N,
                          ost.ctypes.data_as(C.POINTER(C.c_int32)), 1)
for flags in (0, _capi.ORT_NO_LDS, _capi.ORT_FAST_MATH):
    xv = np.empty((S, N), dtype=np.float32); yv = np.empty((S, N), dtype=np.float32)
    st = np.empty(N, dtype=np.int32)
    out = _capi.ort_grid_out_f32()
    out.xv, out.yv, out.ld, out.status = xv.ctypes.data, yv.ctypes.data, N, st.ctypes.data
    b = _capi.make_bundles([dict(system=0, stop=0, U=0.1, V=0.0, yaxis_off=0, xaxis_off=k)])
    _capi.check(eng.ctx.lib.ort_trace_grid_f32(eng.ctx.h, sysd.h, 1, b, axes.ctypes.data, axes.size, k, k, C.byref(out), flags))
    print('flags',flags,'status mismatch',np.count_nonzero(st!=ost), 'hist st', np.bincount(st&0xffff), 'oracle', np.bincount(ost))
    ok = st==ost
    print('  max rel x', cm.rel_err(xv[:,ok],oxv[:,ok],1.0).max(), 'y', cm.rel_err(yv[:,ok],oyv[:,ok],1.0).max())
    bad=np.where(~ok)[0][:5]; print('  bad idx',bad, st[bad], ost[bad]); 
    if len(bad): print(xv[:,bad[0]], oxv[:,bad[0]])

print("---- per-surface comparison, ray 5000")
for flags in (0, _capi.ORT_NO_LDS):
    for trial in range(2):
        xv = np.zeros((S, N), dtype=np.float32); yv = np.zeros((S, N), dtype=np.float32)
        st = np.empty(N, dtype=np.int32)
        out = _capi.ort_grid_out_f32()
        out.xv, out.yv, out.ld, out.status = xv.ctypes.data, yv.ctypes.data, N, st.ctypes.data
        b = _capi.make_bundles([dict(system=0, stop=0, U=0.1, V=0.0, yaxis_off=0, xaxis_off=k)])
        _capi.check(eng.ctx.lib.ort_trace_grid_f32(eng.ctx.h, sysd.h, 1, b, axes.ctypes.data, axes.size, k, k, C.byref(out), flags))
        d = np.abs(xv-oxv).max(axis=1)
        print('flags',flags,'trial',trial,'max abs err per surface', d)
        j = 5000
        print('   x', xv[:,j]); 
        if trial==0 and flags==0: print('  ox', oxv[:,j])
        bad = np.argwhere(np.abs(xv-oxv) > 1e-3)
        print('   n bad', len(bad), 'first bad', bad[:5].tolist(), 'lanes', sorted(set((bad[:,1]%128).tolist()))[:20])
