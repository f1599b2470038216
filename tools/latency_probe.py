"""Single-geometry prediction latency (the ASE-calculator use case): host array in, host E/F out."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

def run(N, M, P=1):
    R, E, F = synth_geometries(N, M + 1000, seed=0)
    Rf = R.reshape(M + 1000, -1)
    ctx = _lib.Context(0)
    D = N * (N - 1) // 2
    tp = np.arange(D, dtype=np.int64)[None]
    xd, gd = ctx.desc_from_R(Rf[:M], N)
    rs = np.random.RandomState(0)
    ctx.predict_upload_model(xd, rs.normal(size=xd.shape), tp, 20.0, None)
    for B in (1, 2, 4, 6, 8, 64, 1000):
        q = np.ascontiguousarray(Rf[M:M + B])
        for _ in range(50):
            ctx.predict(q)
        ts = []
        for _ in range(500):
            t0 = time.perf_counter()
            ctx.predict(q)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print('N=%d M=%d B=%d: host-to-host latency median %.1f us, p10 %.1f, p90 %.1f' % (N, M, B, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)), flush=True)
    ctx.close()

def raw(N, M):
    """The same call without the NumPy / Context.predict wrapper: ctypes straight into gdml_predict (what a C caller pays)."""
    import ctypes as C
    R, E, F = synth_geometries(N, M + 8, seed=0)
    Rf = R.reshape(M + 8, -1)
    ctx = _lib.Context(0)
    tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
    xd, gd = ctx.desc_from_R(Rf[:M], N)
    ctx.predict_upload_model(xd, np.random.RandomState(0).normal(size=xd.shape), tp, 20.0, None)
    for fused in (1, 0):
        ctx.set_option('predict.fused', fused)
        for B in (1, 4):
            q = np.ascontiguousarray(Rf[M:M + B])
            Eo, Fo = np.empty(B), np.empty((B, 3 * N))
            args = (ctx._h, q.ctypes.data_as(C.c_void_p), B, None, None, Eo.ctypes.data_as(C.c_void_p), Fo.ctypes.data_as(C.c_void_p))
            fn = ctx._lib.gdml_predict
            for _ in range(200):
                fn(*args)
            ts = []
            for _ in range(2000):
                t0 = time.perf_counter()
                fn(*args)
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts) * 1e6
            print('raw gdml_predict N=%d M=%d B=%d predict.fused=%d: median %.1f us, p10 %.1f, p90 %.1f' %
                  (N, M, B, fused, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)), flush=True)
    ctx.close()


if __name__ == '__main__':
    run(21, 1000)
    run(9, 200)
    raw(21, 1000)
    raw(9, 200)
