"""Predict-kernel probe: throughput of gdml_predict_dev for a few batch sizes."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

def run(N, M, B, reps=5):
    R, E, F = synth_geometries(N, M + B, seed=0)
    Rf = R.reshape(M + B, -1)
    ctx = _lib.Context(0)
    D = N * (N - 1) // 2
    tp = np.arange(D, dtype=np.int64)[None]
    xd, gd = ctx.desc_from_R(Rf[:M], N)
    ctx.train_upload(xd, gd, tp)
    rs = np.random.RandomState(0)
    ctx.predict_upload_model(xd, rs.normal(size=xd.shape), tp, 20.0, None)
    lib = ctx._lib
    dR, dE, dF = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ctx._check(lib.gdml_dev_alloc(ctx._h, B * 3 * N * 8, C.byref(dR)))
    ctx._check(lib.gdml_dev_alloc(ctx._h, B * 8, C.byref(dE)))
    ctx._check(lib.gdml_dev_alloc(ctx._h, B * 3 * N * 8, C.byref(dF)))
    Rq = np.ascontiguousarray(Rf[M:])
    ctx._check(lib.gdml_memcpy_h2d(ctx._h, dR, Rq.ctypes.data_as(C.c_void_p), Rq.nbytes))
    ctx.profile(True)
    ts = []
    for _ in range(reps):
        ctx._check(lib.gdml_predict_dev(ctx._h, dR, B, None, None, dE, dF))
        ts.append(ctx.phase_ms('predict')[0])
    kms, kn, kw = ctx.kernel_stat('predict')
    print('N=%d M=%d B=%d v1=%s: phase %.3f ms, kernel %.3f ms -> %.2e geoms/s, %.1f TF alg' % (
        N, M, B, os.environ.get('GDML_OPTIONS', '-'), min(ts), kms / kn, B / (min(ts) * 1e-3), kw / kms / 1e9), flush=True)
    ctx.close()

if __name__ == '__main__':
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
