"""Compare the MFMA predict kernel with the VALU bulk kernel and the wave kernel on the same inputs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

def run(N, M, B, seed=0):
    R, E, F = synth_geometries(N, M + B, seed=seed)
    Rf = R.reshape(M + B, -1)
    ctx = _lib.Context(0)
    D = N * (N - 1) // 2
    tp = np.arange(D, dtype=np.int64)[None]
    xd, gd = ctx.desc_from_R(Rf[:M], N)
    rs = np.random.RandomState(seed)
    ctx.predict_upload_model(xd, rs.normal(size=xd.shape), tp, 20.0, None)
    out = {}
    for name, opts in (('mfma', {}), ('bulk', {'predict.mfma': 0}), ('wave', {'predict.wave_only': 1})):
        ctx.set_option('predict.mfma', 1)
        ctx.set_option('predict.wave_only', 0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        out[name] = ctx.predict(Rf[M:])
    Fw = out['wave'][1]; Ew = out['wave'][0]
    sc = np.abs(Fw).max()
    for name in ('mfma', 'bulk'):
        dF = np.abs(out[name][1] - Fw).max() / sc
        dE = np.abs(out[name][0] - Ew).max() / np.abs(Ew).max()
        print('N=%d M=%d B=%d %s vs wave: max|dF|/max|F| = %.2e, max|dE|/max|E| = %.2e' % (N, M, B, name, dF, dE), flush=True)
    ctx.close()

if __name__ == '__main__':
    for a in ((6, 40, 256), (9, 100, 300), (21, 1000, 1000), (21, 333, 1000), (12, 1000, 257), (23, 100, 512), (7, 37, 300)):
        run(*a)
