"""Prediction contraction for large molecules: GEMM-pipeline path (predict_wide.hip) vs the wave kernel.
    python tools/predict_wide_probe.py N M P B   (P in {1, 3, 9, 27}: independent 3-cycles)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import synth_geometries
from sgdml_amd import _lib
from sgdml_amd.utils.desc import Desc

N, M, P, B = [int(v) for v in sys.argv[1:5]]
gens = []
for a in (3, 17, 30)[: {1: 0, 3: 1, 9: 2, 27: 3}[P]]:
    g = list(range(N)); g[a], g[a + 1], g[a + 2] = a + 1, a + 2, a
    gens.append(tuple(g))
perms = {tuple(range(N))}
front = list(perms)
while front:
    nxt = []
    for a in front:
        for g in gens:
            c = tuple(a[i] for i in g)
            if c not in perms:
                perms.add(c); nxt.append(c)
    front = nxt
perms = np.array(sorted(perms))
R, E, F = synth_geometries(N, M + B, seed=0)
ctx = _lib.Context(0)
xd, gd = ctx.desc_from_R(R[:M].reshape(M, -1), N)
tp = np.array([Desc.perm(p) for p in perms])
rs = np.random.RandomState(0)
ctx.predict_upload_model(xd, rs.normal(size=xd.shape), tp, 40.0, None)
Rq = R[M:].reshape(B, -1)
D = N * (N - 1) // 2
for name, opt in (('gemm pipeline', 0), ('wave kernel', 1)):
    ctx.set_option('predict.wave_only', opt)
    ts = []
    for rep in range(3):
        Ep, Fp = ctx.predict(Rq)
        ts.append(ctx.phase_ms('predict')[0])
    if opt == 0:
        F_ref = Fp
    else:
        print('   max |dF| / max|F| = %.2e' % (np.abs(Fp - F_ref).max() / np.abs(Fp).max()))
    ms = min(ts)
    print('N=%d D=%d M=%d P=%d (MP=%d) B=%d  %-14s %8.2f ms  %6.1f TFLOP/s algorithmic (10 D per pair)' % (
        N, D, M, len(perms), M * len(perms), B, name, ms, 10.0 * D * B * M * len(perms) / ms / 1e9), flush=True)
