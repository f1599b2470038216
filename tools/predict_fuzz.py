"""Randomised comparison of the MFMA predict kernel with the wave kernel (same inputs, same process)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    N = int(rs.randint(3, 24)); M = int(rs.randint(5, 500)); B = int(rs.randint(256, 1600))
    P = int(rs.choice([1, 1, 2])); with_aE = bool(rs.randint(0, 2)); sig = float(rs.choice([5.0, 12.0, 40.0]))
    R, E, F = synth_geometries(N, M + B, seed=trial)
    Rf = R.reshape(M + B, -1)
    D = N * (N - 1) // 2
    perms = np.arange(N)[None]
    if P == 2:
        p2 = np.arange(N); p2[[0, 1]] = p2[[1, 0]]; perms = np.vstack([perms, p2])
    # descriptor permutation of an atom permutation (same rule as the library's inverse)
    iu = {}
    k = 0
    for i in range(N):
        for j in range(i):
            iu[(i, j)] = k; k += 1
    tp = np.array([[iu[(max(p[i], p[j]), min(p[i], p[j]))] for i in range(N) for j in range(i)] for p in perms], dtype=np.int64)
    ctx = _lib.Context(0)
    xd, gd = ctx.desc_from_R(Rf[:M], N)
    ja = rs.normal(size=xd.shape) * 10.0 ** rs.uniform(-2, 4)
    aE = rs.normal(size=M) if with_aE else None
    ctx.predict_upload_model(xd, ja, tp, sig, aE)
    E1, F1 = ctx.predict(Rf[M:])
    ctx.set_option('predict.wave_only', 1)
    E0, F0 = ctx.predict(Rf[M:])
    dF = np.abs(F1 - F0).max() / np.abs(F0).max(); dE = np.abs(E1 - E0).max() / np.abs(E0).max()
    worst = max(worst, dF, dE)
    flag = '' if max(dF, dE) < 1e-11 else '   <-- CHECK'
    print('N=%2d M=%3d B=%4d P=%d aE=%d sig=%4.0f: dF %.1e dE %.1e%s' % (N, M, B, P, with_aE, sig, dF, dE, flag), flush=True)
    ctx.close()
print('worst relative difference: %.2e' % worst)
