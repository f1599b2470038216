"""Every rank of `bench.py --gpus 8` trains its own sigma-grid task on its own seeded data: check on one GPU
that all eight tasks factor (info = 0) and solve to the residual bound."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

N, M = 21, 1000
ctx = _lib.Context(0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
for rank in range(8):
    sig = 20.0 + 10.0 * rank
    R, E, F = synth_geometries(N, M, seed=rank)
    Rf = R.reshape(M, -1)
    y = F.ravel().copy(); y /= np.std(y)
    xd, gd = ctx.desc_from_R(Rf, N)
    ctx.train_upload(xd, gd, tp)
    ctx.assemble_K(sig, False, alloc_extra_rows=1)
    ctx.chol_set_rhs(y)
    try:
        info = ctx.chol_factor(1e-10)
        a = ctx.chol_solve(None)
        ctx.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
        Kv = ctx.kernel_matvec(1e-10, False, -a)
        print('rank %d sig %.0f: info %d, residual %.2e' % (rank, sig, info, np.linalg.norm(-Kv - y) / np.linalg.norm(y)), flush=True)
    except Exception as e:
        print('rank %d sig %.0f: FAILED %s' % (rank, sig, str(e)[:100]), flush=True)
