"""A/B of factorisation options at the benchmark size: python tools/chol_ab.py "key=val,key=val" "key=val" ...
(each argument = one option set; '-' = defaults).  Prints factor time (2 repetitions) and the solve residual."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import synth_geometries
from sgdml_amd import _lib

M, N = int(os.environ.get('AB_M', '1000')), 21
R, E, F = synth_geometries(N, M, seed=0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
y = F.ravel() / np.std(F)
for spec in (sys.argv[1:] or ['-']):
    ctx = _lib.Context(0)
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tp)
    if spec != '-':
        for kv in spec.split(','):
            k, v = kv.split('=')
            ctx.set_option(k, float(v))
    ts, ta = [], []
    for rep in range(3):
        ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
        ta.append(ctx.phase_ms('assemble')[0])
        ctx.chol_set_rhs(y)
        info = ctx.chol_factor(1e-10)
        ts.append(ctx.phase_ms('factor')[0])
    a = ctx.chol_solve(None)
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, 20.0, None)
    Kv = ctx.kernel_matvec(1e-10, False, -a)
    res = np.linalg.norm(-Kv - y) / np.linalg.norm(y)
    print('%-40s factor %s ms  info %d  resid %.2e  assemble %s ms' % (spec, ' '.join('%.1f' % t for t in ts), info, res,
                                                                            ' '.join('%.2f' % t for t in ta)), flush=True)
    ctx.close()
