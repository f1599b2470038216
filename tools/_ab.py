import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib
M, N = 1000, 21
R, E, F = synth_geometries(N, M, seed=0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
y = F.ravel() / np.std(F)
for spec in sys.argv[1:]:
    ctx = _lib.Context(0)
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tp)
    if spec != '-':
        for kv in spec.split(','):
            k, v = kv.split('='); ctx.set_option(k, float(v))
    ts = []
    for rep in range(2):
        ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
        ctx.chol_set_rhs(y)
        try:
            info = ctx.chol_factor(1e-10)
        except Exception as e:
            info = -1
        ts.append(ctx.phase_ms('factor')[0])
    print('%-30s factor %s ms info %s' % (spec, ' '.join('%.1f' % t for t in ts), info), flush=True)
    ctx.close()
