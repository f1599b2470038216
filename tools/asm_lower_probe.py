"""Assembly of the system matrix in the fused lower form (gdml_assemble_A) at the benchmark size, per option set.
    python tools/asm_lower_probe.py "key=val,..." ..."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import synth_geometries
from sgdml_amd import _lib

M, N = int(os.environ.get('AB_M', '1000')), 21
R, E, F = synth_geometries(N, M, seed=0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
for spec in (sys.argv[1:] or ['-']):
    ctx = _lib.Context(0)
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tp)
    full = False
    if spec != '-':
        for kv in spec.split(','):
            k, v = kv.split('=')
            if k == 'full':   # pseudo-option: the full un-negated K (gdml_assemble_K) instead of the lower form
                full = bool(int(v))
            else:
                ctx.set_option(k, float(v))
    ts = []
    for rep in range(14):
        ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=None if full else 1e-10)
        ts.append(ctx.phase_ms('assemble')[0])
    n = M * 3 * N
    by = 8.0 * (M * M if full else 0.5 * M * (M + 1)) * (3 * N) ** 2
    med = float(np.median(ts[6:]))
    print('%-28s assemble %s ms -> median of last 8 %.2f ms = %.0f GB/s (%.3f of 8 TB/s)' % (spec, ' '.join('%.2f' % t for t in ts), med, by / med / 1e6, by / med / 1e6 / 8000), flush=True)
    ctx.close()
