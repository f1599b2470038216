"""Row-local panel solve (panel_trsm_kernel) inside the benchmark factorisation, per option set: summed time over its
launches and the rate of its triangular-solve flops.  python tools/trsm_probe.py "key=val,..." ...   ('-' = defaults;
trsm.debug bits are timing-only ablations: the factorisation result is then wrong)."""
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
    out = []
    for rep in range(2):
        ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
        ctx.chol_set_rhs(y)
        ctx.profile(True)
        try:
            ctx.chol_factor(1e-10)
        except Exception as e:      # ablations break positive definiteness
            pass
        ms, nl, work = ctx.kernel_stat('panel_trsm')
        gms, gnl, gwork = ctx.kernel_stat('gemm_nt_sub_diag')
        out.append('trsm %.1f ms / %d launches (%.1f TFLOP/s) gemm %.1f ms (%.1f TF) factor %.1f ms' % (
            ms, nl, work / ms / 1e9 if ms else 0, gms, gwork / gms / 1e9 if gms else 0, ctx.phase_ms('factor')[0]))
        ctx.profile(False)
    print('%-32s %s' % (spec, ' | '.join(out)), flush=True)
    ctx.close()
