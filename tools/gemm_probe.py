"""GEMM ablation probe: factor the benchmark matrix under GDML_OPTIONS="gemm.debug=<mask>" and report the
aggregated gemm_nt_sub rate (numerically meaningless masks are fine: NOT_PD is caught)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = 21
R, E, F = synth_geometries(N, M, seed=0)
ctx = _lib.Context(0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
ctx.train_upload(xd, gd, tp)
ctx.profile(True)
for rep in range(2):
    ctx.assemble_K(20.0, False)
    ctx.profile(True)
    try:
        ctx.chol_factor(1e-10)
    except np.linalg.LinAlgError:
        pass
    ms, n, w = ctx.kernel_stat('gemm_nt_sub')
    print('dbg=%s rep %d: gemm %.1f TF (%.1f ms in %d launches), factor %.1f ms' % (
        os.environ.get('GDML_OPTIONS', '-'), rep, w / ms / 1e9, ms, n, ctx.phase_ms('factor')[0]), flush=True)
