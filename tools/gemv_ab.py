"""A/B of the preconditioner GEMVs inside the PCG loop: one configs[2]-shaped Nystroem factor (N = 21, N_train = 5000, k = 162:
X is 315 000 x 10 206 = 25.7 GB), then blocks of PCG iterations per value of option pcg.gemv_plain (1 = plain instead of non-temporal loads of X).
  python tools/gemv_ab.py 0 1 0 1      -> ms per PCG iteration (mat-vec + X^T v + X t + vector updates)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from sgdml_amd import _lib

ITERS = int(os.environ.get('AB_ITERS', '40'))
ctx = _lib.Context(0)
wl = bench.make_cg_workload(ctx, 21, 5000, 162, 20.0, 1e-10)
ctx.assemble_K(wl['sig'], False, idx=wl['idx'], alloc_extra_rows=wl['m'])
ctx.nystroem_factor(wl['lam'], wl['idx'])
for v in [int(a) for a in (sys.argv[1:] or ['0'])]:
    ctx.set_option('pcg.gemv_plain', float(v))
    ms = []
    for rep in range(2):
        x, info, iters, resid = ctx.pcg(wl['lam'], False, wl['y'], rtol=0.0, maxiter=ITERS)
        ms.append(ctx.phase_ms('pcg')[0] / ITERS)
    gk = ctx.kernel_stat('gemv_t') if hasattr(ctx, 'kernel_stat') else None
    print('pcg.gemv_plain=%-3d  %.3f %.3f ms per PCG iteration   resid %.3e' % (v, ms[0], ms[1], resid), flush=True)
ctx.close()
