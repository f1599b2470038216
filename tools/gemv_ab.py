"""A/B of the preconditioner application inside the PCG loop at the configs[2] shape (N = 21, N_train = 5000, k = 162: the factor
is 315 000 x 10 206 -- 25.7 GB in fp64, 12.9 GB in fp32): per setting one Nystroem factor, then blocks of PCG iterations.
Settings are comma-separated option lists (pcg. prefix implied):
  python tools/gemv_ab.py precon_form=0 precon_form=3,f32_rw=1 precon_form=3,f32_rw=4,f32_rows_per=2048
-> ms per PCG iteration (mat-vec + both GEMVs + vector updates) and the GEMV kernels' own time / rate (gdml_kernel_stat)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from sgdml_amd import _lib

ITERS = int(os.environ.get('AB_ITERS', '40'))
SHAPE = [int(v) for v in os.environ.get('AB_SHAPE', '21,5000,162').split(',')]
ctx = _lib.Context(0)
wl = bench.make_cg_workload(ctx, SHAPE[0], SHAPE[1], SHAPE[2], 20.0, 1e-10)
for setting in (sys.argv[1:] or ['precon_form=0']):
    opts = dict(kv.split('=') for kv in setting.split(','))
    for k, v in opts.items():
        ctx.set_option('pcg.' + k, float(v))
    ctx.assemble_K(wl['sig'], False, idx=wl['idx'], alloc_extra_rows=wl['m'])
    _, _, info = ctx.nystroem_factor(wl['lam'], wl['idx'], want_lev=False)
    build_ms = ctx.phase_ms('precon')[0]
    ms = []
    for rep in range(2):
        ctx.profile(True)
        x, info_p, iters, resid = ctx.pcg(wl['lam'], False, wl['y'], rtol=0.0, maxiter=ITERS)
        ms.append(ctx.phase_ms('pcg')[0] / ITERS)
        g_ms, g_n, g_by = ctx.kernel_stat('precon_gemv')
        ctx.profile(False)
    print('%-48s info %d  build %7.1f ms  %.3f %.3f ms per PCG iteration  GEMVs %.3f ms = %.2f TB/s  resid %.3e' %
          (setting, info, build_ms, ms[0], ms[1], g_ms / max(1, g_n), g_by / max(1e-9, g_ms * 1e-3) / 1e12, resid),
          'min pivot^2 of the fp32 Gram: %s' % ctx.get_option('pcg.f32_last_min_pivot'), flush=True)
ctx.close()
