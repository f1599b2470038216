"""Iterative-solver probe: Nystroem-preconditioned CG through sgdml_amd.solvers.iterative on synthetic data."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd.train import GDMLTrain
from sgdml_amd.solvers.iterative import Iterative
from sgdml_amd.utils.desc import Desc

def run(N, M, k_ind, sig=20.0, tol=1e-4):
    R, E, F = synth_geometries(N, M, seed=0)
    np.random.seed(0)
    task = {'R_train': R, 'F_train': F, 'E_train': E, 'sig': sig, 'lam': 1e-10, 'use_E_cstr': False,
            'idxs_train': np.arange(M), 'perms': np.arange(N)[None]}
    tr = GDMLTrain()
    ctx = tr._context()
    desc = Desc(N); desc._ctx = ctx
    xd, gd = desc.from_R(R.reshape(M, -1))
    lin = np.arange(desc.dim)
    y = F.ravel().copy(); ystd = np.std(y); y /= ystd
    it = Iterative(tr, desc, None, None, False)
    n = M * 3 * N
    t0 = time.time()
    lev = it._lev_scores(xd, gd, lin, sig, 1e-10, False, k_ind)
    t1 = time.time()
    idxs = it.inducing_pts_from_lev_scores(lev, k_ind * 3 * N)
    it._init_precon_operator(task, xd, gd, lin, idxs)
    t2 = time.time()
    pre_ms = ctx.phase_ms('precon')[0]; asm_ms = ctx.phase_ms('assemble')[0]
    it._init_kernel_operator(task, xd, gd, lin, 1e-10, n)
    hist = []
    t3 = time.time()
    x, info, iters, resid = ctx.pcg(1e-10, False, y, rtol=tol, maxiter=3000, callback=lambda i, r, fetch_x: hist.append(r) or False, cb_every=25)
    t4 = time.time()
    pcg_ms = ctx.phase_ms('pcg')[0]
    print('N=%d M=%d n=%d k=%d: lev %.2fs, precon build %.2fs (assemble %.0f ms, factor %.0f ms), pcg %d iters info=%d in %.2fs (%.1f ms/iter) resid/|y| %.2e' % (
        N, M, n, k_ind, t1 - t0, t2 - t1, asm_ms, pre_ms, iters, info, pcg_ms / 1e3, pcg_ms / max(1, iters), resid / np.linalg.norm(y)), flush=True)
    print('   resid history (every 25):', ' '.join('%.1e' % (h / np.linalg.norm(y)) for h in hist[:20]), flush=True)
    tr.__del__()

if __name__ == '__main__':
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
