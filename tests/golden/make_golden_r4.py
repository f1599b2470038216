#!/usr/bin/env python
"""
Round-4 golden fixtures: the REFERENCE's iterative solver with a real permutation group, and its index-list column
assembly at a mid-size molecule.  Build container only:

    python tests/golden/make_golden_r4.py [case ...]      (no argument = all cases)

Cases (files next to this script):
  pcg_n12_p6_m200  Iterative.solve (iterative.py:473-825) on N = 12 atoms with a 6-element group (C3 x C2), M = 200
                   (n = 7200), k = 40 inducing points chosen by the reference's own leverage sampling: the inducing
                   columns, 96 sampled rows of the K_nm it assembled for them (index-list mode, train.py:1376-1407),
                   residual after every iteration, coefficients, predictions.
  cols_n24_p6      _assemble_kernel_mat with an index list (incl. partial blocks) and with a point slice on N = 24,
                   P = 6, M = 12 (n = 864): sampled rows of both results -- column modes above the 9-atom fixtures.
  n150_p2_m3       a 150-atom molecule (D = 11 175) with a two-element group, M = 3 (n = 1350): sampled K entries, analytic
                   train, predictions -- beyond the 128 atoms the kernels were limited to before round 4.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import make_golden_r2 as g2  # noqa: E402  (reference loader, task builder)
import make_golden_r3 as g3  # noqa: E402  (_spy_cg)
from oracle import gdml_oracle as orc  # noqa: E402  (synthetic data generator only)


def _group_c3_c2(N, a, b):
    rot = list(range(N))
    rot[a], rot[a + 1], rot[a + 2] = a + 1, a + 2, a
    swp = list(range(N))
    swp[b], swp[b + 1] = b + 1, b
    perms = g2.group_closure([tuple(rot), tuple(swp)], N)
    assert perms.shape[0] == 6
    return perms


def case_pcg_n12_p6_m200():
    r = g2.ref()
    Desc, GDMLPredict, Iterative, gt = r['Desc'], r['GDMLPredict'], r['Iterative'], r['train']
    import sgdml.solvers.iterative as it_mod

    N, M, sig, lam, k = 12, 200, 20, 1e-10, 40
    perms = _group_c3_c2(N, 0, 5)
    ds = orc.synth_dataset(N, M + 12, seed=41, jitter=0.3)
    task = g2.make_task(ds, M, perms, sig, lam)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'][:M].reshape(M, -1))
    y = ds['F'][:M].ravel().copy()
    y_std = np.std(y)
    y /= y_std
    hist, starts = [], []
    real_cg, spy = g3._spy_cg(it_mod, hist, starts)
    it_mod.sp.sparse.linalg.cg = spy
    orig_k = Iterative.max_n_inducing_pts
    Iterative.max_n_inducing_pts = staticmethod(lambda n_train, n_atoms, mb: k)
    np.random.seed(17)
    t0 = time.time()
    try:
        it = Iterative(gt, desc, 1, 1, False)
        alphas, tol, n_iters, resid, train_rmse, inducing, is_conv = it.solve(
            task, R_desc, R_d_desc, tril_perms_lin, y, y_std, tol=1e-4)
    finally:
        it_mod.sp.sparse.linalg.cg = real_cg
        Iterative.max_n_inducing_pts = orig_k
    dt = time.time() - t0
    print('  pcg_n12_p6: k=%d iters=%d resid=%.3e conv=%s cg calls at %s  %.1fs' % (
        len(inducing) // (3 * N), n_iters, resid, is_conv, starts, dt), flush=True)
    assert is_conv and len(starts) == 1
    inducing = np.asarray(inducing)
    # the K_nm the reference's preconditioner was built from: its own index-list assembly, sampled rows
    K_nm = gt._assemble_kernel_mat(R_desc, R_d_desc, tril_perms_lin, sig, desc, col_idxs=inducing)
    rs = np.random.RandomState(42)
    rows = np.sort(rs.choice(K_nm.shape[0], 96, replace=False))
    model = gt.create_model(task, 'cg', R_desc, R_d_desc, tril_perms_lin, y_std, alphas)
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    g2.save('pcg_n12_p6_m200', R_train=ds['R'][:M], E_train=ds['E'][:M], F_train=ds['F'][:M], z=ds['z'], perms=perms,
            sig=np.float64(sig), lam=np.float64(lam), y=y, y_std=np.float64(y_std), inducing_pts_idxs=inducing,
            k=np.int64(k), K_nm_rows=rows, K_nm_sample=np.array(K_nm[rows]), K_nm_absmax=np.float64(np.abs(K_nm).max()),
            K_nm_fro=np.float64(np.linalg.norm(K_nm)), resid_hist=np.array(hist), n_iters=np.int64(n_iters),
            resid=np.float64(resid), is_conv=np.bool_(is_conv), alphas=alphas, model_c=np.float64(model['c']),
            R_test=Rt, E_test=E_test, F_test=F_test, ref_seconds=np.float64(dt), seed=np.int64(17))


def case_cols_n24_p6():
    r = g2.ref()
    Desc, gt = r['Desc'], r['train']
    N, M, sig = 24, 12, 15
    perms = _group_c3_c2(N, 3, 17)
    ds = orc.synth_dataset(N, M, seed=43, jitter=0.3)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'].reshape(M, -1))
    n = M * 3 * N
    rs = np.random.RandomState(44)
    idx = np.sort(rs.choice(n, 150, replace=False))  # arbitrary columns: partial blocks of most points
    K_idx = gt._assemble_kernel_mat(R_desc, R_d_desc, tril_perms_lin, sig, desc, col_idxs=idx)
    # a slice of whole points must start at point 0 in the reference: its worker addresses block column j at 3N j of the
    # (narrower) result whatever the slice's start (train.py:156-159 with :1357-1374)
    p0, p1 = 0, 8
    K_pts = gt._assemble_kernel_mat(R_desc, R_d_desc, tril_perms_lin, sig, desc, col_idxs=np.s_[:p1 * 3 * N])
    rows = np.sort(rs.choice(n, 200, replace=False))
    print('  cols_n24_p6: K_idx %s K_pts %s max %.3e' % (K_idx.shape, K_pts.shape, np.abs(K_idx).max()), flush=True)
    g2.save('cols_n24_p6', R_train=ds['R'], z=ds['z'], perms=perms, sig=np.float64(sig), col_idxs=idx,
            points=np.array([p0, p1]), rows=rows, K_idx_sample=np.array(K_idx[rows]), K_pts_sample=np.array(K_pts[rows]),
            K_idx_fro=np.float64(np.linalg.norm(K_idx)), K_pts_fro=np.float64(np.linalg.norm(K_pts)),
            K_absmax=np.float64(np.abs(K_pts).max()))


def case_n150_p2_m3():
    N = 150
    swp = list(range(N))
    swp[70], swp[71] = 71, 70
    perms = g2.group_closure([tuple(swp)], N)
    assert perms.shape[0] == 2
    g2._train_and_sample('n150_p2_m3', N, 3, perms, 80, 4, seed=65, jitter=0.3, n_rows=160, n_cols=400)


CASES = ['pcg_n12_p6_m200', 'cols_n24_p6', 'n150_p2_m3']

if __name__ == '__main__':
    for c in sys.argv[1:] or CASES:
        print(c, flush=True)
        globals()['case_' + c]()
