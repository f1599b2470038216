#!/usr/bin/env python
"""
Round-6 golden fixture: the REFERENCE's analytic training WITH ENERGY CONSTRAINTS on a system that spans several row blocks
of the distributed Cholesky (the only energy-constraint fixture so far, n5_p2_ecstr, has n = 128 = one block).  Build container
only:

    python tests/golden/make_golden_r6.py

  ecstr_n9_p6_m40   N = 9 atoms, 6-element group, M = 40, use_E_cstr (n = 1080 + 40 = 1120), lam = 1e-8: the M energy ROWS of the
                    K the reference assembled (train.py:235-300; 40 x 1120), the label vector (train.py:937-947), the
                    coefficients alphas_F / alphas_E of GDMLTrain.train (analytic.py:65-99), integration constant, predictions.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import make_golden_r2 as g2  # noqa: E402  (reference loader, task builder)
from oracle import gdml_oracle as orc  # noqa: E402  (synthetic data generator only)


def case_ecstr_n9_p6_m40():
    r = g2.ref()
    Desc, GDMLPredict, gt = r['Desc'], r['GDMLPredict'], r['train']
    N, M, sig, lam, n_test = 9, 40, 20, 1e-8, 6
    rot = list(range(N)); rot[0], rot[1], rot[2] = 1, 2, 0
    swp = list(range(N)); swp[3], swp[4] = 4, 3
    perms = g2.group_closure([tuple(rot), tuple(swp)], N)
    assert perms.shape[0] == 6
    ds = orc.synth_dataset(N, M + n_test, seed=66, jitter=0.3)
    task = g2.make_task(ds, M, perms, sig, lam, use_E_cstr=True)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'][:M].reshape(M, -1))
    K = gt._assemble_kernel_mat(R_desc, R_d_desc, tril_perms_lin, sig, desc, use_E_cstr=True)
    n_ff = M * 3 * N
    assert K.shape == (n_ff + M, n_ff + M)
    K_E_rows = np.array(K[n_ff:, :])
    sym = np.abs(K - K.T).max()
    eig_min = np.linalg.eigvalsh(-K + lam * np.eye(K.shape[0]))[0]
    del K
    t0 = time.time()
    model = gt.train(task)
    assert model['solver_name'] == 'analytic'
    E_train = ds['E'][:M].ravel()
    y = np.hstack((ds['F'][:M].ravel(), -E_train + E_train.mean()))  # train.py:937-947
    y_std = np.std(y)
    y = y / y_std
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:M + n_test]
    E_test, F_test = pred.predict(Rt.reshape(n_test, -1))
    print('  ecstr_n9_p6_m40: n=%d |K-K^T|max %.1e min eig(A) %.2e train %.1fs' % (n_ff + M, sym, eig_min, time.time() - t0),
          flush=True)
    g2.save('ecstr_n9_p6_m40', R_train=ds['R'][:M], E_train=ds['E'][:M], F_train=ds['F'][:M], z=ds['z'], perms=perms,
            sig=np.float64(sig), lam=np.float64(lam), K_E_rows=K_E_rows, y=y, y_std=np.float64(y_std),
            alphas_F=model['alphas_F'], alphas_E=model['alphas_E'], model_c=np.float64(model['c']),
            model_std=np.float64(model['std']), R_test=Rt, E_test=E_test, F_test=F_test)


def case_pcg_ecstr_n9_p6_m150():
    """The reference's Iterative.solve WITH energy constraints (iterative.py:473-825 on the (3N + 1) M system of train.py:235-300):
    N = 9, 6-element group, M = 150 (n = 4050 + 150), k = 3 inducing points, lam = 1e-8: the inducing columns it drew (energy
    columns can be among them), sampled rows of the K_nm it assembled for them INCLUDING energy rows, the residual after every
    iteration, the coefficients (forces and energies) and predictions from the model."""
    import make_golden_r3 as g3

    r = g2.ref()
    Desc, GDMLPredict, Iterative, gt = r['Desc'], r['GDMLPredict'], r['Iterative'], r['train']
    import sgdml.solvers.iterative as it_mod

    N, M, sig, lam, k, n_test = 9, 150, 20, 1e-8, 3, 6
    rot = list(range(N)); rot[0], rot[1], rot[2] = 1, 2, 0
    swp = list(range(N)); swp[3], swp[4] = 4, 3
    perms = g2.group_closure([tuple(rot), tuple(swp)], N)
    ds = orc.synth_dataset(N, M + n_test, seed=67, jitter=0.3)
    task = g2.make_task(ds, M, perms, sig, lam, use_E_cstr=True)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'][:M].reshape(M, -1))
    E_train = ds['E'][:M].ravel()
    y = np.hstack((ds['F'][:M].ravel(), -E_train + E_train.mean()))  # train.py:937-947
    y_std = np.std(y)
    y = y / y_std
    hist, starts = [], []
    real_cg, spy = g3._spy_cg(it_mod, hist, starts)
    it_mod.sp.sparse.linalg.cg = spy
    orig_k = Iterative.max_n_inducing_pts
    Iterative.max_n_inducing_pts = staticmethod(lambda n_train, n_atoms, mb: k)
    np.random.seed(23)
    t0 = time.time()
    try:
        it = Iterative(gt, desc, 1, 1, False)
        alphas, tol, n_iters, resid, train_rmse, inducing, is_conv = it.solve(
            task, R_desc, R_d_desc, tril_perms_lin, y, y_std, tol=1e-4)
    finally:
        it_mod.sp.sparse.linalg.cg = real_cg
        Iterative.max_n_inducing_pts = orig_k
    print('  pcg_ecstr: k=%d cols=%d (energy columns: %d) iters=%d resid=%.3e conv=%s cg calls at %s  %.1fs' % (
        k, len(inducing), int((np.asarray(inducing) >= M * 3 * N).sum()), n_iters, resid, is_conv, starts, time.time() - t0),
        flush=True)
    assert is_conv and len(starts) == 1
    inducing = np.asarray(inducing)
    K_nm = gt._assemble_kernel_mat(R_desc, R_d_desc, tril_perms_lin, sig, desc, use_E_cstr=True, col_idxs=inducing)
    rs = np.random.RandomState(42)
    n_ff = M * 3 * N
    rows = np.sort(np.concatenate((rs.choice(n_ff, 80, replace=False), n_ff + rs.choice(M, 16, replace=False))))
    model = gt.create_model(task, 'cg', R_desc, R_d_desc, tril_perms_lin, y_std, alphas[:n_ff], alphas_E=alphas[n_ff:])
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    g2.save('pcg_ecstr_n9_p6_m150', R_train=ds['R'][:M], E_train=ds['E'][:M], F_train=ds['F'][:M], z=ds['z'], perms=perms,
            sig=np.float64(sig), lam=np.float64(lam), y=y, y_std=np.float64(y_std), inducing_pts_idxs=inducing, k=np.int64(k),
            K_nm_rows=rows, K_nm_sample=np.array(K_nm[rows]), K_nm_absmax=np.float64(np.abs(K_nm).max()),
            K_nm_fro=np.float64(np.linalg.norm(K_nm)), resid_hist=np.array(hist), n_iters=np.int64(n_iters),
            resid=np.float64(resid), alphas=alphas, R_test=Rt, E_test=E_test, F_test=F_test, seed=np.int64(23))


if __name__ == '__main__':
    for c in sys.argv[1:] or ['ecstr_n9_p6_m40', 'pcg_ecstr_n9_p6_m150']:
        globals()['case_' + c]()
