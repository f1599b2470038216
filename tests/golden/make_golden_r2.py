#!/usr/bin/env python
"""
Round-2 golden fixtures: outputs of the REFERENCE at BASELINE.json's configuration SHAPES (large n,
small on disk) and of the host-side procedures the product restates.  Build container only:

    python tests/golden/make_golden_r2.py [case ...]      (no argument = all cases)

Cases (files next to this script; every array float64 / int64):
  perm_c3        find_perms (utils/perm.py:395-404) on a molecule with a C3 x C3 x C2 group
  strat_sample   GDMLTrain.draw_strat_sample (train.py:1537-1646) under fixed seeds
  cfg0_n9_p6     configs[0] shape: N=9, P=6, M=200, analytic train; y, 64 sampled rows x 600 sampled
                 columns of K, alphas, c, std, predictions + validation errors (cli.py:1564-1640)
  cfg3_n42_p27   configs[3] shape: N=42 (D=861), P=27 = 3^3, M=6: sampled K rows, solve, predictions
  cfg4_n60_p1    configs[4] shape: N=60 (D=1770), P=1, M=4: sampled K rows, solve, predictions
  pcg_n9_m400    Iterative.solve (iterative.py:473-825) with the inducing columns it drew: iteration
                 count, residual norm after every iteration, final alphas, predictions
  nys_qr         Nystroem factor through the reference's QR branch (iterative.py:313-324, second Cholesky forced
                 to fail) next to its Cholesky branch
  lu_branch      a near-singular system on which scipy's Cholesky fails and the reference takes its LU
                 branch (analytic.py:101-114): alphas, predictions

/root/reference is imported from a scratch copy (it writes cache files into its package directory).
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import gdml_oracle as orc  # noqa: E402  (synthetic data generator only)

REF = '/root/reference/sgdml'
_ref = {}


def ref():
    if not _ref:
        scratch = tempfile.mkdtemp(prefix='sgdml_ref_')
        shutil.copytree(REF, os.path.join(scratch, 'sgdml'))
        sys.dont_write_bytecode = True
        sys.path.insert(0, scratch)
        import sgdml  # noqa: F401
        from sgdml.train import GDMLTrain
        from sgdml.predict import GDMLPredict
        from sgdml.utils.desc import Desc
        from sgdml.solvers.iterative import Iterative
        from sgdml.solvers import analytic as analytic_mod
        from sgdml.utils import perm as perm_mod

        _ref.update(GDMLTrain=GDMLTrain, GDMLPredict=GDMLPredict, Desc=Desc, Iterative=Iterative,
                    analytic_mod=analytic_mod, perm_mod=perm_mod, train=GDMLTrain(max_processes=1))
    return _ref


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-14s -> %d KiB' % (name, os.path.getsize(path) // 1024), flush=True)


def group_closure(gens, n):
    have = {tuple(range(n))}
    frontier = [tuple(range(n))]
    while frontier:
        nxt = []
        for a in frontier:
            for g in gens:
                c = tuple(a[i] for i in g)
                if c not in have:
                    have.add(c)
                    nxt.append(c)
        frontier = nxt
    return np.array(sorted(have), dtype=np.int64)


def make_task(ds, M, perms, sig, lam, use_E_cstr=False, n_valid=0):
    return {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': ds['z'], 'R_train': ds['R'][:M], 'F_train': ds['F'][:M], 'E_train': ds['E'][:M],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(M, M + n_valid), 'md5_valid': 'x',
        'sig': sig, 'lam': lam, 'use_E': True, 'use_E_cstr': use_E_cstr, 'use_sym': perms.shape[0] > 1,
        'perms': perms,
    }


def errors_like_cli(E_ref, F_ref, E_pred, F_pred):
    """MAE / RMSE exactly as cli.py:1564-1605 accumulates them (energies per geometry, forces per component)."""
    de, df = E_ref - E_pred, (F_ref - F_pred).ravel()
    return np.array([np.abs(de).mean(), np.sqrt((de**2).mean()), np.abs(df).mean(), np.sqrt((df**2).mean())])


# ----------------------------------------------------------------------------------------------


def case_perm_c3():
    base = [[0, 0, 0], [0, 0, 1.4]]
    for k in range(3):
        a = 2 * np.pi * k / 3
        base.append([np.cos(a), np.sin(a), -0.4])
    for k in range(3):
        a = 2 * np.pi * k / 3 + 0.3
        base.append([0.9 * np.cos(a), 0.9 * np.sin(a), 1.9])
    base = np.array(base)
    z = np.array([6, 6, 1, 1, 1, 1, 1, 1])
    rs = np.random.RandomState(3)
    R = base[None] + rs.normal(scale=0.08, size=(40,) + base.shape)
    perms = ref()['perm_mod'].find_perms(R, z, max_processes=1)
    # second molecule: heteroatoms break the C2 that swaps the two rotors
    z2 = np.array([6, 8, 1, 1, 1, 9, 9, 9])
    R2 = base[None] + rs.normal(scale=0.05, size=(25,) + base.shape)
    perms2 = ref()['perm_mod'].find_perms(R2, z2, max_processes=1)
    # periodic variant (minimum-image distances)
    lat = np.array([[4.0, 0.2, 0.0], [0.0, 4.2, 0.1], [0.1, 0.0, 4.4]])
    perms3 = ref()['perm_mod'].find_perms(R2, z2, lat_and_inv=(lat, np.linalg.inv(lat)), max_processes=1)
    print('  perms', perms.shape, perms2.shape, perms3.shape)
    save('perm_c3', R=R, z=z, perms=perms, R2=R2, z2=z2, perms2=perms2, lat=lat, perms3=perms3)


def case_strat_sample():
    gt = ref()['train']
    rs = np.random.RandomState(11)
    T = np.concatenate([rs.normal(-3, 1.0, 700), rs.normal(2, 0.5, 500), rs.uniform(-8, 8, 60)])
    out = {'T': T}
    for k, (seed, n, n_excl) in enumerate([(0, 50, 0), (1, 200, 0), (2, 100, 150), (3, 1, 0), (4, 7, 3), (5, 1260, 0)]):
        np.random.seed(seed)
        excl = np.sort(np.random.choice(T.size, n_excl, replace=False)) if n_excl else None
        np.random.seed(seed + 100)
        idx = gt.draw_strat_sample(T, n, excl_idxs=excl)
        out['seed%d' % k], out['n%d' % k] = np.int64(seed + 100), np.int64(n)
        out['excl%d' % k] = np.array([], dtype=np.int64) if excl is None else excl.astype(np.int64)
        out['idx%d' % k] = np.asarray(idx, dtype=np.int64)
    out['n_cases'] = np.int64(6)
    save('strat_sample', **out)


def _train_and_sample(name, N, M, perms, sig, n_test, seed, jitter, n_rows=64, n_cols=600, n_valid=0):
    r = ref()
    Desc, GDMLPredict, gt = r['Desc'], r['GDMLPredict'], r['train']
    ds = orc.synth_dataset(N, M + n_test + n_valid, seed=seed, jitter=jitter)
    lam = 1e-10
    task = make_task(ds, M, perms, sig, lam, n_valid=n_valid)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'][:M].reshape(M, -1))
    t0 = time.time()
    K = gt._assemble_kernel_mat(R_desc, R_d_desc, tril_perms_lin, sig, desc)
    t_asm = time.time() - t0
    n = K.shape[0]
    rs = np.random.RandomState(seed + 1)
    rows = np.sort(rs.choice(n, min(n_rows, n), replace=False))
    cols = np.sort(rs.choice(n, min(n_cols, n), replace=False))
    K_sample = np.array(K[np.ix_(rows, cols)])
    K_absmax = np.abs(K).max()
    K_fro = np.linalg.norm(K)
    del K

    used_lu = []
    amod = r['analytic_mod']
    orig_solve = amod.sp.linalg.solve

    def spy_solve(*a, **kw):
        used_lu.append(1)
        return orig_solve(*a, **kw)

    amod.sp.linalg.solve = spy_solve
    t0 = time.time()
    model = gt.train(task)
    t_train = time.time() - t0
    amod.sp.linalg.solve = orig_solve
    assert model['solver_name'] == 'analytic'
    y = ds['F'][:M].ravel().copy()
    y_std = np.std(y)
    y /= y_std
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:M + n_test]
    t0 = time.time()
    E_test, F_test = pred.predict(Rt.reshape(n_test, -1))
    t_pred = time.time() - t0
    out = dict(
        R_train=ds['R'][:M], E_train=ds['E'][:M], F_train=ds['F'][:M], z=ds['z'], perms=perms,
        sig=np.float64(sig), lam=np.float64(lam), rows=rows, cols=cols, K_sample=K_sample,
        K_absmax=np.float64(K_absmax), K_fro=np.float64(K_fro), y=y, y_std=np.float64(y_std),
        alphas=model['alphas_F'], model_c=np.float64(model['c']), model_std=np.float64(model['std']),
        R_test=Rt, E_test=E_test, F_test=F_test, used_lu=np.int64(len(used_lu)),
        ref_seconds=np.array([t_asm, t_train, t_pred]),
    )
    if n_valid:
        Rv = ds['R'][M + n_test:]
        Ev, Fv = pred.predict(Rv.reshape(n_valid, -1))
        out.update(R_valid=Rv, E_valid_ref=ds['E'][M + n_test:], F_valid_ref=ds['F'][M + n_test:],
                   valid_errors=errors_like_cli(ds['E'][M + n_test:], ds['F'][M + n_test:].reshape(n_valid, -1), Ev, Fv))
    print('  %s: n=%d assemble %.1fs train %.1fs predict %.2fs lu=%d |K|max %.3e' %
          (name, n, t_asm, t_train, t_pred, len(used_lu), K_absmax), flush=True)
    save(name, **out)


def case_cfg0_n9_p6():
    perms = group_closure([(1, 2, 0, 3, 4, 5, 6, 7, 8), (0, 1, 2, 4, 3, 5, 6, 7, 8)], 9)
    assert perms.shape[0] == 6
    _train_and_sample('cfg0_n9_p6', 9, 200, perms, 20, 50, seed=21, jitter=0.3, n_valid=100)


def case_cfg3_n42_p27():
    gens = []
    for a in (3, 17, 30):
        g = list(range(42))
        g[a], g[a + 1], g[a + 2] = a + 1, a + 2, a
        gens.append(tuple(g))
    perms = group_closure(gens, 42)
    assert perms.shape[0] == 27
    _train_and_sample('cfg3_n42_p27', 42, 6, perms, 40, 5, seed=22, jitter=0.3, n_rows=96, n_cols=756)


def case_cfg4_n60_p1():
    _train_and_sample('cfg4_n60_p1', 60, 4, np.arange(60)[None, :], 50, 5, seed=23, jitter=0.3, n_rows=96, n_cols=720)


def case_pcg_n9_m400():
    import inspect

    r = ref()
    Desc, GDMLPredict, Iterative, gt = r['Desc'], r['GDMLPredict'], r['Iterative'], r['train']
    import sgdml.solvers.iterative as it_mod

    N, M, sig, lam = 9, 400, 20, 1e-10
    ds = orc.synth_dataset(N, M + 20, seed=31, jitter=0.3)
    perms = np.arange(N)[None, :]
    task = make_task(ds, M, perms, sig, lam)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'][:M].reshape(M, -1))
    y = ds['F'][:M].ravel().copy()
    y_std = np.std(y)
    y /= y_std

    hist = []
    real_cg = it_mod.sp.sparse.linalg.cg

    def spy_cg(A, b, x0=None, M=None, rtol=1e-5, atol=0.0, maxiter=None, callback=None):
        def cb(xk):
            fl = inspect.currentframe().f_back.f_locals  # scipy's cg frame
            hist.append(float(np.linalg.norm(fl['r'])))
            callback(xk)

        return real_cg(A, b, x0=x0, M=M, rtol=rtol, atol=atol, maxiter=maxiter, callback=cb)

    it_mod.sp.sparse.linalg.cg = spy_cg
    np.random.seed(5)
    it = Iterative(gt, desc, 0.2, 1, False)  # max_memory = 0.2 GB -> k = 21 inducing points
    t0 = time.time()
    alphas, tol, n_iters, resid, train_rmse, inducing, is_conv = it.solve(
        task, R_desc, R_d_desc, tril_perms_lin, y, y_std, tol=1e-4)
    dt = time.time() - t0
    it_mod.sp.sparse.linalg.cg = real_cg
    print('  pcg: k=%d iters=%d resid=%.3e conv=%s  %.1fs' % (len(inducing) // (3 * N), n_iters, resid, is_conv, dt),
          flush=True)
    model = gt.create_model(task, 'cg', R_desc, R_d_desc, tril_perms_lin, y_std, alphas)
    model['c'] = gt._recov_int_const(model, task, R_desc=R_desc, R_d_desc=R_d_desc)
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    save('pcg_n9_m400', R_train=ds['R'][:M], E_train=ds['E'][:M], F_train=ds['F'][:M], z=ds['z'], perms=perms,
         sig=np.float64(sig), lam=np.float64(lam), y=y, y_std=np.float64(y_std), inducing_pts_idxs=np.asarray(inducing),
         resid_hist=np.array(hist), n_iters=np.int64(n_iters), resid=np.float64(resid), is_conv=np.bool_(is_conv),
         alphas=alphas, model_c=np.float64(model['c']), R_test=Rt, E_test=E_test, F_test=F_test,
         ref_seconds=np.float64(dt))


def case_lu_branch():
    """K has an exact null space (3N > D for N = 6), so with lam = 1e-18 below the rounding level of the
    factorisation scipy's cho_factor raises and the reference solves by LU."""
    r = ref()
    Desc, GDMLPredict, gt = r['Desc'], r['GDMLPredict'], r['train']
    amod = r['analytic_mod']
    N, M, sig = 6, 24, 10
    found = None
    for lam in (1e-18,):
        ds = orc.synth_dataset(N, M + 6, seed=41, jitter=0.3, n_conformers=2)
        perms = np.arange(N)[None, :]
        task = make_task(ds, M, perms, sig, lam)
        used_lu = []
        orig_solve = amod.sp.linalg.solve

        def spy_solve(*a, **kw):
            used_lu.append(1)
            return orig_solve(*a, **kw)

        amod.sp.linalg.solve = spy_solve
        model = gt.train(task)
        amod.sp.linalg.solve = orig_solve
        print('  lam %.0e lu=%d |alpha|max %.3e' % (lam, len(used_lu), np.abs(model['alphas_F']).max()), flush=True)
        if used_lu:
            found = (lam, ds, task, model)
            break
    assert found is not None, 'no LU case found'
    lam, ds, task, model = found
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    y = ds['F'][:M].ravel().copy()
    y_std = np.std(y)
    y /= y_std
    save('lu_branch', R_train=ds['R'][:M], E_train=ds['E'][:M], F_train=ds['F'][:M], z=ds['z'], perms=task['perms'],
         sig=np.float64(sig), lam=np.float64(lam), y=y, y_std=np.float64(y_std), alphas=model['alphas_F'],
         model_c=np.float64(model['c']), model_std=np.float64(model['std']), R_test=Rt, E_test=E_test, F_test=F_test)


def case_nys_qr():
    """Iterative._nystroem_cholesky_factor with the second Cholesky made to fail, so that the reference takes its
    QR branch (iterative.py:313-324).  The branch is hard to reach with real data (scipy's Cholesky only gives up
    on Gram matrices that are indefinite by more than the 1e-15 of jitter it is allowed), hence the forced failure."""
    r = ref()
    Desc, Iterative, gt = r['Desc'], r['Iterative'], r['train']
    N, M, sig, lam = 6, 14, 10, 1e-10
    ds = orc.synth_dataset(N, M, seed=51, jitter=0.3)
    perms = np.arange(N)[None, :]
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(ds['R'].reshape(M, -1))
    n = M * 3 * N
    col_idxs = np.sort(np.random.RandomState(2).choice(n, 40, replace=False))
    it = Iterative(gt, desc, None, 1, False)
    fac_chol = it._nystroem_cholesky_factor(R_desc, R_d_desc, tril_perms_lin, sig, lam, use_E_cstr=False,
                                            col_idxs=col_idxs).copy()
    calls = []
    orig = it._cho_factor_stable

    def failing_second(Mat, pre_reg=False, eps_mag_max=1):
        calls.append(1)
        return None if len(calls) == 2 else orig(Mat, pre_reg=pre_reg, eps_mag_max=eps_mag_max)

    it._cho_factor_stable = failing_second
    fac_qr = it._nystroem_cholesky_factor(R_desc, R_d_desc, tril_perms_lin, sig, lam, use_E_cstr=False,
                                          col_idxs=col_idxs, callback=lambda *a, **k: None).copy()
    assert len(calls) == 2
    d = np.abs(fac_qr.T @ fac_qr - fac_chol.T @ fac_chol).max() / np.abs(fac_chol.T @ fac_chol).max()
    print('  nys_qr: |P_qr - P_chol| rel %.2e' % d, flush=True)
    save('nys_qr', R_train=ds['R'], perms=perms, sig=np.float64(sig), lam=np.float64(lam), col_idxs=col_idxs,
         L_inv_K_mn_qr=fac_qr, L_inv_K_mn_chol=fac_chol)


CASES = dict(nys_qr=case_nys_qr, perm_c3=case_perm_c3, strat_sample=case_strat_sample, cfg0_n9_p6=case_cfg0_n9_p6,
             cfg3_n42_p27=case_cfg3_n42_p27, cfg4_n60_p1=case_cfg4_n60_p1, pcg_n9_m400=case_pcg_n9_m400,
             lu_branch=case_lu_branch)

if __name__ == '__main__':
    for c in (sys.argv[1:] or list(CASES)):
        print(c, flush=True)
        CASES[c]()
