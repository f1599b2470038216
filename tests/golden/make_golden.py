#!/usr/bin/env python
"""
Generate golden fixtures by running the REFERENCE implementation (pure Python) of the
hot path on seeded synthetic inputs.  Run in the build container only:

    python tests/golden/make_golden.py

The reference is imported from a scratch copy of /root/reference/sgdml (the reference
writes a cache file into its own package dir, predict.py:1044-1074; /root/reference is
read-only for us).  /root/reference does not exist on the GPU box, so the outputs are
committed as small .npz files next to this script and only those are read by tests.

Fixture contents (all float64 / int64):
  inputs : R (M,N,3), perms (P,N), sig, lam, [lattice]
  ref    : R_desc, R_d_desc, tril_perms_lin, K (un-negated, train.py:1535),
           K_cols (index-list columns + alloc_extra_rows), col_idxs,
           y, alphas (reference analytic solve), model_* (reference create_model +
           _recov_int_const), R_test, E_test, F_test (reference NumPy GDMLPredict.predict),
           E_train_pred, F_train_pred (training-set mode), Kv (reference _K_vec),
           L_inv_K_mn (reference _nystroem_cholesky_factor on inducing col_idxs)
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import gdml_oracle as orc  # noqa: E402  (synthetic data generator only)

REF = '/root/reference/sgdml'


def _import_reference():
    scratch = tempfile.mkdtemp(prefix='sgdml_ref_')
    shutil.copytree(REF, os.path.join(scratch, 'sgdml'))
    sys.dont_write_bytecode = True
    sys.path.insert(0, scratch)
    import sgdml  # noqa: F401
    from sgdml.train import GDMLTrain
    from sgdml.predict import GDMLPredict
    from sgdml.utils.desc import Desc
    from sgdml.solvers.iterative import Iterative

    return GDMLTrain, GDMLPredict, Desc, Iterative


CASES = {
    # name: (n_atoms, M, perms, sig, use_E_cstr, lattice?)
    'n6_p1': dict(N=6, M=12, perms=[[0, 1, 2, 3, 4, 5]], sig=10, E_cstr=False),
    'n5_p4': dict(
        N=5,
        M=10,
        perms=[[0, 1, 2, 3, 4], [1, 0, 2, 3, 4], [0, 1, 3, 2, 4], [1, 0, 3, 2, 4]],
        sig=20,
        E_cstr=False,
    ),
    'n5_p2_ecstr': dict(N=5, M=8, perms=[[0, 1, 2, 3, 4], [0, 2, 1, 3, 4]], sig=15, E_cstr=True),
    'n4_p6_pbc': dict(
        N=4,
        M=9,
        perms=[[0, 1, 2, 3], [1, 2, 0, 3], [2, 0, 1, 3], [1, 0, 2, 3], [0, 2, 1, 3], [2, 1, 0, 3]],
        sig=12,
        E_cstr=False,
        lattice=[[6.0, 0.3, 0.0], [0.0, 5.5, 0.2], [0.1, 0.0, 7.0]],
    ),
    'n9_p1': dict(N=9, M=9, perms=[list(range(9))], sig=30, E_cstr=False),
    # round 6: a WELL-CONDITIONED periodic case (lam = 1e-4: max|J alpha| ~ 1e2..1e4, so the round-off floor of the prediction
    # sum is <= 1e-12 and the minimum-image prologue of desc / predict is pinned at 1e-10; n4_p6_pbc stays as the
    # ill-conditioned one).  The 5.2 .. 6.4 A cell is smaller than twice the molecule: most frames wrap several pairs.
    'n10_p2_pbc': dict(
        N=10,
        M=12,
        perms=[list(range(10)), [1, 0] + list(range(2, 10))],
        sig=14,
        E_cstr=False,
        lam=1e-4,
        lattice=[[5.6, 0.4, 0.0], [0.0, 5.2, 0.3], [0.2, 0.0, 6.4]],
    ),
}


def main():
    GDMLTrain, GDMLPredict, Desc, Iterative = _import_reference()
    gdml_train = GDMLTrain(max_processes=1)

    only = sys.argv[1:]  # e.g. `make_golden.py n10_p2_pbc`: regenerate only the named cases
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
        N, M = cfg['N'], cfg['M']
        seed = abs(hash(name)) % 1000 if False else sum(map(ord, name))
        np.random.seed(seed)
        ds = orc.synth_dataset(N, M + 7, seed=seed, jitter=0.25)
        R_all, E_all, F_all = ds['R'], ds['E'], ds['F']
        R_train, E_train, F_train = R_all[:M], E_all[:M], F_all[:M]
        R_test = R_all[M:]
        perms = np.array(cfg['perms'], dtype=np.int64)
        sig, lam = cfg['sig'], cfg.get('lam', 1e-10)
        use_E_cstr = cfg['E_cstr']

        task = {
            'type': 't',
            'code_version': '1.0.3',
            'dataset_name': np.array('synth'),
            'dataset_theory': np.array('pair'),
            'z': ds['z'],
            'R_train': R_train,
            'F_train': F_train,
            'E_train': E_train,
            'idxs_train': np.arange(M),
            'md5_train': 'x',
            'idxs_valid': np.arange(M, M + 7),
            'md5_valid': 'x',
            'sig': sig,
            'lam': lam,
            'use_E': True,
            'use_E_cstr': use_E_cstr,
            'use_sym': perms.shape[0] > 1,
            'perms': perms,
        }
        lat_and_inv = None
        if 'lattice' in cfg:
            lat = np.array(cfg['lattice'])
            task['lattice'] = lat
            lat_and_inv = (lat, np.linalg.inv(lat))

        desc = Desc(N, max_processes=1)
        tril_perms = np.array([Desc.perm(p) for p in perms])
        tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
        R_desc, R_d_desc = desc.from_R(R_train.reshape(M, -1), lat_and_inv=lat_and_inv)

        K = gdml_train._assemble_kernel_mat(
            R_desc, R_d_desc, tril_perms_lin, sig, desc, use_E_cstr=use_E_cstr
        ).copy()

        n = K.shape[0]
        rs = np.random.RandomState(seed + 1)
        col_idxs = np.sort(rs.choice(n, min(2 * 3 * N + 1, n // 2), replace=False))
        extra = len(col_idxs)
        K_cols = gdml_train._assemble_kernel_mat(
            R_desc,
            R_d_desc,
            tril_perms_lin,
            sig,
            desc,
            use_E_cstr=use_E_cstr,
            col_idxs=col_idxs,
            alloc_extra_rows=extra,
        ).copy()[:n]
        stop_pts = 3
        K_slice = gdml_train._assemble_kernel_mat(
            R_desc,
            R_d_desc,
            tril_perms_lin,
            sig,
            desc,
            use_E_cstr=use_E_cstr,
            col_idxs=np.s_[: stop_pts * 3 * N],
        ).copy()

        # full training through the reference (analytic solver)
        model = gdml_train.train(task)
        assert model['solver_name'] == 'analytic'

        # the label vector exactly as train.py:939-947 builds it
        y = F_train.ravel().copy()
        if use_E_cstr:
            y = np.hstack((y, -E_train + np.mean(E_train)))
        y_std = np.std(y)
        y = y / y_std
        alphas = np.hstack((model['alphas_F'], model['alphas_E'])) if use_E_cstr else model['alphas_F']

        pred = GDMLPredict(model, max_processes=1, use_torch=False)
        E_test, F_test = pred.predict(R_test.reshape(len(R_test), -1))
        pred.set_R_desc(R_desc)
        pred.set_R_d_desc(R_d_desc)
        E_tr, F_tr = pred.predict()

        # matrix-free K v (iterative.py:183-204) on a random vector
        it = Iterative(gdml_train, desc, None, 1, False)
        K_op = it._init_kernel_operator(task, R_desc, R_d_desc, tril_perms_lin, lam, n)
        v = rs.normal(size=n)
        K_op.matvec(v)  # primes (first call returns v)
        Kv = K_op.matvec(v)

        # Nystroem factor on the index-list columns
        L_inv_K_mn = it._nystroem_cholesky_factor(
            R_desc, R_d_desc, tril_perms_lin, sig, lam, use_E_cstr=use_E_cstr, col_idxs=col_idxs
        ).copy()

        out = dict(
            R_train=R_train,
            E_train=E_train,
            F_train=F_train,
            R_test=R_test,
            perms=perms,
            sig=np.float64(sig),
            lam=np.float64(lam),
            use_E_cstr=np.bool_(use_E_cstr),
            R_desc=R_desc,
            R_d_desc=R_d_desc,
            tril_perms_lin=tril_perms_lin,
            K=K,
            col_idxs=col_idxs,
            K_cols=K_cols,
            K_slice=K_slice,
            y=y,
            y_std=np.float64(y_std),
            alphas=alphas,
            model_R_desc=model['R_desc'],
            model_R_d_desc_alpha=model['R_d_desc_alpha'],
            model_c=np.float64(model['c']),
            model_std=np.float64(model['std']),
            E_test=E_test,
            F_test=F_test,
            E_train_pred=E_tr,
            F_train_pred=F_tr,
            v=v,
            Kv=Kv,
            L_inv_K_mn=L_inv_K_mn,
        )
        if use_E_cstr:
            out['model_alphas_E'] = model['alphas_E']
        if 'lattice' in cfg:
            out['lattice'] = np.array(cfg['lattice'])
            # how much of the case actually exercises the minimum-image convention (desc.py:44-77)
            x_open, _ = desc.from_R(np.vstack([R_train, R_test]).reshape(M + len(R_test), -1))
            x_pbc, _ = desc.from_R(np.vstack([R_train, R_test]).reshape(M + len(R_test), -1), lat_and_inv=lat_and_inv)
            print('   pairs wrapped by the minimum image: %.1f %%, max|J alpha| %.3e'
                  % (100.0 * np.mean(np.abs(x_open - x_pbc) > 1e-9), np.abs(model['R_d_desc_alpha']).max()))
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print(name, 'n=%d' % n, 'K max', np.abs(K).max(), '->', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
