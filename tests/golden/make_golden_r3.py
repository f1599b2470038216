#!/usr/bin/env python
"""
Round-3 golden fixtures: outputs of the REFERENCE for the policy paths of the iterative solver, the `sgdml all`
assistant, and mid-size instances of the BASELINE configuration shapes.  Build container only:

    python tests/golden/make_golden_r3.py [case ...]      (no argument = all cases)

Cases (files next to this script):
  pcg_restart      GDMLTrain.train -> Iterative.solve (iterative.py:473-825) on a system where the reference's own
                   restart policy fires (k = 1 inducing point, 100 stagnating steps, restart with ceil(1.2 k) = 2) and the
                   run then converges; with a deterministic clock (1.2 s per timer call -> a checkpoint every 100
                   iterations, iterative.py:675-735): inducing columns of every stage, residual norm after every iteration,
                   the iteration at which each CG call started, checkpoint models (alphas, c, solver_iters, solver_resid),
                   final model and predictions.
  pcg_warm_start   create_task_from_model (train.py:649-725) on one of those checkpoints -> train: the run resumes from
                   alphas0_F / solver_iters / inducing_pts_idxs (the reuse branch iterative.py:525-526) and converges.
  cli_sweep        the unmodified `sgdml all` (cli.py:612-742) on a synthetic dataset with a nonlinear pair potential
                   (so that the validation error has a minimum in sigma and the early stop cli.py:1136-1147 fires):
                   permutations found, sampled indices, full-precision validation table, selected sigma, test errors.
  cfg1_n21_m100    configs[1] molecule size at M = 100 (n = 6300): sampled K entries, solve, predictions.
  cfg3_n42_p27_m60 configs[3] molecule and permutation group at M = 60 (n = 7560).
  n100_m3          a 100-atom molecule (configs[4] ">= 100 atoms"): K, solve, predictions.
"""
import inspect
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import make_golden_r2 as g2  # noqa: E402  (reference loader, task builder, _train_and_sample)
from oracle import gdml_oracle as orc  # noqa: E402  (synthetic data generator only)


class FakeClock(object):
    """timeit.default_timer stand-in: every call advances by `step` seconds."""

    def __init__(self, step):
        self.t, self.step = 0.0, step

    def __call__(self):
        self.t += self.step
        return self.t


def _iter_setup(N, M, sig, lam, seed, jitter, n_extra=10):
    r = g2.ref()
    Desc = r['Desc']
    ds = orc.synth_dataset(N, M + n_extra, seed=seed, jitter=jitter)
    perms = np.arange(N)[None, :]
    task = g2.make_task(ds, M, perms, sig, lam)
    return ds, task


def _spy_cg(it_mod, hist, starts):
    real_cg = it_mod.sp.sparse.linalg.cg

    def spy_cg(A, b, x0=None, M=None, rtol=1e-5, atol=0.0, maxiter=None, callback=None):
        starts.append(len(hist))

        def cb(xk):
            fl = inspect.currentframe().f_back.f_locals  # scipy's cg frame
            hist.append(float(np.linalg.norm(fl['r'])))
            callback(xk)

        return real_cg(A, b, x0=x0, M=M, rtol=rtol, atol=atol, maxiter=maxiter, callback=cb)

    return real_cg, spy_cg


def _ckpt_fields(m):
    return dict(alphas_F=np.array(m['alphas_F']), c=np.float64(m['c']), solver_iters=np.int64(m['solver_iters']),
                solver_resid=np.float64(m['solver_resid']), inducing_pts_idxs=np.asarray(m['inducing_pts_idxs']),
                norm_y_train=np.float64(m['norm_y_train']), std=np.float64(m['std']))


def case_pcg_restart():
    r = g2.ref()
    GDMLPredict, Iterative, gt = r['GDMLPredict'], r['Iterative'], r['train']
    import sgdml.solvers.iterative as it_mod

    N, M, sig, lam, k0 = 9, 150, 10, 1e-10, 1
    ds, task = _iter_setup(N, M, sig, lam, seed=31, jitter=0.3)
    hist, starts, ckpts, inducing_stages = [], [], [], []
    real_cg, spy = _spy_cg(it_mod, hist, starts)
    it_mod.sp.sparse.linalg.cg = spy
    orig_k = Iterative.max_n_inducing_pts
    Iterative.max_n_inducing_pts = staticmethod(lambda n_train, n_atoms, mb: k0)
    orig_ind = Iterative.inducing_pts_from_lev_scores

    def spy_ind(self, lev_scores, n):
        idx = orig_ind(self, lev_scores, n)
        inducing_stages.append(np.array(idx))
        return idx

    Iterative.inducing_pts_from_lev_scores = spy_ind
    real_timer = it_mod.timeit.default_timer
    it_mod.timeit.default_timer = FakeClock(1.2)
    old_mem = gt._max_memory
    gt._max_memory = 1e-6  # GB: the analytic solver "does not fit" -> train() takes the iterative branch (train.py:961-963)
    np.random.seed(5)
    t0 = time.time()
    try:
        model = gt.train(task, save_progr_callback=lambda m: ckpts.append(_ckpt_fields(m)))
    finally:
        it_mod.sp.sparse.linalg.cg = real_cg
        Iterative.max_n_inducing_pts = orig_k
        Iterative.inducing_pts_from_lev_scores = orig_ind
        it_mod.timeit.default_timer = real_timer
        gt._max_memory = old_mem
    dt = time.time() - t0
    n_iters = int(model['solver_iters'])
    print('  pcg_restart: iters=%d resid=%.3e cg calls at %s k per stage %s checkpoints %d  %.1fs' % (
        n_iters, model['solver_resid'], starts, [len(s) // (3 * N) for s in inducing_stages], len(ckpts), dt), flush=True)
    assert len(starts) >= 2, 'the reference did not restart'
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train'], 'the reference did not converge'
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    out = dict(R_all=ds['R'], E_all=ds['E'], F_all=ds['F'], z=ds['z'], n_train=np.int64(M), sig=np.float64(sig),
               lam=np.float64(lam), k0=np.int64(k0), seed=np.int64(5), clock_step=np.float64(1.2),
               cg_starts=np.array(starts), resid_hist=np.array(hist), n_iters=np.int64(n_iters),
               solver_resid=np.float64(model['solver_resid']), solver_tol=np.float64(model['solver_tol']),
               norm_y_train=np.float64(model['norm_y_train']), alphas=np.array(model['alphas_F']),
               model_c=np.float64(model['c']), model_std=np.float64(model['std']),
               final_inducing=np.asarray(model['inducing_pts_idxs']), n_stages=np.int64(len(inducing_stages)),
               n_ckpt=np.int64(len(ckpts)), R_test=Rt, E_test=E_test, F_test=F_test, ref_seconds=np.float64(dt))
    for s, idx in enumerate(inducing_stages):
        out['inducing_stage%d' % s] = idx
    keep = sorted(set([0, len(ckpts) // 2, len(ckpts) - 1]))
    out['ckpt_keep'] = np.array(keep)
    out['ckpt_iters_all'] = np.array([c['solver_iters'] for c in ckpts])
    for j in keep:
        for key, v in ckpts[j].items():
            out['ckpt%d_%s' % (j, key)] = v
    g2.save('pcg_restart', **out)
    return ds, task, ckpts, model


def case_pcg_warm_start():
    """Resume from the middle checkpoint of pcg_restart (k = 2 after the restart; the reuse branch needs the memory model
    to return that k)."""
    r = g2.ref()
    GDMLPredict, Iterative, gt = r['GDMLPredict'], r['Iterative'], r['train']
    import sgdml.solvers.iterative as it_mod

    fx = np.load(os.path.join(HERE, 'pcg_restart.npz'))
    N = fx['R_all'].shape[1]
    M = int(fx['n_train'])
    j = int(fx['ckpt_keep'][1])
    ds = {'R': fx['R_all'], 'E': fx['E_all'], 'F': fx['F_all'], 'z': fx['z']}
    task0 = g2.make_task(ds, M, np.arange(N)[None, :], float(fx['sig']), float(fx['lam']))
    # the checkpoint as a model dictionary: what create_task_from_model reads (train.py:676-725)
    ck = {
        'idxs_train': task0['idxs_train'], 'e_err': {'mae': np.nan, 'rmse': np.nan}, 'perms': task0['perms'],
        'dataset_name': task0['dataset_name'], 'dataset_theory': task0['dataset_theory'], 'z': task0['z'],
        'md5_train': task0['md5_train'], 'idxs_valid': task0['idxs_valid'], 'md5_valid': task0['md5_valid'],
        'sig': task0['sig'], 'lam': task0['lam'], 'use_E': True,
        'alphas_F': fx['ckpt%d_alphas_F' % j], 'solver_iters': int(fx['ckpt%d_solver_iters' % j]),
        'inducing_pts_idxs': fx['ckpt%d_inducing_pts_idxs' % j],
    }
    k_ck = len(ck['inducing_pts_idxs']) // (3 * N)
    task = gt.create_task_from_model(ck, ds)
    hist, starts = [], []
    real_cg, spy = _spy_cg(it_mod, hist, starts)
    it_mod.sp.sparse.linalg.cg = spy
    orig_k = Iterative.max_n_inducing_pts
    Iterative.max_n_inducing_pts = staticmethod(lambda n_train, n_atoms, mb: k_ck)
    old_mem = gt._max_memory
    gt._max_memory = 1e-6
    np.random.seed(6)
    t0 = time.time()
    try:
        model = gt.train(task)
    finally:
        it_mod.sp.sparse.linalg.cg = real_cg
        Iterative.max_n_inducing_pts = orig_k
        gt._max_memory = old_mem
    dt = time.time() - t0
    print('  pcg_warm_start: from iteration %d (k=%d): solver_iters=%d (+%d) resid=%.3e cg calls at %s  %.1fs' % (
        ck['solver_iters'], k_ck, model['solver_iters'], len(hist), model['solver_resid'], starts, dt), flush=True)
    assert np.array_equal(np.asarray(model['inducing_pts_idxs']), ck['inducing_pts_idxs']), 'reuse branch not taken'
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = ds['R'][M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    g2.save('pcg_warm_start', ckpt_index=np.int64(j), k=np.int64(k_ck), seed=np.int64(6),
            task_alphas0_F=np.array(task['alphas0_F']), task_solver_iters=np.int64(task['solver_iters']),
            task_inducing=np.asarray(task['inducing_pts_idxs']), resid_hist=np.array(hist), cg_starts=np.array(starts),
            solver_iters=np.int64(model['solver_iters']), solver_resid=np.float64(model['solver_resid']),
            norm_y_train=np.float64(model['norm_y_train']), solver_tol=np.float64(model['solver_tol']),
            alphas=np.array(model['alphas_F']), model_c=np.float64(model['c']), E_test=E_test, F_test=F_test,
            ref_seconds=np.float64(dt))


def _nonlinear_dataset(n_frames, seed):
    """Two C3 rotors on a C-C axis (the molecule of make_golden_r2.case_perm_c3), energies a nonlinear function of the
    inverse distances, E = sum_k x_k + 0.35 sin(2.5 x_k), forces by the chain rule through the descriptor Jacobian."""
    base = [[0, 0, 0], [0, 0, 1.4]]
    for k in range(3):
        a = 2 * np.pi * k / 3
        base.append([np.cos(a), np.sin(a), -0.4])
    for k in range(3):
        a = 2 * np.pi * k / 3 + 0.3
        base.append([0.9 * np.cos(a), 0.9 * np.sin(a), 1.9])
    base = np.array(base)
    z = np.array([6, 6, 1, 1, 1, 1, 1, 1])
    rs = np.random.RandomState(seed)
    R = base[None] + rs.normal(scale=0.09, size=(n_frames,) + base.shape)
    x, g = orc.desc_from_R(R.reshape(n_frames, -1))
    J = orc.d_desc_from_comp(g)  # (M,D,3N)
    E = np.sum(x + 0.35 * np.sin(2.5 * x), axis=1)
    dE = 1.0 + 0.35 * 2.5 * np.cos(2.5 * x)
    F = -np.einsum('md,mdc->mc', dE, J).reshape(n_frames, -1, 3)
    return R, E, F, z


def case_cli_sweep():
    """Own process (the assistant creates its own GDMLTrain singleton): python make_golden_r3.py cli_sweep"""
    assert not g2._ref, 'run cli_sweep in a process of its own'
    scratch = tempfile.mkdtemp(prefix='sgdml_ref_')
    shutil.copytree(g2.REF, os.path.join(scratch, 'sgdml'))
    sys.dont_write_bytecode = True
    sys.path.insert(0, scratch)
    from sgdml import cli
    from sgdml.utils import io

    n_train, n_valid, n_test = 40, 30, 60
    sigs = [2, 4, 8, 16, 32, 64, 128]
    R, E, F, z = _nonlinear_dataset(200, seed=9)
    d = {'type': 'd', 'code_version': '1.0.3', 'name': np.array('rotors'), 'theory': np.array('toy'), 'z': z, 'R': R,
         'F': F, 'E': E, 'r_unit': 'Ang', 'e_unit': 'kcal/mol'}
    d['md5'] = io.dataset_md5(d)
    work = tempfile.mkdtemp(prefix='sgdml_cli_')
    cwd = os.getcwd()
    os.chdir(work)
    table, selected = [], {}
    try:
        np.savez_compressed('rotors.npz', **d)
        orig_select = cli.select

        def spy_select(model_dir, overwrite, model_file=None, **kw):
            mdir, files = model_dir
            for f in files:
                with np.load(os.path.join(mdir, f), allow_pickle=True) as m:
                    e, ff = m['e_err'].item(), m['f_err'].item()
                    table.append((float(m['sig']), e['mae'], e['rmse'], ff['mae'], ff['rmse']))
                    if not selected:
                        selected.update(idxs_train=np.array(m['idxs_train']), idxs_valid=np.array(m['idxs_valid']),
                                        perms=np.array(m['perms']))
            return orig_select(model_dir, overwrite, model_file, **kw)

        cli.select = spy_select
        argv = sys.argv
        sys.argv = ['sgdml', 'all', 'rotors.npz', str(n_train), str(n_valid), str(n_test), '-s'] + [str(s) for s in sigs] + \
            ['--cpu', '-p', '1']
        np.random.seed(123)
        try:
            cli.main()
        finally:
            sys.argv = argv
            cli.select = orig_select
        finals = [f for f in os.listdir('.') if f.startswith('rotors-toy-train') and f.endswith('.npz')]
        assert len(finals) == 1, finals
        with np.load(finals[0], allow_pickle=True) as m:
            best = dict(sig=float(m['sig']), e_err=m['e_err'].item(), f_err=m['f_err'].item(), n_test=int(m['n_test']),
                        alphas_F=np.array(m['alphas_F']), c=float(m['c']), std=float(m['std']))
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    table.sort(key=lambda row: sigs.index(int(row[0])))
    print('  cli_sweep: trained sigmas %s of %s, selected %g, perms %s' % ([int(t[0]) for t in table], sigs, best['sig'],
                                                                          selected['perms'].shape), flush=True)
    for row in table:
        print('    sig %4g  e %.6f/%.6f  f %.6f/%.6f' % row)
    g2.save('cli_sweep', R=R, E=E, F=F, z=z, n_train=np.int64(n_train), n_valid=np.int64(n_valid), n_test=np.int64(n_test),
            sigs=np.array(sigs), seed=np.int64(123), table=np.array(table), best_sig=np.float64(best['sig']),
            best_e_err=np.array([best['e_err']['mae'], best['e_err']['rmse']]),
            best_f_err=np.array([best['f_err']['mae'], best['f_err']['rmse']]), best_n_test=np.int64(best['n_test']),
            best_alphas=best['alphas_F'], best_c=np.float64(best['c']), best_std=np.float64(best['std']),
            idxs_train=selected['idxs_train'], idxs_valid=selected['idxs_valid'], perms=selected['perms'],
            dataset_md5=np.array(d['md5']))


def case_cfg1_n21_m100():
    g2._train_and_sample('cfg1_n21_m100', 21, 100, np.arange(21)[None, :], 20, 20, seed=61, jitter=0.3, n_rows=128, n_cols=900)


def case_cfg3_n42_p27_m60():
    gens = []
    for a in (3, 17, 30):
        g = list(range(42))
        g[a], g[a + 1], g[a + 2] = a + 1, a + 2, a
        gens.append(tuple(g))
    perms = g2.group_closure(gens, 42)
    assert perms.shape[0] == 27
    g2._train_and_sample('cfg3_n42_p27_m60', 42, 60, perms, 40, 10, seed=62, jitter=0.3, n_rows=128, n_cols=900)


def case_n100_m3():
    g2._train_and_sample('n100_m3', 100, 3, np.arange(100)[None, :], 60, 4, seed=63, jitter=0.3, n_rows=160, n_cols=400)


def case_cfg2_traj_m300():
    """The synthetic configs[2] workload of bench.py (synth_trajectory: N = 21, low-dimensional "trajectory") at the
    largest N_train the reference's CPU path solves in minutes: its own Iterative.solve with its own leverage-score
    inducing points (memory budget chosen for k = 30), residual after every iteration, final alphas, predictions."""
    r = g2.ref()
    Desc, GDMLPredict, Iterative, gt = r['Desc'], r['GDMLPredict'], r['Iterative'], r['train']
    import sgdml.solvers.iterative as it_mod
    import bench

    N, M, sig, lam, k = 21, 300, 20, 1e-10, 30
    R, E, F = bench.synth_trajectory(N, M + 20, seed=3)
    ds = {'R': R, 'E': E, 'F': F, 'z': np.full(N, 6)}
    perms = np.arange(N)[None, :]
    task = g2.make_task(ds, M, perms, sig, lam)
    desc = Desc(N, max_processes=1)
    tril_perms = np.array([Desc.perm(p) for p in perms])
    tril_perms_lin = (tril_perms + np.arange(len(perms))[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(R[:M].reshape(M, -1))
    y = F[:M].ravel().copy()
    y_std = np.std(y)
    y /= y_std
    hist, starts = [], []
    real_cg, spy = _spy_cg(it_mod, hist, starts)
    it_mod.sp.sparse.linalg.cg = spy
    orig_k = Iterative.max_n_inducing_pts
    Iterative.max_n_inducing_pts = staticmethod(lambda n_train, n_atoms, mb: k)
    np.random.seed(7)
    t0 = time.time()
    try:
        it = Iterative(gt, desc, 1, 1, False)
        alphas, tol, n_iters, resid, train_rmse, inducing, is_conv = it.solve(
            task, R_desc, R_d_desc, tril_perms_lin, y, y_std, tol=1e-4)
    finally:
        it_mod.sp.sparse.linalg.cg = real_cg
        Iterative.max_n_inducing_pts = orig_k
    dt = time.time() - t0
    print('  cfg2_traj: k=%d iters=%d resid=%.3e conv=%s cg calls at %s  %.1fs' % (
        len(inducing) // (3 * N), n_iters, resid, is_conv, starts, dt), flush=True)
    model = gt.create_model(task, 'cg', R_desc, R_d_desc, tril_perms_lin, y_std, alphas)
    pred = GDMLPredict(model, max_processes=1, use_torch=False)
    Rt = R[M:]
    E_test, F_test = pred.predict(Rt.reshape(len(Rt), -1))
    g2.save('cfg2_traj_m300', R_train=R[:M], E_train=E[:M], F_train=F[:M], z=ds['z'], perms=perms, sig=np.float64(sig),
            lam=np.float64(lam), y=y, y_std=np.float64(y_std), inducing_pts_idxs=np.asarray(inducing), k=np.int64(k),
            resid_hist=np.array(hist), cg_starts=np.array(starts), n_iters=np.int64(n_iters), resid=np.float64(resid),
            is_conv=np.bool_(is_conv), alphas=alphas, R_test=Rt, E_test=E_test, F_test=F_test,
            ref_seconds=np.float64(dt), traj_seed=np.int64(3))


CASES = ['pcg_restart', 'pcg_warm_start', 'cli_sweep', 'cfg1_n21_m100', 'cfg3_n42_p27_m60', 'n100_m3', 'cfg2_traj_m300']

if __name__ == '__main__':
    todo = sys.argv[1:] or CASES
    for c in todo:
        print(c, flush=True)
        if c == 'cli_sweep' and len(todo) > 1:  # needs a process of its own
            import subprocess
            subprocess.check_call([sys.executable, os.path.abspath(__file__), 'cli_sweep'])
        else:
            globals()['case_' + c]()
