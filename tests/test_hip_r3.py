"""GPU parity, round 3 (all through the C ABI), against fixtures produced by the REFERENCE
(tests/golden/make_golden_r3.py):
  * the policy paths of the iterative solver the M = 5000 workload takes: restart with 1.2x inducing points,
    checkpoint models (incl. their integration constant), create_task_from_model -> warm start;
  * the `sgdml all` loop (sweep.sigma_sweep) against the unmodified reference CLI: permutations found, sampled
    indices, validation table, early stop, selected sigma, test errors;
  * jitter escalation of the stabilised Cholesky (forced failures), near-duplicate query geometries."""
import os

import numpy as np
import pytest

from oracle import gdml_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


class FakeClock(object):
    """Stand-in for the `timeit` module inside solvers/iterative.py: every default_timer() call advances by `step`
    seconds, so the "every two minutes" checkpoint cadence (iterative.py:675-680) becomes a fixed iteration stride."""

    def __init__(self, step):
        self.t, self.step = 0.0, step

    def default_timer(self):
        self.t += self.step
        return self.t


def _task(fx, M, **extra):
    N = fx['R_all'].shape[1]
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': fx['z'], 'R_train': fx['R_all'][:M], 'F_train': fx['F_all'][:M], 'E_train': fx['E_all'][:M],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(M, M), 'md5_valid': 'x',
        'sig': int(fx['sig']), 'lam': float(fx['lam']), 'use_E': True, 'use_E_cstr': False, 'use_sym': False,
        'perms': np.arange(N)[None, :],
    }
    task.update(extra)
    return task


def _predict_F(model, R):
    from sgdml_amd.predict import GDMLPredict

    pred = GDMLPredict(model)
    return pred.predict(R.reshape(len(R), -1))


def test_iterative_restart_and_checkpoints_follow_reference(monkeypatch):
    """The reference's own run (k = 1 inducing point): 100 stagnating steps -> CGRestartException -> ceil(1.2 k) = 2
    inducing points drawn from the leverage scores of the first preconditioner -> convergence after 2176 iterations.
    Ours, same seed: same inducing columns in both stages, restart after the same 100 steps, iteration count within
    10 %, converged, same predictions; checkpoints every 100 iterations carry solver_iters / solver_resid /
    inducing columns / c consistent with the run, and agree with the reference's checkpoints at the same iteration."""
    from sgdml_amd.solvers import iterative as it_mod
    from sgdml_amd.solvers.iterative import Iterative
    from sgdml_amd.train import GDMLTrain

    fx = load('pcg_restart')
    M, N = int(fx['n_train']), fx['R_all'].shape[1]
    stages, ckpts, hist, cg_starts = [], [], [], []
    orig_ind = Iterative.inducing_pts_from_lev_scores

    def spy_ind(self, lev_scores, n):
        idx = orig_ind(self, lev_scores, n)
        stages.append(np.array(idx))
        return idx

    monkeypatch.setattr(Iterative, 'inducing_pts_from_lev_scores', spy_ind)
    monkeypatch.setattr(it_mod, 'timeit', FakeClock(1.2))
    tr = GDMLTrain()
    try:
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = int(fx['k0'])
        tr._emulate_ref_rng = True  # replay of a freshly installed reference: the restart's inducing columns depend on it
        ctx = tr._context()
        orig_pcg = ctx.pcg

        def spy_pcg(*a, **kw):
            cg_starts.append(len(hist))
            cb = kw['callback']
            kw['callback'] = lambda it, r, fetch_x: (hist.append(r), cb(it, r, fetch_x))[1]
            return orig_pcg(*a, **kw)

        ctx.pcg = spy_pcg
        np.random.seed(int(fx['seed']))
        model = tr.train(_task(fx, M), save_progr_callback=lambda m: ckpts.append(dict(m)))
    finally:
        tr.__del__()

    # ---- stages and restart
    assert len(stages) == int(fx['n_stages']) == 2
    assert np.array_equal(stages[0], fx['inducing_stage0'])
    assert np.array_equal(stages[1], fx['inducing_stage1'])
    assert np.array_equal(model['inducing_pts_idxs'], fx['final_inducing'])
    assert cg_starts == list(fx['cg_starts'])  # second CG call after exactly 100 steps, like the reference
    n_ref = int(fx['n_iters'])
    assert abs(int(model['solver_iters']) - n_ref) <= n_ref // 10, (model['solver_iters'], n_ref)
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
    assert abs(model['norm_y_train'] - float(fx['norm_y_train'])) <= 1e-9 * float(fx['norm_y_train'])
    # residual history: the first steps to rounding, the stagnating first stage within the drift of two PCG runs
    ref_hist = fx['resid_hist']
    ours = np.array(hist)  # callback k reports ||r_k|| after update k, like the reference's spy on scipy's cg
    np.testing.assert_allclose(ours[:8], ref_hist[:8], rtol=1e-6)
    np.testing.assert_allclose(ours[:99], ref_hist[:99], rtol=0.15)
    # ---- final model predicts like the reference's
    E, F = _predict_F(model, fx['R_test'])
    assert np.abs(F - fx['F_test']).max() <= 5e-3 * np.abs(fx['F_test']).max()
    assert abs(model['c'] - float(fx['model_c'])) <= 5e-3
    # ---- checkpoints
    # under the fake clock tt = 1.2 s up to rounding, so ceil(120 / tt) flips between 100 and 101 with the float noise of
    # the running clock value: the reference wrote checkpoints at 301 ... 801, 1011, 2021 -- and so do we, the timer being
    # read in the same places (iterative.py:618-621, :675-680)
    its = [int(c['solver_iters']) for c in ckpts]
    assert its == [int(v) for v in fx['ckpt_iters_all']], its
    task = _task(fx, M)
    for c in (ckpts[2], ckpts[-1]):
        assert c['solver_name'] == 'cg' and c['solver_tol'] == 1e-4
        assert abs(c['solver_resid'] - hist[int(c['solver_iters']) - 1]) <= 1e-12 * hist[0]
        k_stage = 1 if c['solver_iters'] <= 100 else 2
        assert len(c['inducing_pts_idxs']) == k_stage * 3 * N
        # c of a checkpoint = mean(E_train - E_pred) with the checkpoint's own coefficients (iterative.py:711-720)
        m0 = dict(c, c=0.0)
        E0, _ = _predict_F(m0, task['R_train'])
        assert abs(c['c'] - np.mean(task['E_train'] - E0)) <= 1e-9 * abs(c['c'])
    for j in fx['ckpt_keep'][:2]:  # reference checkpoints at iterations both runs wrote one
        it_ref = int(fx['ckpt%d_solver_iters' % j])
        mine = ckpts[its.index(it_ref)]
        assert abs(mine['solver_resid'] - float(fx['ckpt%d_solver_resid' % j])) <= 0.2 * float(fx['ckpt%d_solver_resid' % j])
        assert abs(mine['c'] - float(fx['ckpt%d_c' % j])) <= 0.1
        assert np.array_equal(mine['inducing_pts_idxs'], fx['ckpt%d_inducing_pts_idxs' % j])


def test_warm_start_from_reference_checkpoint():
    """create_task_from_model on the reference's checkpoint (iteration 701, k = 2) and train: the task carries
    alphas0_F / solver_iters / inducing_pts_idxs like the reference's, the inducing columns are reused (same k), the
    iteration counter continues, the run converges and predicts like the reference's resumed run."""
    from sgdml_amd.train import GDMLTrain

    fx, ws = load('pcg_restart'), load('pcg_warm_start')
    M, N = int(fx['n_train']), fx['R_all'].shape[1]
    j = int(ws['ckpt_index'])
    t0 = _task(fx, M)
    ck = {
        'idxs_train': t0['idxs_train'], 'e_err': {'mae': np.nan, 'rmse': np.nan}, 'perms': t0['perms'],
        'dataset_name': t0['dataset_name'], 'dataset_theory': t0['dataset_theory'], 'z': t0['z'],
        'md5_train': 'x', 'idxs_valid': t0['idxs_valid'], 'md5_valid': 'x', 'sig': t0['sig'], 'lam': t0['lam'],
        'use_E': True, 'alphas_F': fx['ckpt%d_alphas_F' % j], 'solver_iters': int(fx['ckpt%d_solver_iters' % j]),
        'inducing_pts_idxs': fx['ckpt%d_inducing_pts_idxs' % j],
    }
    ds = {'R': fx['R_all'], 'E': fx['E_all'], 'F': fx['F_all'], 'z': fx['z']}
    tr = GDMLTrain()
    try:
        task = tr.create_task_from_model(ck, ds)
        assert np.array_equal(task['alphas0_F'], ws['task_alphas0_F'])
        assert int(task['solver_iters']) == int(ws['task_solver_iters'])
        assert np.array_equal(task['inducing_pts_idxs'], ws['task_inducing'])
        assert task['use_E'] and not task['use_E_cstr'] and not task['use_sym']
        assert np.array_equal(task['R_train'], fx['R_all'][:M]) and np.array_equal(task['E_train'], fx['E_all'][:M])
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = int(ws['k'])
        np.random.seed(int(ws['seed']))
        model = tr.train(task)
    finally:
        tr.__del__()
    assert np.array_equal(model['inducing_pts_idxs'], ck['inducing_pts_idxs'])  # reuse branch (iterative.py:525-526)
    n_ref = int(ws['solver_iters'])
    assert int(model['solver_iters']) > ck['solver_iters']  # the counter continues from the checkpoint
    assert abs(int(model['solver_iters']) - n_ref) <= max(n_ref // 10, 50), (model['solver_iters'], n_ref)
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
    E, F = _predict_F(model, fx['R_test'])
    assert np.abs(F - ws['F_test']).max() <= 5e-3 * np.abs(ws['F_test']).max()
    assert abs(model['c'] - float(ws['model_c'])) <= 5e-3


def test_sigma_sweep_matches_reference_cli():
    """sweep.sigma_sweep against the unmodified `sgdml all` of the reference on the same dataset and seed: the
    permutation group found, the stratified train / validation samples, the validation table of every sigma the
    reference trained (it stopped early after sigma = 64: cli.py:1136-1147), the selected sigma (lowest validation
    force RMSE, cli.py:1871-1872) and the test errors of the selected model."""
    from sgdml_amd.sweep import sigma_sweep
    from sgdml_amd.train import GDMLTrain
    from sgdml_amd.utils import io

    fx = load('cli_sweep')
    ds = {'type': 'd', 'code_version': '1.0.3', 'name': np.array('rotors'), 'theory': np.array('toy'), 'z': fx['z'],
          'R': fx['R'], 'F': fx['F'], 'E': fx['E'], 'r_unit': 'Ang', 'e_unit': 'kcal/mol'}
    ds['md5'] = io.dataset_md5(ds)
    assert ds['md5'] == fx['dataset_md5'].item()
    tr = GDMLTrain()
    try:
        np.random.seed(int(fx['seed']))
        np.random.choice(len(ds['R']), 1)  # the CLI's "example geometry" banner draws one index first (cli.py:294, :642)
        best, table, timings = sigma_sweep(tr, ds, int(fx['n_train']), int(fx['n_valid']), int(fx['n_test']),
                                           sigs=[int(s) for s in fx['sigs']], emulate_cli_rng=True)
    finally:
        tr.__del__()
    assert np.array_equal(best['perms'], fx['perms'])
    assert np.array_equal(best['idxs_train'], fx['idxs_train'])
    assert np.array_equal(best['idxs_valid'], fx['idxs_valid'])
    ref = fx['table']
    assert [row[0] for row in table] == [int(s) for s in ref[:, 0]]  # same models trained: same early stop
    np.testing.assert_allclose(np.array(table)[:, 1:], ref[:, 1:], rtol=2e-3)  # alphas are conditioning-limited
    assert float(best['sig']) == float(fx['best_sig'])
    assert int(best['n_test']) == int(fx['best_n_test'])
    np.testing.assert_allclose([best['e_err']['mae'], best['e_err']['rmse']], fx['best_e_err'], rtol=2e-3)
    np.testing.assert_allclose([best['f_err']['mae'], best['f_err']['rmse']], fx['best_f_err'], rtol=2e-3)
    assert abs(best['c'] - float(fx['best_c'])) <= 2e-4 * max(1.0, abs(float(fx['best_c'])))


def test_jitter_escalation_of_the_stabilised_cholesky():
    """_cho_factor_stable (iterative.py:414-471): with the first k factorisation attempts of K_mm declared failed
    (option nys.force_fail) the matrix carries eps + 1e-15 + ... + 10^(k-16) on its diagonal, cumulatively, like the
    reference's loop; leverage scores and preconditioner must equal the oracle's for that matrix.  With all 17
    attempts failed the reference gives up ("Failed to factorize despite strong regularization")."""
    import scipy.linalg as sla

    from sgdml_amd import _lib

    g = load('n6_p1')
    lam, sig = float(g['lam']), float(g['sig'])
    tp = orc.tril_perms_from_lin(g['tril_perms_lin'], g['R_desc'].shape[1])
    idx = g['col_idxs']
    K_nm = orc.assemble_K(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], sig, col_idxs=idx)
    c = _lib.Context()
    try:
        c.train_upload(g['R_desc'], g['R_d_desc'], tp)
        for k in (0, 3, 12):
            c.set_option('nys.force_fail', k)
            c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
            lev, fac, info = c.nystroem_factor(lam, idx, want_factor=True)
            assert info >> 8 >= k and (info & 1) == 0
            # oracle for the matrix the escalation arrives at
            K_mm = -K_nm[idx, :].copy()
            jit = np.finfo(float).eps + sum(10.0 ** mag for mag in range(-15, -15 + (info >> 8)))
            K_mm[np.diag_indices_from(K_mm)] += jit
            L = sla.cholesky(K_mm, lower=True)
            X = sla.solve_triangular(L, K_nm.T, lower=True).T
            inner = X.T @ X + lam * np.eye(len(idx))
            L2 = sla.cholesky(inner, lower=True)
            ref = sla.solve_triangular(L2, X.T, lower=True)  # m x n
            P_ref, P = ref.T @ ref, fac.T @ fac
            assert np.abs(P - P_ref).max() <= 1e-7 * np.abs(P_ref).max(), k
            np.testing.assert_allclose(lev, (ref**2).sum(0), rtol=1e-6)
        c.set_option('nys.force_fail', 17)
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        with pytest.raises(Exception, match='despite strong regularization'):
            c.nystroem_factor(lam, idx)
    finally:
        c.close()


@pytest.mark.parametrize('n_atoms,n_train,n_query', [(21, 300, 512), (9, 600, 300), (21, 40, 7)])
def test_predict_near_duplicate_queries(n_atoms, n_train, n_query):
    """Query geometries 1e-3 ... 1e-5 Angstrom away from training geometries (and exact copies): the MFMA kernel forms
    |d|^2 = |x|^2 + |X|^2 - 2 x.X, whose cancellation is worst here.  north_star's tolerance: forces 1e-10 of the
    largest force, energies 1e-10 max(1, |E|), against the oracle."""
    from sgdml_amd import _lib

    ds = orc.synth_dataset(n_atoms, n_train, seed=11, jitter=0.3)
    R = ds['R']
    xo, go = orc.desc_from_R(R.reshape(n_train, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(n_atoms)[None])
    rs = np.random.RandomState(3)
    alphas = rs.normal(size=(n_train, 3 * n_atoms))
    ja = orc.d_desc_dot_vec(go, alphas)
    sig = 15.0
    src = rs.randint(0, n_train, n_query)
    scale = 10.0 ** rs.choice([-3.0, -4.0, -5.0, -np.inf], n_query)  # -inf: exact copies
    Rq = R[src] + scale[:, None, None] * rs.normal(size=(n_query, n_atoms, 3))
    xq, gq = orc.desc_from_R(Rq.reshape(n_query, -1))
    E_ref, F_ref = orc.predict_from_desc(xq, gq, xo, ja, tp, sig)
    c = _lib.Context()
    try:
        c.predict_upload_model(xo, ja, tp, sig, None)
        E, F = c.predict(Rq.reshape(n_query, -1))
    finally:
        c.close()
    assert np.abs(F - F_ref).max() <= 1e-10 * np.abs(F_ref).max()
    assert np.abs(E - E_ref).max() <= 1e-10 * max(1.0, np.abs(E_ref).max())


def test_configs2_trajectory_workload_vs_reference():
    """bench.py's configs[2] workload family (synth_trajectory, N = 21) at N_train = 300, where the reference's CPU path
    converges in minutes (make_golden_r3.case_cfg2_traj_m300: k = 30 inducing points drawn by the reference, 183
    iterations to solver_tol 1e-4): with the reference's inducing columns our PCG needs the same number of iterations
    (+-10 %), follows its residual history and gives the same predictions."""
    from sgdml_amd import _lib
    from sgdml_amd.utils.desc import Desc

    g = load('cfg2_traj_m300')
    M, N = g['R_train'].shape[:2]
    sig, lam, y = float(g['sig']), float(g['lam']), g['y']
    c = _lib.Context()
    try:
        xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
        tp = orc.tril_perms_from_atom_perms(g['perms'])
        idx = g['inducing_pts_idxs']
        c.train_upload(xd, gd, tp)
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        c.nystroem_factor(lam, idx)
        c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
        hist = []
        x, info, iters, resid = c.pcg(lam, False, y, rtol=1e-4, maxiter=5000, callback=lambda it, r, fetch_x: hist.append(r) or False)
        assert info == 0
        n_ref = int(g['n_iters'])
        assert abs(iters - n_ref) <= max(2, n_ref // 10), (iters, n_ref)
        ref = g['resid_hist']
        ours = np.array(hist)
        np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-6)
        # beyond the first steps two correct PCG runs on this cond ~ 1e10 system decorrelate pointwise (a different
        # summation order in a dot product is enough): compare WHEN the residual first passes each level instead
        from _pcg_compare import assert_same_convergence
        assert_same_convergence(ours, ref, np.linalg.norm(y))
        d = Desc(N)
        F = []
        for coeffs in (g['alphas'], -x):
            c.predict_upload_model(xd, d.d_desc_dot_vec(gd, coeffs.reshape(M, -1)), tp, sig, None)
            F.append(c.predict(g['R_test'].reshape(len(g['R_test']), -1))[1])
        assert np.abs(F[1] - F[0]).max() <= 5e-3 * np.abs(F[0]).max()
        assert np.abs(F[0] * float(g['y_std']) - g['F_test']).max() <= 1e-8 * np.abs(g['F_test']).max()
    finally:
        c.close()


def _group(n_atoms, gens):
    idt = tuple(range(n_atoms))
    G, frontier = {idt}, [idt]
    while frontier:
        nxt = []
        for g in frontier:
            for h in gens:
                c = tuple(g[i] for i in h)
                if c not in G:
                    G.add(c)
                    nxt.append(c)
        frontier = nxt
    rest = sorted(G - {idt})
    return np.array([idt] + rest)


@pytest.mark.parametrize('n_atoms,n_train,n_gen', [(21, 7, 2), (12, 8, 3), (9, 10, 2), (16, 5, 1), (8, 11, 2)])
def test_assemble_pts_kernel_vs_oracle(n_atoms, n_train, n_gen):
    """assemble_pts.hip (8 <= N <= 21 with a permutation group: whole-point strips, producer / consumer wavefronts, LDS-DMA row
    images) against the oracle in every dense mode it serves: full K, full K with the energy-constraint rows and columns,
    the lower form A = -K + lam I of the analytic solver, a range of column points; n_train is not a multiple of the points
    per strip, groups of 2, 4, 6 and 27 permutations (one and several steps per row point).  1e-12 of max|K|."""
    from sgdml_amd import _lib

    N, M = n_atoms, n_train
    idt = list(range(N))

    def swap(a):
        p = idt[:]
        p[a], p[a + 1] = p[a + 1], p[a]
        return tuple(p)

    def rot3(a):
        p = idt[:]
        p[a], p[a + 1], p[a + 2] = p[a + 1], p[a + 2], p[a]
        return tuple(p)

    gens = {1: [swap(0)], 2: [swap(0), swap(N - 2)], 3: [rot3(0), rot3(3), rot3(N - 3)]}[n_gen]
    perms = _group(N, gens)
    ds = orc.synth_dataset(N, M, seed=5, jitter=0.3)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(perms)
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    sig, lam = 13.0, 1e-7
    Ko = orc.assemble_K(xo, go, lin, sig)
    KoE = orc.assemble_K(xo, go, lin, sig, use_E_cstr=True)
    scale = np.abs(Ko).max()
    n, N3 = M * 3 * N, 3 * N
    c = _lib.Context()
    try:
        c.set_option('asm.pts', 2)
        c.train_upload(xo, go, tp)
        K = c.assemble_K(sig, False, to_host=True)
        assert np.abs(K - Ko).max() <= 1e-12 * scale
        KE = c.assemble_K(sig, True, to_host=True)
        assert np.abs(KE - KoE).max() <= 1e-12 * np.abs(KoE).max()
        c.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
        A = c.K_to_host()[:n]
        low = np.kron(np.tril(np.ones((M, M))), np.ones((N3, N3))).astype(bool)
        assert np.abs((A - (-Ko + lam * np.eye(n)))[low]).max() <= 1e-12 * scale
        p0, p1 = M // 3, M // 3 + max(1, M // 2)
        Kp = c.assemble_K(sig, False, points=(p0, p1), to_host=True)
        assert np.abs(Kp - Ko[:, p0 * N3:p1 * N3]).max() <= 1e-12 * scale
        # the same through the general kernel: the two agree to rounding
        c.set_option('asm.pts', 0)
        K2 = c.assemble_K(sig, False, to_host=True)
        assert np.abs(K2 - K).max() <= 1e-13 * scale
    finally:
        c.close()
