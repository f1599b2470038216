"""Host-side policy of the drop-in classes on the CPU: GDMLTrain.train -> Iterative.solve -> create_model -> _recov_int_const with
the GPU context replaced by a stand-in whose every numerical method is the NumPy oracle (same method names, argument meaning
and return values as sgdml_amd._lib.Context).  What runs here is exactly the Python that runs on the GPU box -- leverage
sampling, the restart policy, the checkpoint cadence, the lazy iterate fetch, the distributed-mode synchronisation points --
against the reference's own runs (fixtures of tests/golden/make_golden_r3.py).  The GPU suite runs the same scenarios through
the library (tests/test_hip_r3.py); this file keeps the host logic covered in the CPU-only check of every round."""
import contextlib
import os

import numpy as np
import pytest

from oracle import gdml_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


class OracleContext(object):
    """sgdml_amd._lib.Context with the oracle behind it (only what train.py / solvers / predict.py call)."""

    def __init__(self):
        self.calls = []
        self._bcast = None
        self.n_fetch = 0

    # -- plumbing
    def close(self):
        pass

    def mem_info(self):
        return 0, 64 * 2**30, 288 * 2**30

    def resident_K_bytes(self):
        return 0

    def comm_info(self):
        return (0, 1)

    def comm_suspended(self):
        return contextlib.nullcontext(self)

    # -- descriptors / residency
    def desc_from_R(self, R, n_atoms, lat_and_inv=None):
        return orc.desc_from_R(np.asarray(R).reshape(-1, 3 * n_atoms), lat_and_inv)

    def train_upload(self, R_desc, R_d_desc, tril_perms):
        self.xd, self.gd, self.tp = np.asarray(R_desc), np.asarray(R_d_desc), np.asarray(tril_perms)
        self.lin = orc.tril_perms_lin_from_tril_perms(self.tp)
        self.n_train, self.n_atoms = self.xd.shape[0], orc.n_atoms_from_dim_d(self.xd.shape[1])

    # -- assembly / Nystroem factor
    def assemble_K(self, sig, use_E_cstr=False, points=None, idx=None, alloc_extra_rows=0, to_host=False, for_cholesky=None):
        self.calls.append('assemble')
        self._asm = (float(sig), bool(use_E_cstr), None if idx is None else np.asarray(idx))
        if to_host:
            return orc.assemble_K(self.xd, self.gd, self.lin, sig, use_E_cstr, col_idxs=np.s_[:] if idx is None else idx,
                                  alloc_extra_rows=alloc_extra_rows)

    def nystroem_factor(self, lam, idx, want_factor=False, want_lev=True):
        self.calls.append('nystroem')
        sig, use_E, _ = self._asm
        self._fac = orc.nystroem_factor(self.xd, self.gd, self.lin, sig, lam, np.asarray(idx), use_E)
        return ((self._fac**2).sum(0) if want_lev else None), (self._fac if want_factor else None), 0

    def nystroem_lev_scores(self):  # the scores on demand (round 5: the main build no longer returns them)
        return (self._fac**2).sum(0)

    def precon_apply(self, lam, v):
        return orc.precon_apply(self._fac, lam, v)

    # -- prediction / operator
    def predict_upload_model(self, R_desc, R_d_desc_alpha, tril_perms, sig, alphas_E=None):
        self.m_xd, self.m_ja, self.m_tp, self.m_sig, self.m_aE = (np.asarray(R_desc), np.asarray(R_d_desc_alpha),
                                                                np.asarray(tril_perms), float(sig), alphas_E)
        self.model_n_train, self.model_n_atoms = self.m_xd.shape[0], orc.n_atoms_from_dim_d(self.m_xd.shape[1])

    def set_alphas(self, alphas_F, alphas_E=None):
        self.m_ja = orc.d_desc_dot_vec(self.gd, np.asarray(alphas_F).reshape(self.n_train, -1))
        self.m_aE = alphas_E

    def predict(self, R=None, lat_and_inv=None, return_E=True):
        self.calls.append('predict')
        if R is None:
            xq, gq = self.xd, self.gd
        else:
            xq, gq = orc.desc_from_R(np.asarray(R).reshape(-1, 3 * self.model_n_atoms), lat_and_inv)
        E, F = orc.predict_from_desc(xq, gq, self.m_xd, self.m_ja, self.m_tp, self.m_sig, self.m_aE)
        return (E if return_E else None), F

    def kernel_matvec(self, lam, use_E_cstr, v):
        return orc.kernel_matvec(self.xd, self.gd, self.tp, self.m_sig, lam, np.asarray(v), use_E_cstr)

    def pcg(self, lam, use_E_cstr, y, x0=None, rtol=1e-4, maxiter=1000, use_precon=True, callback=None, cb_every=1):
        """scipy-cg semantics like gdml_pcg; callback(it, ||r_it||, fetch_x) after every update."""
        self.calls.append('pcg')
        y = np.asarray(y, dtype=np.float64)
        x = np.zeros_like(y) if x0 is None else np.array(x0, dtype=np.float64)
        atol = rtol * np.linalg.norm(y)
        A = lambda v: -self.kernel_matvec(lam, use_E_cstr, v)  # noqa: E731
        r = y - A(x) if x0 is not None else y.copy()
        rho_prev, p, it = None, None, 0

        def fetch_x():
            self.n_fetch += 1
            return x.copy()

        for it in range(maxiter):
            rn = np.linalg.norm(r)
            if rn < atol:
                return x, 0, it, rn
            z = self.precon_apply(lam, r) if use_precon else r
            rho = r @ z
            p = z.copy() if it == 0 else z + (rho / rho_prev) * p
            q = A(p)
            alpha = rho / (p @ q)
            x += alpha * p
            r -= alpha * q
            rho_prev = rho
            if callback is not None and (it + 1) % cb_every == 0:
                if callback(it + 1, float(np.linalg.norm(r)), fetch_x):
                    return x, 2, it + 1, float(np.linalg.norm(r))
        return x, 1, maxiter, float(np.linalg.norm(r))


class FakeClock(object):
    def __init__(self, step):
        self.t, self.step = 0.0, step

    def default_timer(self):
        self.t += self.step
        return self.t


def _task(fx, M):
    N = fx['R_all'].shape[1]
    return {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': fx['z'], 'R_train': fx['R_all'][:M], 'F_train': fx['F_all'][:M], 'E_train': fx['E_all'][:M],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(M, M), 'md5_valid': 'x',
        'sig': int(fx['sig']), 'lam': float(fx['lam']), 'use_E': True, 'use_E_cstr': False, 'use_sym': False,
        'perms': np.arange(N)[None, :],
    }


@pytest.fixture
def trainer(monkeypatch):
    from sgdml_amd.train import GDMLTrain

    ctx = OracleContext()
    tr = GDMLTrain()
    monkeypatch.setattr(tr, '_context', lambda: ctx)
    yield tr, ctx
    tr._ctx = None
    tr.__del__()


def test_restart_policy_and_checkpoints_follow_the_reference_on_cpu(trainer, monkeypatch):
    """The reference's restarting run (k = 1 -> 2 inducing points): the same inducing columns in both stages, the restart after
    exactly 100 steps, checkpoints at the iterations the reference wrote them under the same fake clock, convergence within
    10 % of its iteration count -- with the iterate fetched from the 'device' only for the restart and the checkpoints."""
    from sgdml_amd.solvers import iterative as it_mod
    from sgdml_amd.solvers.iterative import Iterative

    tr, ctx = trainer
    fx = load('pcg_restart')
    M = int(fx['n_train'])
    stages, ckpts, cg_starts, hist = [], [], [], []
    orig_ind = Iterative.inducing_pts_from_lev_scores

    def spy_ind(self, lev_scores, n):
        idx = orig_ind(self, lev_scores, n)
        stages.append(np.array(idx))
        return idx

    monkeypatch.setattr(Iterative, 'inducing_pts_from_lev_scores', spy_ind)
    monkeypatch.setattr(it_mod, 'timeit', FakeClock(1.2))
    orig_pcg = ctx.pcg

    def spy_pcg(*a, **kw):
        cg_starts.append(len(hist))
        cb = kw['callback']
        kw['callback'] = lambda it, r, fetch_x: (hist.append(r), cb(it, r, fetch_x))[1]
        return orig_pcg(*a, **kw)

    ctx.pcg = spy_pcg
    tr._force_solver = 'cg'
    tr._force_n_inducing_pts = int(fx['k0'])
    tr._emulate_ref_rng = True
    np.random.seed(int(fx['seed']))
    model = tr.train(_task(fx, M), save_progr_callback=lambda m: ckpts.append(dict(m)))

    assert len(stages) == 2
    assert np.array_equal(stages[0], fx['inducing_stage0']) and np.array_equal(stages[1], fx['inducing_stage1'])
    assert cg_starts == list(fx['cg_starts'])  # the second CG call starts after exactly 100 steps
    n_ref = int(fx['n_iters'])
    assert abs(int(model['solver_iters']) - n_ref) <= n_ref // 10
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
    np.testing.assert_allclose(np.array(hist[:8]), fx['resid_hist'][:8], rtol=1e-6)
    its = [int(c['solver_iters']) for c in ckpts]
    assert its == [int(v) for v in fx['ckpt_iters_all']][:len(its)] and len(its) >= len(fx['ckpt_iters_all']) - 1
    # the iterate crossed the boundary once per checkpoint and once for the restart, not once per iteration
    assert ctx.n_fetch == len(ckpts) + 1
    # a checkpoint's integration constant is the mean error of its own coefficients (iterative.py:711-720)
    c0 = ckpts[0]
    ctx.set_alphas(c0['alphas_F'])
    E0, _ = ctx.predict(None)
    assert abs(c0['c'] - np.mean(fx['E_all'][:M] - E0 * c0['std'])) <= 1e-9 * abs(c0['c'])


def test_global_rng_is_left_alone_unless_the_emulation_is_asked_for(trainer, monkeypatch):
    """Iterative.solve draws from np.random exactly what the reference's GPU path draws (leverage columns, inducing columns:
    np.random.choice); the rand(n_train, 3N) of the reference's CPU worker benchmark (iterative.py:175 -> predict.py:833-858) is
    spent only with GDMLTrain._emulate_ref_rng, which the replay tests of a freshly installed reference set."""
    tr, ctx = trainer
    fx = load('pcg_restart')
    M, N = 40, fx['R_all'].shape[1]
    drawn = []
    orig_rand = np.random.rand
    monkeypatch.setattr(np.random, 'rand', lambda *shape: (drawn.append(shape), orig_rand(*shape))[1])
    after = {}
    for emulate in (False, True):
        del drawn[:]
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = 6
        tr._emulate_ref_rng = emulate
        np.random.seed(3)
        tr.train(_task(fx, M))
        after[emulate] = (list(drawn), np.random.random())
    assert after[False][0] == []
    assert after[True][0] == [(M, 3 * N)]
    assert after[False][1] != after[True][1]  # the emulation moves the caller's stream, the default does not beyond the solver's own draws


def test_distributed_cadence_broadcasts_only_when_a_checkpoint_can_be_written(trainer, monkeypatch):
    """Sharded mode (a broadcast channel on the context): every random draw and every decision that gates a collective is
    rank 0's.  Without a checkpoint writer nothing is broadcast inside the CG loop; with one, rank 0's clock is broadcast in
    the iterations that can write a checkpoint (multiples of 10) only."""
    from sgdml_amd.solvers import iterative as it_mod

    tr, ctx = trainer
    fx = load('pcg_restart')
    M = 40
    sent = []
    ctx._bcast = lambda arr, src=0: (sent.append(np.asarray(arr).copy()), arr)[1]
    monkeypatch.setattr(it_mod, 'timeit', FakeClock(1.2))
    counts = {}
    for writer in (None, lambda m: None):
        del sent[:]
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = 6
        np.random.seed(3)
        model = tr.train(_task(fx, M), save_progr_callback=writer)
        iters = int(model['solver_iters'])
        floats = [a for a in sent if a.dtype == np.float64 and a.size == 1]
        counts[writer is None] = (iters, len(floats), len(sent))
    iters0, clock0, total0 = counts[True]
    iters1, clock1, total1 = counts[False]
    assert iters0 == iters1 and iters0 > 20
    assert clock0 == 0  # no writer: no per-iteration broadcast at all
    assert clock1 == (iters1 + 9) // 10  # a writer: the clock of iterations 0, 10, 20, ...
    assert total1 - clock1 == total0  # everything else (draws, solver choice, k, writer flag) is the same


def test_descriptors_are_reused_across_a_sigma_sweep(trainer):
    """sigma-grid reuse: GDMLTrain computes descriptors / Jacobians of a set of training geometries once --
    a second train() on the same geometries (another sigma, other labels) gets the same arrays back, read-only; geometries
    changed IN PLACE, a lattice, or another molecule size are different keys; models get writable copies."""
    from sgdml_amd.utils.desc import Desc

    tr, ctx = trainer
    n_calls = []
    real = ctx.desc_from_R
    ctx.desc_from_R = lambda R, n, lat=None: (n_calls.append(1), real(R, n, lat))[1]
    ds = orc.synth_dataset(5, 12, seed=3, jitter=0.2)
    R = ds['R'].reshape(12, -1).copy()
    desc = Desc(5)
    desc._ctx = ctx
    seen = []
    a = tr._train_descriptors(desc, R, None, lambda *args, **kw: seen.append(kw.get('sec_disp_str')))
    b = tr._train_descriptors(desc, R.copy(), None, lambda *args, **kw: seen.append(kw.get('sec_disp_str')))
    assert len(n_calls) == 1 and a[0] is b[0] and a[1] is b[1] and seen[-1] == 'reused'
    assert not a[0].flags.writeable and not a[1].flags.writeable
    xo, go = orc.desc_from_R(R)
    assert np.array_equal(a[0], xo) and np.array_equal(a[1], go)
    R[0, 0] += 0.25  # same buffer, new content
    c = tr._train_descriptors(desc, R, None, None)
    assert len(n_calls) == 2 and c[0] is not a[0] and not np.array_equal(c[0], a[0])
    lat = 20.0 * np.eye(3)
    d = tr._train_descriptors(desc, R, (lat, np.linalg.inv(lat)), None)
    assert len(n_calls) == 3 and d[0] is not c[0]
    # the model dictionary keeps a writable array (the reference's is writable: a view of a local)
    model = tr.create_model({'z': np.arange(5), 'R_train': R.reshape(12, 5, 3), 'F_train': np.zeros((12, 5, 3)), 'sig': 10,
                             'lam': 1e-10, 'use_E': False, 'use_E_cstr': False, 'use_sym': False, 'perms': np.arange(5)[None],
                             'dataset_name': np.array('x'), 'dataset_theory': np.array('y'), 'idxs_train': np.arange(12),
                             'md5_train': 'm', 'idxs_valid': np.arange(0), 'md5_valid': 'm', 'type': 't', 'code_version': '1.0.3'},
                            'analytic', d[0], d[1], np.arange(10), 1.0, np.zeros(12 * 15))
    assert model['R_desc'].flags.writeable and np.array_equal(model['R_desc'], d[0].T)


def test_bench_labels_follow_the_library_dispatch_of_the_perm2_kernel():
    """bench.py names the assembly kernel of a configuration in its `roofline_assemble` entries; the dispatch itself is C++
    (csrc/assemble_perm2.hip assemble_perm2_applicable: N <= 42, P >= asm.perm2_min_p, N >= asm.perm2_min_n, groups below 16
    elements four atoms later).  Both are read here so that one cannot move without the other."""
    import re
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'sgdml_amd', 'csrc', 'assemble_perm2.hip')).read()
    min_n = int(re.search(r'"asm\.perm2_min_n",\s*(\d+)\)', src).group(1))
    min_p = int(re.search(r'"asm\.perm2_min_p",\s*(\d+)\)', src).group(1))
    max_n = int(re.search(r'constexpr int P2_MAXN = (\d+);', src).group(1))
    assert 'ts.P >= 16 || ts.N >= min_n + 4' in src

    def lib_rule(n, p):
        if p < min_p or n < min_n or n > max_n:
            return False
        return p >= 16 or n >= min_n + 4

    sys.path.insert(0, root)
    import bench

    for n in range(25, 50):
        for p in (1, 2, 6, 12, 16, 27, 64):
            assert bench.assembly_kernel_name(n, p).startswith('assemble_perm2_kernel') == lib_rule(n, p), (n, p)
    # P = 1 beyond 21 atoms: the direct kernel of assemble_big1.hip (its rule is read from the source as well)
    big = open(os.path.join(root, 'sgdml_amd', 'csrc', 'assemble_big1.hip')).read()
    assert 'ts.P == 1 && ts.N >= 22 && ts.N <= 256' in big
    assert bench.assembly_kernel_name(21, 1).startswith('assemble_strip') and bench.assembly_kernel_name(100, 1).startswith('assemble_big1_kernel')
    assert bench.assembly_kernel_name(22, 1).startswith('assemble_big1') and bench.assembly_kernel_name(300, 1) == 'assemble_perm_kernel'
    assert bench.assembly_kernel_name(100, 2) == 'assemble_perm_kernel'


def test_bench_lines_of_every_gpu_count_name_the_same_workload():
    """bench.py --gpus N: `metric` and `config.workload` are ONE string per workload for every N (bench.line_skeleton), so that the
    driver's N = 1, 2, 4, 8 values form a curve (round-5 review: the N = 1 line printed configs[1] seconds, the N > 1 line a
    configs[2] step); the default workload is the same (analytic) at every N; only `parallelism` names the partitioning."""
    import inspect

    import bench

    for wl, kw in (('analytic', {}), ('cg', {'cg_iters': 50, 'cg_inducing': 200})):
        lines = [bench.line_skeleton(wl, 21, 1000 if wl == 'analytic' else 5000, w, **kw) for w in (1, 2, 4, 8)]
        assert len({ln['metric'] for ln in lines}) == 1 and len({ln['workload'] for ln in lines}) == 1
        assert lines[0]['parallelism'] == 'single GPU' and len({ln['parallelism'] for ln in lines}) == 4
    assert 'configs[1]' in bench.line_skeleton('analytic', 21, 1000, 8)['workload']
    src = inspect.getsource(bench.main)
    assert "workload = 'analytic' if args.workload == 'auto' else args.workload" in src  # one default for every N
    # both producers of a line take their strings from the skeleton and publish the curve point under the same key
    for fn in (bench.run_analytic, bench.run_multi):
        body = inspect.getsource(fn)
        assert 'line_skeleton(' in body and "out['scale_point']" in body


def test_cost_rule_for_the_number_of_inducing_points():
    """Iterative.cost_n_inducing_pts (round 6; GDMLTrain.inducing_pts_policy = 'cost'): never more than the memory rule allows,
    the memory rule itself for systems whose build is predicted under a second (the reference's behaviour, iterative.py:498-503),
    the minimiser of B k^2 + t_mv C / k otherwise; deterministic, monotone in the mat-vec cost, per-rank cost under sharding."""
    from sgdml_amd.solvers.iterative import Iterative as I

    assert I.cost_n_inducing_pts(300, 21, 1, 300) == 300        # small: exact preconditioner, as the reference would build
    assert I.cost_n_inducing_pts(10, 5, 4, 10) == 10
    k2 = I.cost_n_inducing_pts(5000, 21, 1, 144)                # configs[2]: measured optimum 40-80 (profiles/r06_k_sweep.txt)
    assert 50 <= k2 <= 90
    k3 = I.cost_n_inducing_pts(2000, 42, 27, 143)               # configs[3]: measured optimum 70-85, flat to ~110
    assert 70 <= k3 <= 120
    assert I.cost_n_inducing_pts(3000, 100, 1, 24) == 24        # configs[4] at 64 GB: memory-limited below the cost optimum
    assert I.cost_n_inducing_pts(2000, 42, 1, 350) < 60         # the 74 s case of profiles/r06_train_flow.txt
    assert I.cost_n_inducing_pts(2000, 42, 27, 350) > I.cost_n_inducing_pts(2000, 42, 1, 350)  # dearer mat-vec -> more points
    assert I.cost_n_inducing_pts(5000, 21, 1, 20) == 20         # never above the memory rule
    assert I.cost_n_inducing_pts(5000, 21, 1, 144, world=2) == k2  # build and mat-vec both shard: same optimum


def test_chunked_host_helpers_keep_their_bits():
    """Round 6: the distance matrices of the symmetry search and J v of create_model run in chunks of geometries (cache-sized
    temporaries); same bits as the one-shot NumPy expressions they replace, with and without a lattice, ragged last chunk."""
    from sgdml_amd.utils import perm
    from sgdml_amd.utils.desc import Desc

    rs = np.random.RandomState(5)
    R = rs.normal(size=(70, 9, 3)) * 3
    lat = np.diag([7.0, 8.0, 9.0]) + 0.3 * rs.normal(size=(3, 3))
    for li in (None, (lat, np.linalg.inv(lat))):
        diff = R[:, :, None, :] - R[:, None, :, :]
        if li is not None:
            diff = diff - np.einsum('ij,mabj->mabi', li[0], np.rint(np.einsum('ij,mabj->mabi', li[1], diff)))
        assert np.array_equal(perm._dist_matrices(R, li), np.sqrt((diff**2).sum(-1)))
        assert np.array_equal(perm._dist_matrices(R, li, chunk=7), perm._dist_matrices(R, li, chunk=1000))
    d = Desc(9)
    gd, a = rs.normal(size=(130, 36, 3)), rs.normal(size=(130, 27))
    i, j = d.tril_indices
    v = a.reshape(130, -1, 3)
    assert np.array_equal(d.d_desc_dot_vec(gd, a), np.sum(gd * (v[:, j, :] - v[:, i, :]), axis=-1))
    assert d.d_desc_dot_vec(gd[0], a[0]).shape == (1, 36)
    assert np.array_equal(d.d_desc_dot_vec(gd[:1], a), np.sum(gd[:1] * (v[:, j, :] - v[:, i, :]), axis=-1))  # broadcast form
