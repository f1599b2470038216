"""GPU parity tests: the HIP path (through the C ABI) against the reference-generated golden
fixtures and against the NumPy oracle on seeded inputs.  Tolerances (fp64, stated per test):
  K        : max|dK| <= 1e-12 * max|K|                       
  solve    : ||(-K + lam I)(-alpha) - y|| / ||y|| <= 1e-10   (cond ~ 1/lam, no elementwise alpha parity)
  predict  : |dF| <= 1e-10 max|F| + cancellation floor       (see test_oracle_golden.cancel_floor)
"""
import numpy as np
import pytest

from oracle import gdml_oracle as orc
from tests._tol import solve_tol
from tests.test_oracle_golden import _lat, _model, _tril_perms, cancel_floor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from sgdml_amd import _lib

    c = _lib.Context()
    yield c
    c.close()


@pytest.fixture
def ctx_factory():
    from sgdml_amd import _lib

    made = []

    def make():
        c = _lib.Context()
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()


def test_desc(golden, ctx):
    g = golden
    M, N = g['R_train'].shape[:2]
    xd, gd = ctx.desc_from_R(g['R_train'].reshape(M, -1), N, _lat(g))
    np.testing.assert_allclose(xd, g['R_desc'], rtol=1e-13, atol=0)
    np.testing.assert_allclose(gd, g['R_d_desc'], rtol=1e-12, atol=1e-15)


def test_K_full(golden, ctx):
    g = golden
    ctx.train_upload(g['R_desc'], g['R_d_desc'], _tril_perms(g))
    K = ctx.assemble_K(float(g['sig']), bool(g['use_E_cstr']), to_host=True)
    assert K.shape == g['K'].shape
    assert np.abs(K - g['K']).max() <= 1e-12 * np.abs(g['K']).max()
    # device-resident copy is the same matrix
    assert np.array_equal(ctx.K_to_host(), K)


def test_K_columns(golden, ctx):
    g = golden
    n = g['K'].shape[0]
    ex = len(g['col_idxs'])
    ctx.train_upload(g['R_desc'], g['R_d_desc'], _tril_perms(g))
    Kc = ctx.assemble_K(float(g['sig']), bool(g['use_E_cstr']), idx=g['col_idxs'], alloc_extra_rows=ex, to_host=True)
    assert Kc.shape == (n + ex, ex)
    scale = np.abs(g['K']).max()
    assert np.abs(Kc[:n] - g['K_cols']).max() <= 1e-12 * scale
    N3 = 3 * g['R_train'].shape[1]
    pts = g['K_slice'].shape[1] // N3
    Ks = ctx.assemble_K(float(g['sig']), bool(g['use_E_cstr']), points=(0, pts), to_host=True)
    assert np.abs(Ks - g['K_slice']).max() <= 1e-12 * scale


def test_K_bad_columns(golden, ctx):
    g = golden
    ctx.train_upload(g['R_desc'], g['R_d_desc'], _tril_perms(g))
    with pytest.raises(ValueError):
        ctx.assemble_K(float(g['sig']), False, idx=np.array([3, 2]))  # unsorted (train.py:1345)
    with pytest.raises(ValueError):
        ctx.assemble_K(float(g['sig']), False, idx=np.array([0, 10**9]))


def test_analytic_solve(golden, ctx):
    g = golden
    lam = float(g['lam'])
    ctx.train_upload(g['R_desc'], g['R_d_desc'], _tril_perms(g))
    ctx.assemble_K(float(g['sig']), bool(g['use_E_cstr']))
    assert ctx.chol_factor(lam) == 0
    alphas = ctx.chol_solve(g['y'])
    A = -g['K'] + lam * np.eye(g['K'].shape[0])
    r = A @ (-alphas) - g['y']
    assert np.linalg.norm(r) / np.linalg.norm(g['y']) < solve_tol(A, alphas, g['y'])  # 1e-10; tests/_tol.py for n4_p6_pbc
    # factor parity with LAPACK on the lower triangle
    import scipy.linalg as sla

    L = np.tril(ctx.K_to_host())
    Lref = sla.cholesky(A, lower=True)
    assert np.abs(L @ L.T - A).max() <= 1e-13 * np.abs(A).max()
    assert np.abs(L - Lref).max() <= 1e-6 * np.abs(Lref).max()


def test_predict(golden, ctx):
    g = golden
    m = _model(g)
    tp = _tril_perms(g)
    R_desc_train = np.ascontiguousarray(m['R_desc'].T)
    ctx.predict_upload_model(R_desc_train, m['R_d_desc_alpha'], tp, float(g['sig']), m.get('alphas_E'))
    fl = cancel_floor(g)
    E, F = ctx.predict(g['R_test'].reshape(len(g['R_test']), -1), _lat(g))
    E = E * m['std'] + m['c']
    F = F * m['std']
    assert np.abs(F - g['F_test']).max() <= 1e-10 * np.abs(g['F_test']).max() + fl
    assert np.abs(E - g['E_test']).max() <= 1e-10 * max(1.0, np.abs(g['E_test']).max()) + fl * float(g['sig'])
    # training-set mode
    ctx.train_upload(g['R_desc'], g['R_d_desc'], tp)
    E, F = ctx.predict(None)
    E = E * m['std'] + m['c']
    F = F * m['std']
    assert np.abs(F - g['F_train_pred']).max() <= 1e-10 * np.abs(g['F_train_pred']).max() + fl
    assert np.abs(E - g['E_train_pred']).max() <= 1e-10 * max(1.0, np.abs(g['E_train_pred']).max()) + fl * float(g['sig'])
    # force-only call returns no energies
    E2, F2 = ctx.predict(g['R_test'].reshape(len(g['R_test']), -1), _lat(g), return_E=False)
    assert E2 is None and np.array_equal(F2 * m['std'], ctx.predict(g['R_test'].reshape(len(g['R_test']), -1), _lat(g))[1] * m['std'])


def test_kernel_matvec(golden, ctx):
    g = golden
    tp = _tril_perms(g)
    M = g['R_desc'].shape[0]
    ctx.train_upload(g['R_desc'], g['R_d_desc'], tp)
    ctx.predict_upload_model(g['R_desc'], np.zeros_like(g['R_desc']), tp, float(g['sig']),
                             np.zeros(M) if bool(g['use_E_cstr']) else None)
    Kv = ctx.kernel_matvec(float(g['lam']), bool(g['use_E_cstr']), g['v'])
    assert np.abs(Kv - g['Kv']).max() <= 1e-11 * np.abs(g['Kv']).max()


def test_dropin_train_predict(golden):
    """End to end through the reference's public API: task -> GDMLTrain.train -> GDMLPredict."""
    from sgdml_amd.predict import GDMLPredict
    from sgdml_amd.train import GDMLTrain

    g = golden
    M, N = g['R_train'].shape[:2]
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': np.ones(N, dtype=int) * 6, 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(M, M + 7), 'md5_valid': 'x',
        'sig': int(g['sig']), 'lam': float(g['lam']), 'use_E': True, 'use_E_cstr': bool(g['use_E_cstr']),
        'use_sym': g['perms'].shape[0] > 1, 'perms': g['perms'],
    }
    if 'lattice' in g:
        task['lattice'] = g['lattice']
    trainer = GDMLTrain()
    try:
        model = trainer.train(task)
    finally:
        trainer.__del__()
    assert model['solver_name'] == 'analytic' and model['type'] == 'm'
    assert np.array_equal(model['tril_perms_lin'], g['tril_perms_lin'])
    np.testing.assert_allclose(model['std'], float(g['y_std']), rtol=1e-14)
    pred = GDMLPredict(model)
    E, F = pred.predict(g['R_test'].reshape(len(g['R_test']), -1))
    # alpha is conditioning-limited (lam = 1e-10): compare predictions loosely with the reference's
    assert np.abs(F - g['F_test']).max() <= 2e-4 * np.abs(g['F_test']).max()
    assert np.abs(E - g['E_test']).max() <= 2e-4 * max(1.0, np.abs(g['E_test']).max())
    # training-set mode agrees with what the reference's model gives on its training set
    pred.set_R_desc(g['R_desc'])
    pred.set_R_d_desc(g['R_d_desc'])
    E_tr, F_tr = pred.predict()
    assert np.abs(F_tr - g['F_train_pred']).max() <= 2e-4 * np.abs(g['F_train_pred']).max()
    # integration constant (train.py:1258) within the same conditioning-limited band
    assert abs(model['c'] - float(g['model_c'])) <= 2e-4 * max(1.0, abs(float(g['model_c'])))


@pytest.mark.parametrize('N,M,P', [(21, 24, 1), (12, 30, 2), (33, 6, 1)])
def test_K_and_predict_vs_oracle_seeded(ctx, N, M, P):
    ds = orc.synth_dataset(N, M + 5, seed=N * 100 + M, jitter=0.2)
    R = ds['R'].reshape(M + 5, -1)
    perms = [list(range(N))]
    if P == 2:
        p2 = list(range(N))
        p2[0], p2[1] = 1, 0
        perms.append(p2)
    tp = orc.tril_perms_from_atom_perms(np.array(perms))
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    xd, gd = ctx.desc_from_R(R[:M], N)
    xo, go = orc.desc_from_R(R[:M])
    np.testing.assert_allclose(xd, xo, rtol=1e-13)
    sig = 25.0
    ctx.train_upload(xo, go, tp)
    K = ctx.assemble_K(sig, False, to_host=True)
    Ko = orc.assemble_K(xo, go, lin, sig)
    assert np.abs(K - Ko).max() <= 1e-12 * np.abs(Ko).max()
    assert np.abs(K - K.T).max() <= 1e-13 * np.abs(K).max()
    rs = np.random.RandomState(5)
    v = rs.normal(size=K.shape[0])
    JA = orc.d_desc_dot_vec(go, v.reshape(M, -1))
    ctx.predict_upload_model(xo, JA, tp, sig, None)
    E, F = ctx.predict(R[M:])
    xq, gq = orc.desc_from_R(R[M:])
    Eo, Fo = orc.predict_from_desc(xq, gq, xo, JA, tp, sig)
    assert np.abs(F - Fo).max() <= 1e-11 * np.abs(Fo).max()
    assert np.abs(E - Eo).max() <= 1e-11 * np.abs(Eo).max()


@pytest.mark.parametrize('n', [700, 1537])
def test_cholesky_multi_panel(ctx, n):
    """Factor sizes that span several 512-wide panels and a ragged last block."""
    import ctypes as C

    import scipy.linalg as sla
    from sgdml_amd import _lib

    rs = np.random.RandomState(n)
    B = rs.normal(size=(n, n + 20))
    A = B @ B.T / n + 0.5 * np.eye(n)
    # drive the device factorisation directly on an uploaded matrix via a K-shaped problem:
    # use N=2 atoms (3N=6) is not flexible enough -> call the raw entry through a fake training set
    # is overkill; instead check through gdml_chol_factor by loading A as "-K": assemble a dummy K of
    # the right size and overwrite it.
    M = (n + 5) // 6
    ds = orc.synth_dataset(2, M, seed=1)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = np.zeros((1, 1), dtype=np.int64)
    ctx.train_upload(xo, go, tp)
    ctx.assemble_K(10.0, False)
    rows, cols, _ = ctx.K_shape()
    n2 = rows
    B = rs.normal(size=(n2, n2 + 20))
    A = B @ B.T / n2 + 0.5 * np.eye(n2)
    p, ld = C.c_void_p(), C.c_int64()
    ctx._check(ctx._lib.gdml_K_dev(ctx._h, C.byref(p), C.byref(ld)))
    buf = np.zeros((n2, ld.value))
    buf[:, :n2] = -A
    ctx._check(ctx._lib.gdml_memcpy_h2d(ctx._h, p, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    assert ctx.chol_factor(0.0) == 0
    L = np.tril(ctx.K_to_host())
    Lref = sla.cholesky(A, lower=True)
    assert np.abs(L - Lref).max() <= 1e-11 * np.abs(Lref).max()
    y = rs.normal(size=n2)
    x = -ctx.chol_solve(y)
    assert np.linalg.norm(A @ x - y) <= 1e-11 * np.linalg.norm(y)


@pytest.mark.parametrize('opts', [
    {'chol.outer': 1024, 'chol.outer_min_rows': 512, 'chol.fused_min_rows': 256},   # panel pairs (K = 1024 bulk in two halves)
    {'chol.outer': 512, 'chol.fused_min_rows': 256},                                # one-level fused schedule
    {'chol.fused_diag': 0},                                                         # look-ahead schedule on two streams
    {'chol.small_update': 0},                                                       # rank-64 chain updates through the GEMM tile kernel
])
def test_cholesky_schedules_small(ctx_factory, opts):
    """The schedules that only engage on large matrices by default (panel pairs with a K = 1024 trailing update split in two halves, each hiding one
    diagonal block; the one-level fused schedule), forced on a 3.3k matrix, and the two-stream
    fallback: same factor as LAPACK, the right-hand side row rides through."""
    import ctypes as C

    import scipy.linalg as sla
    c = ctx_factory()
    for k, v in opts.items():
        c.set_option(k, v)
    M = 556  # 2 atoms: 3N = 6 -> n = 3336 = 6.5 inner panels, ragged last block
    ds = orc.synth_dataset(2, M, seed=2)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    c.train_upload(xo, go, np.zeros((1, 1), dtype=np.int64))
    c.assemble_K(10.0, False, alloc_extra_rows=1)
    n = c.K_shape()[0]
    rs = np.random.RandomState(7)
    B = rs.normal(size=(n, n + 30))
    A = B @ B.T / n + 0.5 * np.eye(n)
    p, ld = C.c_void_p(), C.c_int64()
    c._check(c._lib.gdml_K_dev(c._h, C.byref(p), C.byref(ld)))
    buf = np.zeros((n + 1, ld.value))
    buf[:n, :n] = -A
    c._check(c._lib.gdml_memcpy_h2d(c._h, p, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    y = rs.normal(size=n)
    c.chol_set_rhs(y)
    assert c.chol_factor(0.0) == 0
    L = np.tril(c.K_to_host()[:n])
    Lref = sla.cholesky(A, lower=True)
    assert np.abs(L - Lref).max() <= 1e-11 * np.abs(Lref).max()
    x = -c.chol_solve(None)
    assert np.linalg.norm(A @ x - y) <= 1e-11 * np.linalg.norm(y)


def test_cholesky_not_pd_reports_lapack_info(ctx):
    import ctypes as C

    M = 40
    ds = orc.synth_dataset(2, M, seed=1)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    ctx.train_upload(xo, go, np.zeros((1, 1), dtype=np.int64))
    ctx.assemble_K(10.0, False)
    n = ctx.K_shape()[0]
    A = np.eye(n)
    A[100, 100] = -1.0
    p, ld = C.c_void_p(), C.c_int64()
    ctx._check(ctx._lib.gdml_K_dev(ctx._h, C.byref(p), C.byref(ld)))
    buf = np.zeros((n, ld.value))
    buf[:, :n] = -A
    ctx._check(ctx._lib.gdml_memcpy_h2d(ctx._h, p, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    with pytest.raises(np.linalg.LinAlgError, match='101-th leading minor'):
        ctx.chol_factor(0.0)


# ------------------------------------------------------------------ iterative solver path


def test_nystroem_factor_and_precon(golden, ctx):
    g = golden
    lam = float(g['lam'])
    n = g['K'].shape[0]
    idx = g['col_idxs']
    m = len(idx)
    ctx.train_upload(g['R_desc'], g['R_d_desc'], _tril_perms(g))
    ctx.assemble_K(float(g['sig']), bool(g['use_E_cstr']), idx=idx, alloc_extra_rows=m)
    lev, fac, info = ctx.nystroem_factor(lam, idx, want_factor=True)
    assert info == 0 and fac.shape == (m, n)
    # the factor is conditioning-sensitive; the preconditioner action L^T L is what is compared
    P1 = fac.T @ fac
    P2 = g['L_inv_K_mn'].T @ g['L_inv_K_mn']
    assert np.abs(P1 - P2).max() <= 1e-6 * np.abs(P2).max()
    np.testing.assert_allclose(lev, np.einsum('ij,ij->j', fac, fac), rtol=1e-10)
    v = g['v']
    out = ctx.precon_apply(lam, v)
    ref = orc.precon_apply(fac, lam, v)
    assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max()


def test_pcg_solves_system(golden, ctx):
    g = golden
    lam = float(g['lam'])
    use_E = bool(g['use_E_cstr'])
    tp = _tril_perms(g)
    M = g['R_desc'].shape[0]
    idx = g['col_idxs']
    ctx.train_upload(g['R_desc'], g['R_d_desc'], tp)
    ctx.assemble_K(float(g['sig']), use_E, idx=idx, alloc_extra_rows=len(idx))
    ctx.nystroem_factor(lam, idx)
    ctx.predict_upload_model(g['R_desc'], np.zeros_like(g['R_desc']), tp, float(g['sig']),
                             np.zeros(M) if use_E else None)
    seen = []
    x, info, iters, resid = ctx.pcg(lam, use_E, g['y'], rtol=1e-6, maxiter=5000,
                                    callback=lambda it, r, fetch_x: seen.append((it, r)) or False)
    assert info == 0 and iters == len(seen)
    A = -g['K'] + lam * np.eye(g['K'].shape[0])
    assert np.linalg.norm(A @ x - g['y']) <= 2e-6 * np.linalg.norm(g['y'])
    # warm start from the solution converges immediately; a stopping callback reports info = 2
    x2, info2, iters2, _ = ctx.pcg(lam, use_E, g['y'], x0=x, rtol=1e-5, maxiter=50)
    assert info2 == 0 and iters2 <= 1
    x3, info3, iters3, _ = ctx.pcg(lam, use_E, g['y'], rtol=1e-12, maxiter=50, callback=lambda it, r, fetch_x: it >= 2)
    assert info3 == 2 and iters3 == 2
    # The loop is pipelined (iterations are queued `pcg.depth` ahead of the host's tests) -- with scipy's semantics all the
    # same: every depth returns the same iterate after the same number of iterations with the same residual history, the
    # iterate fetched inside callback k is x_k (its true residual is the reported one), and a stop at k returns x_k.
    runs = {}
    for depth in (0, 1, 2, 5):
        ctx.set_option('pcg.depth', depth)
        hist, snaps = [], {}

        def cb(it, r, fetch_x):
            hist.append(r)
            if it in (1, 3):
                snaps[it] = fetch_x()
            return False

        xd_, info_d, iters_d, resid_d = ctx.pcg(lam, use_E, g['y'], rtol=1e-6, maxiter=5000, callback=cb)
        runs[depth] = (xd_, info_d, iters_d, resid_d, np.array(hist), snaps)
        xs_, info_s, iters_s, _ = ctx.pcg(lam, use_E, g['y'], rtol=1e-12, maxiter=50, callback=lambda it, r, f: it >= 3)
        assert info_s == 2 and iters_s == 3 and np.array_equal(xs_, snaps[3])
        x1_, info_1, iters_1, _ = ctx.pcg(lam, use_E, g['y'], rtol=1e-12, maxiter=1)
        assert info_1 == 1 and iters_1 == 1 and np.array_equal(x1_, snaps[1])
    ctx.set_option('pcg.depth', 2)
    ref_run = runs[0]
    assert np.array_equal(ref_run[0], x) and ref_run[2] == iters
    for depth in (1, 2, 5):
        r_ = runs[depth]
        assert np.array_equal(r_[0], ref_run[0]) and r_[1:4] == ref_run[1:4] and np.array_equal(r_[4], ref_run[4])
    for k_, xk_ in ref_run[5].items():
        true_res = np.linalg.norm(A @ xk_ - g['y'])
        assert abs(true_res - ref_run[4][k_ - 1]) <= 1e-6 * np.linalg.norm(g['y'])


def test_dropin_train_iterative(golden):
    """GDMLTrain.train forced onto the CG branch (train.py:986-1050) reaches the solver tolerance."""
    from sgdml_amd.predict import GDMLPredict
    from sgdml_amd.train import GDMLTrain

    g = golden
    if g['name'] == 'n4_p6_pbc':
        pytest.skip('rank-deficient symmetrised kernel: CG stagnates in the reference as well')
    M, N = g['R_train'].shape[:2]
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': np.ones(N, dtype=int) * 6, 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(M, M + 7), 'md5_valid': 'x',
        'sig': int(g['sig']), 'lam': float(g['lam']), 'use_E': True, 'use_E_cstr': bool(g['use_E_cstr']),
        'use_sym': g['perms'].shape[0] > 1, 'perms': g['perms'],
    }
    if 'lattice' in g:
        task['lattice'] = g['lattice']
    np.random.seed(0)
    trainer = GDMLTrain()
    trainer._force_solver = 'cg'
    try:
        model = trainer.train(task)
    finally:
        trainer.__del__()
    assert model['solver_name'] == 'cg'
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train'] * 1.01
    n = g['K'].shape[0]
    alphas = np.hstack((model['alphas_F'], model['alphas_E'])) if bool(g['use_E_cstr']) else model['alphas_F']
    A = -g['K'] + float(g['lam']) * np.eye(n)
    assert np.linalg.norm(A @ (-alphas) - g['y']) <= 2e-4 * np.linalg.norm(g['y'])
    pred = GDMLPredict(model)
    E, F = pred.predict(g['R_test'].reshape(len(g['R_test']), -1))
    assert np.abs(F - g['F_test']).max() <= 0.05 * np.abs(g['F_test']).max()


# ------------------------------------------------------------------ sharding / RCCL plumbing


def test_comm_world1_pcg(ctx_factory):
    """A real RCCL communicator with one rank: every collective is issued (in-place all-gather /
    all-reduce of a single chunk) and the solve must be unchanged."""
    import os

    g = dict(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'n6_p1.npz')))
    lam, sig = float(g['lam']), float(g['sig'])
    tp = _tril_perms(g)
    idx = g['col_idxs']
    res = []
    for with_comm in (False, True):
        c = ctx_factory()
        if with_comm:
            c.comm_init(c.comm_unique_id(), 0, 1)
            assert c.comm_info() == (0, 1)
        c.train_upload(g['R_desc'], g['R_d_desc'], tp)
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        lev, _, _ = c.nystroem_factor(lam, idx)
        c.predict_upload_model(g['R_desc'], np.zeros_like(g['R_desc']), tp, sig, None)
        x, info, iters, resid = c.pcg(lam, False, g['y'], rtol=1e-6, maxiter=3000)
        assert info == 0
        res.append((lev, x, iters))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-12)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-9, atol=1e-9 * np.abs(res[0][1]).max())


@pytest.mark.parametrize('case,world', [('n6_p1', 2), ('n9_p1', 4), ('n5_p4', 3)])
def test_virtual_rank_shards_stitch(ctx_factory, case, world):
    """Shard arithmetic of the kernels on one GPU: 'virtual ranks' (no communicator) each produce
    their row shard of the Nystroem matrix and of K v; stitched together they equal the unsharded
    results."""
    import os

    from sgdml_amd.dist import shard_range

    g = dict(np.load(os.path.join(os.path.dirname(__file__), 'golden', case + '.npz')))
    lam, sig = float(g['lam']), float(g['sig'])
    tp = _tril_perms(g)
    idx = g['col_idxs']
    M, N = g['R_train'].shape[:2]
    N3 = 3 * N
    n = M * N3
    ref = ctx_factory()
    ref.train_upload(g['R_desc'], g['R_d_desc'], tp)
    ref.predict_upload_model(g['R_desc'], np.zeros_like(g['R_desc']), tp, sig, None)
    Kv_full = ref.kernel_matvec(lam, False, g['v'])
    Kc_full = ref.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx), to_host=True)[:n]
    Kv_st = np.full(n, np.nan)
    Kc_st = np.full_like(Kc_full, np.nan)
    for r in range(world):
        p0, p1, per = shard_range(r, world, M)
        c = ctx_factory()
        c.comm_init(None, r, world)
        c.train_upload(g['R_desc'], g['R_d_desc'], tp)
        c.predict_upload_model(g['R_desc'], np.zeros_like(g['R_desc']), tp, sig, None)
        out = c.kernel_matvec(lam, False, g['v'])
        Kv_st[p0 * N3:p1 * N3] = out[p0 * N3:p1 * N3]
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        rows, cols, extra = c.K_shape()
        assert rows == (p1 - p0) * N3 and cols == len(idx) and extra == len(idx)
        Kc_st[p0 * N3:p1 * N3] = c.K_to_host()[:rows]
    assert not np.isnan(Kv_st).any() and not np.isnan(Kc_st).any()
    np.testing.assert_allclose(Kv_st, Kv_full, rtol=0, atol=1e-13 * np.abs(Kv_full).max())
    np.testing.assert_allclose(Kc_st, Kc_full, rtol=0, atol=1e-14 * np.abs(Kc_full).max())


def test_K_large_molecule_global_table(ctx):
    """N = 60 (C60-sized, BASELINE configs[4]): the column point's dense table no longer fits in LDS
    next to the row point's; the kernel then reads G_j from the global dense table."""
    N, M = 60, 3
    ds = orc.synth_dataset(N, M, seed=11, jitter=0.1)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    ctx.train_upload(xo, go, tp)
    K = ctx.assemble_K(40.0, False, to_host=True)
    Ko = orc.assemble_K(xo, go, orc.tril_perms_lin_from_tril_perms(tp), 40.0)
    assert np.abs(K - Ko).max() <= 1e-12 * np.abs(Ko).max()


def test_K_general_kernel_on_small_fixtures(golden, ctx):
    """The general kernel (assemble_perm.hip) forced on the small reference fixtures (permutations, E-constraint rows)."""
    g = golden
    ctx.set_option('asm.wave', 0)
    ctx.set_option('asm.pts', 0)
    ctx.train_upload(g['R_desc'], g['R_d_desc'], _tril_perms(g))
    K = ctx.assemble_K(float(g['sig']), bool(g['use_E_cstr']), to_host=True)
    assert np.abs(K - g['K']).max() <= 1e-12 * np.abs(g['K']).max()


def test_device_error_sums_match_reference_definitions(golden):
    """gdml_predict_errors vs the reference's _online_err recipe (cli.py:1170, :1572-1605) evaluated
    with NumPy on the reference's own predictions."""
    from sgdml_amd.predict import GDMLPredict

    g = golden
    N = g['R_train'].shape[1]
    m = _model(g)
    model = {'type': 'm', 'z': np.ones(N, dtype=int), 'R_desc': m['R_desc'], 'R_d_desc_alpha': m['R_d_desc_alpha'],
             'sig': float(g['sig']), 'c': m['c'], 'std': m['std'], 'perms': g['perms'],
             'tril_perms_lin': g['tril_perms_lin']}
    if 'alphas_E' in m:
        model['alphas_E'] = m['alphas_E']
    if 'lattice' in g:
        model['lattice'] = g['lattice']
    rs = np.random.RandomState(3)
    B = len(g['R_test'])
    F_ref = g['F_test'] + 0.05 * rs.normal(size=g['F_test'].shape)
    E_ref = g['E_test'] + 0.1 * rs.normal(size=B)
    pred = GDMLPredict(model)
    got = pred.test_errors(g['R_test'].reshape(B, -1), F_ref, E_ref)
    f_pred, e_pred = g['F_test'], g['E_test']
    err = np.abs(F_ref - f_pred)
    fl = cancel_floor(g)
    tol = 1e-9 + 10 * fl
    assert abs(got['force'][0] - err.mean()) <= tol and abs(got['force'][1] - np.sqrt((err**2).mean())) <= tol
    e = np.abs(E_ref - e_pred)
    assert abs(got['energy'][0] - e.mean()) <= tol * float(g['sig']) and abs(got['energy'][1] - np.sqrt((e**2).mean())) <= tol * float(g['sig'])
    mp = np.linalg.norm(f_pred.reshape(-1, 3), axis=1)
    mr = np.linalg.norm(F_ref.reshape(-1, 3), axis=1)
    assert abs(got['magnitude'][0] - np.abs(mp - mr).mean()) <= tol
    cos = np.arccos(np.clip(np.einsum('ij,ij->i', f_pred.reshape(-1, 3) / mp[:, None], F_ref.reshape(-1, 3) / mr[:, None]), -1, 1)) / np.pi
    assert abs(got['angle'][0] - cos.mean()) <= 1e-7 + 100 * fl and abs(got['angle'][1] - np.sqrt((cos**2).mean())) <= 1e-7 + 100 * fl


# ------------------------------------------------------------------------------------------
# Large query batches (B >= 256) take the fp64-MFMA kernel; every tile-count instantiation
# (D <= 32, 64, ... 256), odd table sizes, permutations and the energy-constraint terms are checked
# against the oracle.  Tolerance: 1e-11 * max|F| (well-conditioned random coefficients).
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n_atoms,n_train,n_query,n_perms,with_aE', [
    (6, 37, 300, 1, False),    # D = 15  (odd D, odd M: table ends on an odd element)
    (9, 41, 257, 2, True),     # D = 36, permutations + energy constraints
    (12, 70, 300, 1, False),   # D = 66
    (14, 33, 256, 1, True),    # D = 91
    (16, 50, 320, 1, False),   # D = 120
    (18, 35, 300, 1, False),   # D = 153
    (20, 40, 290, 1, True),    # D = 190
    (21, 333, 1000, 1, False), # D = 210, many workgroups, ragged last split / last query tile
    (23, 47, 300, 1, False),   # D = 253
])
def test_predict_large_batch_mfma(ctx_factory, n_atoms, n_train, n_query, n_perms, with_aE):
    ds = orc.synth_dataset(n_atoms, n_train + n_query, seed=n_atoms)
    Rf = ds['R'].reshape(n_train + n_query, -1)
    perms = np.arange(n_atoms)[None]
    if n_perms == 2:
        p2 = np.arange(n_atoms)
        p2[[0, 1]] = p2[[1, 0]]
        perms = np.vstack([perms, p2])
    tp = orc.tril_perms_from_atom_perms(perms)
    xd, gd = orc.desc_from_R(Rf[:n_train])
    rs = np.random.RandomState(5)
    ja = rs.normal(size=xd.shape)
    aE = rs.normal(size=n_train) if with_aE else None
    sig = 12.0
    c = ctx_factory()
    c.predict_upload_model(xd, ja, tp, sig, aE)
    E, F = c.predict(Rf[n_train:])
    rq, rdq = orc.desc_from_R(Rf[n_train:])
    E0, F0 = orc.predict_from_desc(rq, rdq, xd, ja, tp, sig, aE)
    assert np.abs(F - F0).max() <= 1e-11 * np.abs(F0).max()
    assert np.abs(E - E0).max() <= 1e-11 * np.abs(E0).max()


@pytest.mark.parametrize('use_E_cstr', [False, True])
def test_kernel_matvec_large_mfma(ctx_factory, use_E_cstr):
    """Training-set mode of the same kernel (coincident points: |d|^2 from the expansion is clamped)."""
    n_atoms, M = 7, 300
    ds = orc.synth_dataset(n_atoms, M, seed=11)
    Rf = ds['R'].reshape(M, -1)
    tp = orc.tril_perms_from_atom_perms(np.arange(n_atoms)[None])
    xd, gd = orc.desc_from_R(Rf)
    sig, lam = 10.0, 1e-10
    c = ctx_factory()
    c.train_upload(xd, gd, tp)
    c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, np.zeros(M) if use_E_cstr else None)
    rs = np.random.RandomState(2)
    v = rs.normal(size=M * 3 * n_atoms + (M if use_E_cstr else 0))
    Kv = c.kernel_matvec(lam, use_E_cstr, v)
    Kv0 = orc.kernel_matvec(xd, gd, tp, sig, lam, v, use_E_cstr)
    assert np.abs(Kv - Kv0).max() <= 1e-11 * np.abs(Kv0).max()


def test_ase_calculator_with_stub_ase(golden, tmp_path, monkeypatch):
    """SGDMLCalculator (reference intf/ase_calc.py) through a minimal stand-in for ASE: unit handling and
    the results dictionary, single-geometry path."""
    import importlib
    import sys
    import types

    g = golden
    if 'lattice' in g:
        pytest.skip('calculator test uses the non-periodic fixtures')
    ase = types.ModuleType('ase')
    calcs = types.ModuleType('ase.calculators')
    calc = types.ModuleType('ase.calculators.calculator')
    units = types.ModuleType('ase.units')

    class Calculator:
        def __init__(self, *a, **k):
            self.results = {}

        def calculate(self, atoms=None, *a, **k):
            self.atoms = atoms

    calc.Calculator = Calculator
    units.kcal, units.mol = 2.611447418269555e22, 6.022140857e23
    for name, mod in (('ase', ase), ('ase.calculators', calcs), ('ase.calculators.calculator', calc), ('ase.units', units)):
        monkeypatch.setitem(sys.modules, name, mod)
    sys.modules.pop('sgdml_amd.intf.ase_calc', None)
    mod = importlib.import_module('sgdml_amd.intf.ase_calc')

    m = _model(g)
    m.update(type='m', z=np.ones(g['R_train'].shape[1], dtype=int) * 6, perms=g['perms'])
    path = tmp_path / 'model.npz'
    np.savez(path, **m)

    class Atoms:
        def __init__(self, pos):
            self._p = pos

        def get_positions(self):
            return self._p

    c = mod.SGDMLCalculator(str(path))
    try:
        pos = g['R_test'][0].reshape(-1, 3)
        c.calculate(Atoms(pos))
        k = units.kcal / units.mol
        fl = cancel_floor(g)
        assert abs(c.results['energy'] - g['E_test'][0] * k) <= (1e-10 * max(1.0, abs(g['E_test'][0])) + fl * float(g['sig'])) * k
        assert np.abs(c.results['forces'] - g['F_test'][0].reshape(-1, 3) * k).max() <= (1e-10 * np.abs(g['F_test']).max() + fl) * k
    finally:
        del c.gdml_predict
    sys.modules.pop('sgdml_amd.intf.ase_calc', None)


@pytest.mark.parametrize('n_atoms,M', [(5, 40), (7, 110)])
def test_chol_rhs_row_matches_separate_solve(ctx_factory, n_atoms, M):
    """gdml_chol_set_rhs: the right-hand side carried through the factorisation as an extra row gives the
    same solution as the separate forward substitution (both to the solve tolerance 1e-10, and to each
    other within the conditioning of the system); spans several panels (GDML_CHOL_NB default 512: n = 2310)."""
    ds = orc.synth_dataset(n_atoms, M, seed=3)
    Rf = ds['R'].reshape(M, -1)
    tp = orc.tril_perms_from_atom_perms(np.arange(n_atoms)[None])
    xd, gd = orc.desc_from_R(Rf)
    sig, lam = 10.0, 1e-8
    y = ds['F'].reshape(-1) / np.std(ds['F'])
    c = ctx_factory()
    c.train_upload(xd, gd, tp)
    c.assemble_K(sig, False, alloc_extra_rows=1)
    c.chol_set_rhs(y)
    assert c.chol_factor(lam) == 0
    a_fused = c.chol_solve(None)
    a_sep = c.chol_solve(y)
    K = orc.assemble_K(xd, gd, orc.tril_perms_lin_from_tril_perms(tp), sig)
    A = -K + lam * np.eye(K.shape[0])
    for a in (a_fused, a_sep):
        assert np.linalg.norm(A @ (-a) - y) / np.linalg.norm(y) <= 1e-10
    assert np.linalg.norm(a_fused - a_sep) <= 1e-6 * np.linalg.norm(a_sep)
    # state errors
    c2 = ctx_factory()
    c2.train_upload(xd, gd, tp)
    c2.assemble_K(sig, False)
    with pytest.raises(Exception):
        c2.chol_set_rhs(y)          # no spare row
    c2.chol_factor(lam)
    with pytest.raises(Exception):
        c2.chol_solve(None)         # no right-hand side was handed over


def _device_rows(c, rows):
    """Selected rows of the device-resident matrix (tests only)."""
    import ctypes as C

    n_rows, n_cols, extra = c.K_shape()
    p, ld = C.c_void_p(), C.c_int64()
    c._check(c._lib.gdml_K_dev(c._h, C.byref(p), C.byref(ld)))
    out = np.empty((len(rows), n_cols))
    buf = np.empty(ld.value)
    for k, r in enumerate(rows):
        src = C.c_void_p(p.value + int(r) * ld.value * 8)
        c._check(c._lib.gdml_memcpy_d2h(c._h, buf.ctypes.data_as(C.c_void_p), src, buf.nbytes))
        out[k] = buf[:n_cols]
    return out


def test_full_size_properties(ctx_factory):
    """BASELINE.json configs[1] size (N=21, N_train=1000, n=63000), size-independent properties:
    symmetry of K, agreement of the assembled rows with the matrix-free operator (two independent kernels:
    K[r,:] v == (K v)[r]), linearity of the operator, and the round trip (-K + lam I) x = y through the
    Cholesky path.  Tolerances: 1e-12 max|K| for entries, 1e-10 relative for operator identities, 1e-10 solve."""
    N, M = 21, 1000
    ds = orc.synth_dataset(N, M, seed=1, jitter=0.3)
    Rf = ds['R'].reshape(M, -1)
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    c = ctx_factory()
    xd, gd = c.desc_from_R(Rf, N)
    sig, lam = 20.0, 1e-10
    c.train_upload(xd, gd, tp)
    c.assemble_K(sig, False, alloc_extra_rows=1)
    n = M * 3 * N
    rs = np.random.RandomState(0)
    rows = np.sort(rs.choice(n, 6, replace=False))
    Kr = _device_rows(c, rows)
    # symmetry: K[r, c] == K[c, r] for the sampled rows against each other and against sampled columns
    cols = np.sort(rs.choice(n, 5, replace=False))
    Kc = _device_rows(c, cols)
    scale = np.abs(Kr).max()
    for a, r in enumerate(rows):
        for b, cc in enumerate(cols):
            assert abs(Kr[a, cc] - Kc[b, r]) <= 1e-12 * scale
    # a few blocks against the oracle
    for r in rows[:2]:
        i = r // (3 * N)
        blk = orc.assemble_K(xd[[i, 0, M - 1]], gd[[i, 0, M - 1]], orc.tril_perms_lin_from_tril_perms(tp), sig)
        assert np.abs(Kr[list(rows).index(r), :3 * N] - blk[r - i * 3 * N, 3 * N:6 * N]).max() <= 1e-12 * scale
    # assembled rows vs the matrix-free operator
    c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
    v = rs.normal(size=n)
    w = rs.normal(size=n)
    Kv = c.kernel_matvec(0.0, False, v)
    assert np.abs(Kr @ v - Kv[rows]).max() <= 1e-10 * np.abs(Kv).max()
    Kw = c.kernel_matvec(0.0, False, w)
    Kvw = c.kernel_matvec(0.0, False, 2.0 * v - 3.0 * w)
    assert np.abs(Kvw - (2.0 * Kv - 3.0 * Kw)).max() <= 1e-10 * np.abs(Kvw).max()
    # round trip through the factorisation (right-hand side carried as the extra row)
    y = ds['F'].reshape(-1) / np.std(ds['F'])
    c.chol_set_rhs(y)
    assert c.chol_factor(lam) == 0
    x = -c.chol_solve(None)
    Ax = -c.kernel_matvec(lam, False, x)  # (-K + lam I) x
    assert np.linalg.norm(Ax - y) / np.linalg.norm(y) <= 1e-10


def test_smallest_sizes_and_empty_batch(ctx_factory):
    """Edge cases: two atoms / one training point, empty query batch, a single query given as a 1-D array."""
    for N, M, P in ((2, 1, 1), (3, 2, 2), (2, 3, 2)):
        ds = orc.synth_dataset(N, M + 2, seed=N + M)
        Rf = ds['R'].reshape(M + 2, -1)
        perms = np.arange(N)[None]
        if P == 2:
            perms = np.vstack([perms, perms[0][::-1] if N == 2 else np.array([1, 0, 2])])
        tp = orc.tril_perms_from_atom_perms(perms)
        xd, gd = orc.desc_from_R(Rf[:M])
        c = ctx_factory()
        x2, g2 = c.desc_from_R(Rf[:M], N)
        assert np.abs(x2 - xd).max() <= 1e-14 * np.abs(xd).max() and np.abs(g2 - gd).max() <= 1e-14 * np.abs(gd).max()
        c.train_upload(xd, gd, tp)
        K = c.assemble_K(5.0, False, to_host=True)
        K0 = orc.assemble_K(xd, gd, orc.tril_perms_lin_from_tril_perms(tp), 5.0)
        assert np.abs(K - K0).max() <= 1e-12 * np.abs(K0).max()
        ja = np.random.RandomState(1).normal(size=xd.shape)
        c.predict_upload_model(xd, ja, tp, 5.0, None)
        E, F = c.predict(Rf[M:])
        rq, rdq = orc.desc_from_R(Rf[M:])
        E0, F0 = orc.predict_from_desc(rq, rdq, xd, ja, tp, 5.0, None)
        assert np.abs(F - F0).max() <= 1e-11 * max(np.abs(F0).max(), 1e-300)
        E1, F1 = c.predict(Rf[M])  # single geometry, 1-D
        assert F1.shape == (1, 3 * N) and np.abs(F1[0] - F[0]).max() <= 1e-14 * max(np.abs(F[0]).max(), 1e-300)
        E_, F_ = c.predict(np.empty((0, 3 * N)))
        assert E_.shape == (0,) and F_.shape == (0, 3 * N)


def test_predict_sliced_batches(golden, ctx_factory):
    """Batches above Context.max_query_batch are sent in slices; the result is the same as one call."""
    g = golden
    m = _model(g)
    tp = _tril_perms(g)
    c = ctx_factory()
    c.predict_upload_model(np.ascontiguousarray(m['R_desc'].T), m['R_d_desc_alpha'], tp, float(g['sig']), m.get('alphas_E'))
    R = np.tile(g['R_test'].reshape(len(g['R_test']), -1), (3, 1))
    E0, F0 = c.predict(R, _lat(g))
    c.max_query_batch = 4
    E1, F1 = c.predict(R, _lat(g))
    fl = cancel_floor(g) / float(m['std'])  # slices pick other split counts: same sum, other order
    assert np.abs(F0 - F1).max() <= 1e-12 * np.abs(F0).max() + fl
    assert np.abs(E0 - E1).max() <= 1e-12 * max(1.0, np.abs(E0).max()) + fl * float(g['sig'])
    _, F2 = c.predict(R, _lat(g), return_E=False)
    assert np.array_equal(F1, F2)


def test_predict_mfma_randomised(ctx_factory, monkeypatch):
    """Seeded random shapes (atoms, table size, batch, permutations, energy terms, sigma, coefficient scale):
    the MFMA kernel against the wave kernel on identical inputs, 1e-11 relative (observed: < 3e-14)."""
    rs = np.random.RandomState(123)
    for trial in range(12):
        N = int(rs.randint(3, 24)); M = int(rs.randint(5, 300)); B = int(rs.randint(256, 900))
        P = int(rs.choice([1, 2])); with_aE = bool(rs.randint(0, 2)); sig = float(rs.choice([5.0, 12.0, 40.0]))
        ds = orc.synth_dataset(N, M + B, seed=trial)
        Rf = ds['R'].reshape(M + B, -1)
        perms = np.arange(N)[None]
        if P == 2:
            p2 = np.arange(N)
            p2[[0, 1]] = p2[[1, 0]]
            perms = np.vstack([perms, p2])
        tp = orc.tril_perms_from_atom_perms(perms)
        xd, gd = orc.desc_from_R(Rf[:M])
        ja = rs.normal(size=xd.shape) * 10.0 ** rs.uniform(-2, 4)
        aE = rs.normal(size=M) if with_aE else None
        c = ctx_factory()
        c.predict_upload_model(xd, ja, tp, sig, aE)
        E1, F1 = c.predict(Rf[M:])
        c.set_option('predict.wave_only', 1)
        E0, F0 = c.predict(Rf[M:])
        c.set_option('predict.wave_only', 0)
        assert np.abs(F1 - F0).max() <= 1e-11 * np.abs(F0).max(), (N, M, B, P, with_aE, sig)
        assert np.abs(E1 - E0).max() <= 1e-11 * np.abs(E0).max(), (N, M, B, P, with_aE, sig)
        c.close()
        # ... and both against the oracle (predict.py:168-245), north_star's 1e-10
        xq, gq = orc.desc_from_R(Rf[M:])
        Eo, Fo = orc.predict_from_desc(xq, gq, xd, ja, tp, sig, alphas_E=aE)
        assert np.abs(F1 - Fo).max() <= 1e-10 * np.abs(Fo).max(), (N, M, B, P, with_aE, sig)
        assert np.abs(E1 - Eo).max() <= 1e-10 * max(1.0, np.abs(Eo).max()), (N, M, B, P, with_aE, sig)


_P6_9 = np.array([[0, 1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 0, 3, 4, 5, 6, 7, 8], [2, 0, 1, 3, 4, 5, 6, 7, 8],
                  [0, 1, 2, 4, 3, 5, 6, 7, 8], [1, 2, 0, 4, 3, 5, 6, 7, 8], [2, 0, 1, 4, 3, 5, 6, 7, 8]])


@pytest.mark.parametrize('N,M,perms', [(21, 40, None), (9, 70, None), (5, 33, None), (9, 30, _P6_9), (24, 14, None)])
def test_assemble_A_lower_form(ctx_factory, N, M, perms):
    """gdml_assemble_A: A = -K + lam I written directly, blocks on/below the block diagonal only.  The written
    part must equal -K + lam I of the oracle elementwise, and factor + solve must agree with the two-step path
    (gdml_assemble_K, sign flip and shift inside gdml_chol_factor) to the conditioning of the system."""
    ds = orc.synth_dataset(N, M, seed=5, jitter=0.3)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None] if perms is None else perms)
    sig, lam = 20.0, 1e-8
    Ko = orc.assemble_K(xo, go, orc.tril_perms_lin_from_tril_perms(tp), sig)
    n, N3 = Ko.shape[0], 3 * N
    Ao = -Ko + lam * np.eye(n)
    y = ds['F'].ravel() / np.std(ds['F'])
    c = ctx_factory()
    c.train_upload(xo, go, tp)
    c.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
    A = c.K_to_host()[:n]
    blk_lower = np.kron(np.tril(np.ones((M, M))), np.ones((N3, N3))).astype(bool)
    assert np.abs((A - Ao)[blk_lower]).max() <= 1e-12 * np.abs(Ko).max()
    c.chol_set_rhs(y)
    assert c.chol_factor(lam) == 0
    a1 = c.chol_solve(None)
    with pytest.raises(ValueError):  # a different lam than the assembly used
        c.assemble_K(sig, False, for_cholesky=lam)
        c.chol_factor(2 * lam)
    c2 = ctx_factory()
    c2.set_option('asm.lower', 0)  # same entry point, two-step form
    c2.train_upload(xo, go, tp)
    c2.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
    c2.chol_set_rhs(y)
    assert c2.chol_factor(lam) == 0
    a2 = c2.chol_solve(None)
    for a in (a1, a2):
        assert np.linalg.norm(Ao @ (-a) - y) <= 1e-9 * np.linalg.norm(y)
    assert np.abs(a1 - a2).max() <= 1e-6 * np.abs(a2).max()


@pytest.mark.parametrize('N,M', [(21, 37), (20, 13), (16, 19), (12, 30), (11, 29)])
def test_assemble_strip_kernel(ctx_factory, N, M):
    """assemble_strip.hip (64-column strips, 11 <= N <= 21, identity permutation): the full K and the lower form of A
    against the oracle, and against the block-per-wavefront kernel (asm.strip = 0).  The sizes give strips that
    straddle two (N = 21) and three column blocks, a ragged last strip and more than one row chunk."""
    ds = orc.synth_dataset(N, M, seed=3 * N + M, jitter=0.25)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    sig, lam = 17.0, 1e-7
    Ko = orc.assemble_K(xo, go, orc.tril_perms_lin_from_tril_perms(tp), sig)
    n, N3 = Ko.shape[0], 3 * N
    res = {}
    for strip in (1, 0):
        c = ctx_factory()
        c.set_option('asm.strip', strip)
        c.set_option('asm.i_chunk', 8)
        c.train_upload(xo, go, tp)
        res[strip, 'K'] = c.assemble_K(sig, False, to_host=True)
        c.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
        res[strip, 'A'] = c.K_to_host()[:n]
    scale = np.abs(Ko).max()
    blk_lower = np.kron(np.tril(np.ones((M, M))), np.ones((N3, N3))).astype(bool)
    assert np.abs(res[1, 'K'] - Ko).max() <= 1e-12 * scale
    assert np.abs((res[1, 'A'] - (-Ko + lam * np.eye(n)))[blk_lower]).max() <= 1e-12 * scale
    assert np.abs(res[1, 'K'] - res[0, 'K']).max() <= 1e-13 * scale
    assert np.abs((res[1, 'A'] - res[0, 'A'])[blk_lower]).max() <= 1e-13 * scale


@pytest.mark.parametrize('n_atoms,n_train,n_query,n_perms,with_aE', [
    (24, 40, 300, 1, False),   # D = 276: first size past the register-resident MFMA kernel
    (26, 30, 257, 2, True),    # odd D (325), permutations, energy-constraint terms, ragged query tile
    (42, 12, 256, 1, False),   # D = 861 (BASELINE configs[3] molecule size)
    (60, 5, 300, 1, True),     # D = 1770 (configs[4])
])
def test_predict_wide_mfma(ctx_factory, n_atoms, n_train, n_query, n_perms, with_aE):
    """Large molecules (D > 256): the GEMM-pipeline prediction path (predict_wide.hip) against the wave kernel on
    the same model, incl. training-set style coincident points (queries 0..M-1 ARE the training geometries)."""
    rs = np.random.RandomState(n_atoms)
    N, M, B = n_atoms, n_train, n_query
    ds = orc.synth_dataset(N, M + B, seed=4, jitter=0.3)
    Rf = ds['R'].reshape(M + B, -1)
    perms = np.arange(N)[None]
    if n_perms == 2:
        p2 = np.arange(N)
        p2[[0, 1]] = p2[[1, 0]]
        perms = np.vstack([perms, p2])
    tp = orc.tril_perms_from_atom_perms(perms)
    xd, gd = orc.desc_from_R(Rf[:M])
    ja = rs.normal(size=xd.shape)
    aE = rs.normal(size=M) if with_aE else None
    Rq = np.vstack([Rf[:M], Rf[M:M + B - M]])  # first M queries coincide with the training points
    c = ctx_factory()
    c.predict_upload_model(xd, ja, tp, 25.0, aE)
    c.set_option('predict.mfma_wide', 2)  # force the pipeline at test sizes
    E1, F1 = c.predict(Rq)
    c.set_option('predict.wave_only', 1)
    E0, F0 = c.predict(Rq)
    assert np.abs(F1 - F0).max() <= 1e-11 * np.abs(F0).max()
    assert np.abs(E1 - E0).max() <= 1e-11 * np.abs(E0).max()
    # the pipeline directly against the oracle (predict.py:168-245), north_star's 1e-10
    xq, gq = orc.desc_from_R(Rq)
    Eo, Fo = orc.predict_from_desc(xq, gq, xd, ja, tp, 25.0, alphas_E=aE, chunk=32)
    assert np.abs(F1 - Fo).max() <= 1e-10 * np.abs(Fo).max()
    assert np.abs(E1 - Eo).max() <= 1e-10 * max(1.0, np.abs(Eo).max())
