"""Round-6 GPU tests (real MI355X, through the C ABI): the well-conditioned periodic fixture without a round-off floor
(minimum-image prologue of descriptors and predictions pinned at 1e-10), the DEFAULT preconditioner form (fp32 factor +
Gram correction) against the reference's residual trace, the persistent trailing update against the plain one."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import gdml_oracle as orc  # noqa: E402
from _pcg_compare import assert_same_convergence, crossings  # noqa: E402
from tests.test_oracle_golden import _lat, _model, cancel_floor  # noqa: E402

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=True))


@pytest.fixture
def ctx():
    from sgdml_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def test_periodic_fixture_without_roundoff_floor(ctx):
    """n10_p2_pbc (make_golden.py, round 6: lam = 1e-4, max|J alpha| = 3.3e3, 57 % of the descriptor entries wrapped by
    the minimum image, desc.py:44-77): descriptors of training and query geometries, predictions from the REFERENCE's
    model for unseen periodic geometries and in training-set mode, and K -- all at the contract's tolerances with NO
    cancellation floor (it is 1.3e-12 here, 0.3 % of the tolerance; on n4_p6_pbc it was 2e-7 and hid the 8th digit)."""
    from sgdml_amd import _lib

    g = load('n10_p2_pbc')
    assert cancel_floor(g) <= 2e-12
    lat = _lat(g)
    M, N = g['R_train'].shape[:2]
    xd, gd = ctx.desc_from_R(g['R_train'].reshape(M, -1), N, lat)
    np.testing.assert_allclose(xd, g['R_desc'], rtol=1e-13, atol=0)
    np.testing.assert_allclose(gd, g['R_d_desc'], rtol=1e-12, atol=1e-15)
    # the case is periodic in earnest: the open-boundary descriptors differ
    xo, _ = ctx.desc_from_R(g['R_train'].reshape(M, -1), N)
    assert np.mean(np.abs(xo - xd) > 1e-9) > 0.3
    tp = _lib.tril_perms_from_lin(g['tril_perms_lin'], g['R_desc'].shape[1])
    ctx.train_upload(xd, gd, tp)
    K = ctx.assemble_K(float(g['sig']), False, to_host=True)
    assert np.abs(K - g['K']).max() <= 1e-12 * np.abs(g['K']).max()
    m = _model(g)
    ctx.predict_upload_model(np.ascontiguousarray(m['R_desc'].T), m['R_d_desc_alpha'], tp, float(g['sig']), None)
    Rq = g['R_test'].reshape(len(g['R_test']), -1)
    for batch in (Rq, Rq[:1], np.tile(Rq, (40, 1))):  # single-launch path, one geometry, a batch for the MFMA kernel
        E, F = ctx.predict(batch, lat)
        E, F = E * m['std'] + m['c'], F * m['std']
        reps = len(batch) // len(Rq) or 1
        Fr, Er = np.tile(g['F_test'], (reps, 1))[:len(batch)], np.tile(g['E_test'], reps)[:len(batch)]
        assert np.abs(F - Fr).max() <= 1e-10 * np.abs(g['F_test']).max()
        assert np.abs(E - Er).max() <= 1e-10 * max(1.0, np.abs(g['E_test']).max())
    # a query shifted by lattice vectors is the same periodic geometry
    shift = (g['lattice'] @ np.array([1.0, -2.0, 1.0]))[None, None, :]
    E2, F2 = ctx.predict((g['R_test'] + shift).reshape(len(Rq), -1), lat)
    assert np.abs(F2 * m['std'] - g['F_test']).max() <= 1e-9 * np.abs(g['F_test']).max()
    E, F = ctx.predict(None)
    assert np.abs(F * m['std'] - g['F_train_pred']).max() <= 1e-10 * np.abs(g['F_train_pred']).max()
    assert np.abs(E * m['std'] + m['c'] - g['E_train_pred']).max() <= 1e-10 * max(1.0, np.abs(g['E_train_pred']).max())
    # analytic solve at the contract's 1e-10 (a well-conditioned system: no exception needed)
    ctx.assemble_K(float(g['sig']), False)
    assert ctx.chol_factor(float(g['lam'])) == 0
    a = ctx.chol_solve(g['y'])
    A = -g['K'] + float(g['lam']) * np.eye(len(g['y']))
    assert np.linalg.norm(A @ (-a) - g['y']) <= 1e-10 * np.linalg.norm(g['y'])
    assert np.abs(a - g['alphas']).max() <= 1e-8 * np.abs(g['alphas']).max()  # cond ~ 1e3: coefficients themselves compare


def test_default_preconditioner_follows_the_reference_trace():
    """The preconditioner gdml_nystroem_factor picks by default above 1 GiB of factor (pcg.precon_form = 3, forced here
    at the fixture's size) on cfg2_traj_m300 -- bench.py's configs[2] workload family with the inducing columns the
    REFERENCE drew -- against the reference's own residual trace (make_golden_r3.case_cfg2_traj_m300): the first steps
    coincide, the residual first passes 0.3 / 0.1 / 3e-2 / 1e-2 of ||y|| at the reference's iteration (the band of
    assert_same_convergence that form 0 is held to), and it reaches solver_tol in the reference's iteration count
    +-10 % or earlier (the fp64 Gram correction removes the plateau the stored factor's rounding creates: earlier is
    what the form is for; the count is printed)."""
    from sgdml_amd import _lib

    g = load('cfg2_traj_m300')
    M, N = g['R_train'].shape[:2]
    sig, lam, y, idx = float(g['sig']), float(g['lam']), g['y'], g['inducing_pts_idxs']
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    c = _lib.Context()
    try:
        xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
        c.train_upload(xd, gd, tp)
        c.set_option('pcg.precon_form', 3)
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        _, _, info = c.nystroem_factor(lam, idx, want_lev=False)
        assert (info & 6) == 4
        c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
        hist = []
        x, inf, iters, resid = c.pcg(lam, False, y, rtol=1e-4, maxiter=5000, callback=lambda it, r, fetch_x: hist.append(r) or False)
        assert inf == 0
        ours, ref, n_ref = np.array(hist), g['resid_hist'], int(g['n_iters'])
        ny = np.linalg.norm(y)
        print('form 3: %d iterations (reference %d); level crossings ours %s reference %s'
              % (iters, n_ref, crossings(ours, ny), crossings(ref, ny)))
        np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-3)
        assert_same_convergence(ours, ref, ny)
        assert iters <= n_ref + max(2, n_ref // 10), (iters, n_ref)
        r = c.kernel_matvec(lam, False, x) + y
        assert np.linalg.norm(r) <= 1.05e-4 * ny
    finally:
        c.close()


def test_bench_lines_for_one_and_two_ranks_carry_the_same_metric(tmp_path):
    """`python bench.py` and `python bench.py --gpus 2 --comm host` (two ranks sharing the GPU, host-staged collectives, the
    module's own launcher) at a reduced size: ONE JSON line each, identical `metric` and `config.workload`, `scale_point`
    under the same keys, the two-rank value produced by the distributed Cholesky with a residual at the contract; the
    `--workload cg` lines of both rank counts agree with each other as well."""
    import json
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS='2')
    common = ['--steps', '1', '--warmup', '1', '--no-cpu', '--n-train', '96', '--cg-n-train', '160', '--cg-inducing', '8',
              '--cg-iters', '5', '--no-to-tol']

    def line(extra):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + common + extra, env=env, capture_output=True,
                           text=True, timeout=900, cwd=str(tmp_path))
        assert p.returncode == 0, p.stderr[-3000:]
        rows = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        assert len(rows) == 1, p.stdout[-2000:]
        return json.loads(rows[0])

    one = line(['--no-configs', '--no-profile'])
    two = line(['--gpus', '2', '--comm', 'host'])
    assert one['metric'] == two['metric'] and one['config']['workload'] == two['config']['workload']
    assert (one['n_gpus'], two['n_gpus']) == (1, 2) and one['unit'] == two['unit'] == 's'
    assert set(one['scale_point']) == set(two['scale_point']) and two['scale_point']['seconds'] == two['value'] > 0
    assert two['solve_rel_residual'] < 1e-10 and one['solve_rel_residual'] < 1e-10
    assert two['config']['collectives'] == 'host' and two['config']['rccl_ranks_seen'] is None
    assert two['cg']['s_per_step'] > 0 and two['scale_point_cg']['seconds'] == two['cg']['s_per_step']
    assert 'first_call_in_process' in one and one['first_call_in_process']['first_step_s'] > 0
    cg1 = line(['--workload', 'cg'])
    cg2 = line(['--workload', 'cg', '--gpus', '2', '--comm', 'host'])
    assert cg1['metric'] == cg2['metric'] and cg1['config']['workload'] == cg2['config']['workload']
    assert cg1['value'] > 0 and cg2['value'] > 0 and (cg1['n_gpus'], cg2['n_gpus']) == (1, 2)


def test_persistent_trailing_update_is_bitwise_the_plain_one(ctx):
    """Option gemm.persist = 1 (resident workgroups pulling tiles from per-XCD counters, csrc/chol.hip) against the default
    launch-per-tile schedule at a size whose trailing updates are long enough to take it (n = 9450: 2 775 lower tiles in the
    first fused launch, the threshold is four rounds of 512): every tile is the same arithmetic in the same order, so the
    factor's solution is BIT-identical, with the right-hand side carried through and with the merged schedule's tile counter
    (the diagonal-block workgroup waits for tiles that persistent workgroups compute)."""
    N, M = 21, 150
    ds = orc.synth_dataset(N, M, seed=4, jitter=0.3)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    y = ds['F'].ravel() / np.std(ds['F'])
    ctx.train_upload(xd, gd, tp)
    sols = {}
    for persist in (0, 1):
        ctx.set_option('gemm.persist', persist)
        ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
        ctx.chol_set_rhs(y)
        assert ctx.chol_factor(1e-10) == 0
        sols[persist] = ctx.chol_solve(None)
    assert np.array_equal(sols[0], sols[1])
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, 20.0, None)
    r = ctx.kernel_matvec(1e-10, False, -sols[1]) + y
    assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(y)


def test_narrow_tile_trailing_update_solves_the_same_system(ctx):
    """Option gemm.n64 = 1 (128 x 64 tiles, three workgroups per CU, its own diagonal-block role without the transposed copy
    of L_jj -- csrc/chol.hip; measured 6.8 % slower, profiles/r06_gemm_n64_ab.txt, so it stays an A/B option): same system,
    same solution up to the rounding of a different summation order, residual inside the contract.  n = 9450 with the carried
    right-hand-side row: merged schedule with the tile counter (20 narrow tiles per diagonal block), edge tiles in both
    dimensions, the K = 64 updates inside the panel factorisation and the plain launches of the schedule's tail."""
    N, M = 21, 150
    ds = orc.synth_dataset(N, M, seed=4, jitter=0.3)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    y = ds['F'].ravel() / np.std(ds['F'])
    ctx.train_upload(xd, gd, tp)
    sols = {}
    try:
        for n64 in (0, 1):
            ctx.set_option('gemm.n64', n64)
            ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
            ctx.chol_set_rhs(y)
            assert ctx.chol_factor(1e-10) == 0
            sols[n64] = ctx.chol_solve(None)
    finally:
        ctx.set_option('gemm.n64', 0)
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, 20.0, None)
    for n64 in (0, 1):
        r = ctx.kernel_matvec(1e-10, False, -sols[n64]) + y
        assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(y)
    assert np.linalg.norm(sols[1] - sols[0]) <= 1e-6 * np.linalg.norm(sols[0])  # cond(K) ~ 1e9 times the fp64 rounding of the two orders


def test_whole_point_index_lists_run_on_the_perm2_kernel(ctx):
    """K_nm of the iterative solver (an index list that requests every column of the listed points, iterative.py:229-247) for a
    42-atom molecule with a 27-element group: round 6 routes it to assemble_perm2_kernel (column point = jlist[v]) instead of
    the general kernel.  Against the oracle (1e-12 max|K|), against the general kernel (asm.perm2 = 0), with extra rows
    allocated (the Nystroem layout), for a row-point sub-range, and a list with ONE column missing stays on the general
    kernel and is still right."""
    import bench

    N, M = 42, 7
    perms = bench.perm_group(N, 'c3x3')
    ds = orc.synth_dataset(N, M, seed=6, jitter=0.25)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(perms)
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    sig = 60.0
    Ko = orc.assemble_K(xd, gd, lin, sig)
    scale = np.abs(Ko).max()
    ctx.train_upload(xd, gd, tp)
    pts = np.array([1, 4, 6])
    idx = (pts[:, None] * 3 * N + np.arange(3 * N)[None, :]).ravel()
    n_launch = {}
    for opt in (1, 0):
        ctx.set_option('asm.perm2', opt)
        ctx.profile(True)
        Kc = ctx.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx), to_host=True)
        ctx.profile(False)
        assert Kc.shape[1] == len(idx)
        assert np.abs(Kc[:3 * N * M] - Ko[:, idx]).max() <= 1e-12 * scale, opt
        n_launch[opt] = Kc[:3 * N * M].copy()
    assert np.abs(n_launch[1] - n_launch[0]).max() <= 1e-13 * scale
    ctx.set_option('asm.perm2', 1)
    # one column short of whole points: the general kernel (compact column lists), same numbers
    idx2 = np.delete(idx, 5)
    Kc2 = ctx.assemble_K(sig, False, idx=idx2, to_host=True)
    assert np.abs(Kc2 - Ko[:, idx2]).max() <= 1e-12 * scale


def test_discovered_permutations_feed_training_like_the_reference_group():
    """create_task without `perms` discovers the group (sgdml_amd/utils/perm.py, index-exact against the
    reference's find_perms output in tests/golden/perm_c3.npz) and the model trained from it is the model trained from the
    reference's group: same kernel matrix, same coefficients' predictions."""
    from sgdml_amd.predict import GDMLPredict
    from sgdml_amd.train import GDMLTrain

    g = dict(np.load(os.path.join(GOLDEN, 'perm_c3.npz')))
    R, z = g['R'], g['z']
    n = R.shape[0]
    E, F = _pair_labels(R)
    ds = {'type': 'd', 'name': np.array('c3'), 'theory': np.array('pair'), 'z': z, 'R': R, 'F': F, 'E': E}
    models = []
    for perms in (None, g['perms']):
        tr = GDMLTrain()
        try:
            np.random.seed(1)
            task = tr.create_task(ds, n, ds, 0, sig=12, lam=1e-8, perms=perms)  # every point: the fixture's group was found on all of them
            assert {tuple(p) for p in task['perms']} == {tuple(p) for p in g['perms']}
            models.append(tr.train(task))
        finally:
            tr.__del__()
    assert np.array_equal(models[0]['idxs_train'], models[1]['idxs_train'])
    Rq = (R[:8] + 0.05 * np.random.RandomState(2).normal(size=R[:8].shape)).reshape(8, -1)
    out = []
    for m in models:
        p = GDMLPredict(m)
        out.append(p.predict(Rq))
        del p
    # the discovered group may list the same permutations in another order: the perm-summed kernel does not care
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-9 * np.abs(out[1][1]).max()
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-9 * max(1.0, np.abs(out[1][0]).max())


def _pair_labels(R):
    n_frames, n_atoms = R.shape[:2]
    i, j = np.tril_indices(n_atoms, -1)
    diff = R[:, i, :] - R[:, j, :]
    dist = np.sqrt((diff**2).sum(-1))
    E = (1.0 / dist).sum(-1)
    gp = diff / (dist**3)[..., None]
    F = np.zeros_like(R)
    for m in range(n_frames):
        np.add.at(F[m], i, gp[m])
        np.subtract.at(F[m], j, gp[m])
    return E, F


def test_replicated_predictor_shards_query_batches():
    """GDMLPredict(devices=[...]) (replicas + query sharding; the reference wraps its torch model in
    nn.DataParallel, predict.py:375-378): the model is uploaded to one context per entry and a batch is split over them in
    threads.  On a one-GPU box two contexts on GPU 0 stand in for two devices: same energies / forces / error sums as the
    single-context predictor, small batches and the training-set mode stay on the first context."""
    from sgdml_amd.predict import GDMLPredict

    g = load('n5_p4')
    m = dict(_model(g), type='m', z=np.ones(g['R_train'].shape[1], dtype=int), perms=g['perms'])
    Rq = np.tile(g['R_test'].reshape(len(g['R_test']), -1), (40, 1))  # 280 geometries
    rs = np.random.RandomState(0)
    Rq = Rq + 0.01 * rs.normal(size=Rq.shape)
    one = GDMLPredict(m)
    two = GDMLPredict(m, devices=[0, 0])
    assert len(two._replicas) == 1
    E1, F1 = one.predict(Rq)
    E2, F2 = two.predict(Rq)
    # (a shard of 140 may take another kernel than the batch of 280: equal to rounding, not bitwise)
    assert np.abs(F1 - F2).max() <= 1e-12 * np.abs(F1).max() and np.abs(E1 - E2).max() <= 1e-12 * np.abs(E1).max()
    (F3,) = two.predict(Rq[:5], return_E=False)
    assert np.abs(F3 - F1[:5]).max() <= 1e-12 * np.abs(F1).max()
    Fl, El = F1 + 0.01 * rs.normal(size=F1.shape), E1 + 0.01
    e1, e2 = one.test_errors(Rq, Fl, El), two.test_errors(Rq, Fl, El)
    for k in e1:
        assert np.allclose(e1[k], e2[k], rtol=1e-9, atol=0), k
    two.set_R_desc(g['R_desc'])
    two.set_R_d_desc(g['R_d_desc'])
    Et, Ft = two.predict()
    assert np.abs(Ft - g['F_train_pred']).max() <= 1e-10 * np.abs(g['F_train_pred']).max() + cancel_floor(g)
    del one, two


@pytest.mark.parametrize('N,M', [(22, 9), (30, 6), (64, 5), (100, 3), (130, 2)])
def test_direct_kernel_for_large_molecules_without_a_group(ctx, N, M):
    """assemble_big1_kernel (csrc/assemble_big1.hip, round 6: P = 1, 22 <= N <= 256, dense column ranges -- one workgroup per row
    point and up to four adjacent column points) against the oracle at 1e-12 max|K|: all columns, a point range that does not
    start at a group boundary (ragged groups of column points), extra rows allocated, the lower form A = -K + lam I that the
    analytic solver consumes (blocks on / below the block diagonal: the diagonal group of a row is ragged too); and equal to the
    general kernel (asm.big1 = 0), which keeps index lists and energy constraints."""
    ds = orc.synth_dataset(N, M, seed=N, jitter=0.3)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    sig, lam = 40.0, 1e-7
    Ko = orc.assemble_K(xd, gd, lin, sig)
    scale = np.abs(Ko).max()
    ctx.train_upload(xd, gd, tp)
    got = {}
    for opt in (1, 0):
        ctx.set_option('asm.big1', opt)
        K = ctx.assemble_K(sig, False, to_host=True)
        assert np.abs(K - Ko).max() <= 1e-12 * scale, opt
        lo, hi = 1, min(M, 8)
        Ks = ctx.assemble_K(sig, False, points=(lo, hi), alloc_extra_rows=3, to_host=True)
        assert np.abs(Ks[:3 * N * M] - Ko[:, 3 * N * lo:3 * N * hi]).max() <= 1e-12 * scale, opt
        ctx.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
        A = ctx.K_to_host()[:3 * N * M]
        low = np.kron(np.tril(np.ones((M, M))), np.ones((3 * N, 3 * N))).astype(bool)
        assert np.abs((A - (-Ko + lam * np.eye(3 * N * M)))[low]).max() <= 1e-12 * scale, opt
        got[opt] = (K, A[low])
    assert np.abs(got[1][0] - got[0][0]).max() <= 1e-13 * scale
    # the analytic solve through the lower form
    ctx.set_option('asm.big1', 1)
    y = ds['F'].ravel() / np.std(ds['F'])
    ctx.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
    ctx.chol_set_rhs(y)
    assert ctx.chol_factor(lam) == 0
    a = ctx.chol_solve(None)
    Am = -Ko + lam * np.eye(3 * N * M)
    assert np.linalg.norm(Am @ (-a) - y) <= 1e-10 * np.linalg.norm(y)


def _matching_cases():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'perm_c3.npz'))
    lat = g['lat']
    yield 'c3', g['R'], g['z'], None
    yield 'c3_small', g['R2'], g['z2'], None
    yield 'c3_small_lattice', g['R2'], g['z2'], (lat, np.linalg.inv(lat))
    rs = np.random.RandomState(0)
    for N, M in ((30, 60), (70, 24), (130, 8)):  # 130 atoms: more than two lanes' worth of columns per row scan
        R = (rs.normal(size=(N, 3)) * 2.0)[None] + 0.05 * rs.normal(size=(M, N, 3))
        z = rs.choice([1, 6, 8], size=N)
        same = np.where(z == z[0])[0][:2]
        R[::2][:, same] = R[::2][:, same[::-1]]  # every other geometry with two like atoms exchanged: non-identity matches
        yield 'random_n%d' % N, R, z, None


def test_device_matching_is_the_host_matching(ctx):
    """SURVEY 8(f)4 on the device (csrc/perm_match.hip, gdml_perm_match): one wavefront per pair of geometries builds the cost
    matrix, solves the assignment problem (shortest augmenting paths, the algorithm behind scipy's linear_sum_assignment) and
    applies the reference's acceptance rule (perm.py:76-89).  Against the NumPy / SciPy form in sgdml_amd/utils/perm.py, which
    the CPU suite pins to the reference's output: the SAME pairs kept, the SAME assignments, pair costs to 1e-13 relative; and
    find_perms with the device matching returns the reference's groups (tests/golden/perm_c3.npz) element for element.  The
    device side computes its own eigenvectors (sym_eig_kernel), the host side LAPACK's: same assignments."""
    from sgdml_amd.utils import perm

    kept_nontrivial = 0
    for name, R, z, lat in _matching_cases():
        fh, ch = perm.bipartite_match(R, z, lat)
        fd, cd = perm.bipartite_match(R, z, lat, ctx=ctx)
        assert set(fh) == set(fd), name
        for k in fh:
            assert np.array_equal(fh[k], fd[k]), (name, k)
        ch, cd = ch.toarray(), cd.toarray()
        fin = np.isfinite(ch)
        assert np.array_equal(fin, np.isfinite(cd))
        assert np.abs(ch[fin] - cd[fin]).max() <= 1e-13 * np.abs(ch[fin]).max(), name
        kept_nontrivial += len(fh)
    assert kept_nontrivial > 500
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'perm_c3.npz'))
    lat = g['lat']
    assert np.array_equal(perm.find_perms(g['R'], g['z'], ctx=ctx), g['perms'])
    assert np.array_equal(perm.find_perms(g['R2'], g['z2'], ctx=ctx), g['perms2'])
    assert np.array_equal(perm.find_perms(g['R2'], g['z2'], lat_and_inv=(lat, np.linalg.inv(lat)), ctx=ctx), g['perms3'])


def test_device_eigenvectors_match_lapack(ctx):
    """gdml_sym_eig_absv (batched cyclic Jacobi, one workgroup per matrix; what gdml_perm_match runs when it is handed no
    eigenvectors -- the reference: numpy.linalg.eig per geometry, perm.py:183-187): |V| with columns by decreasing eigenvalue
    against LAPACK's eigh for distance matrices of 1 ... 150 atoms -- both matrices in LDS (N <= 100), the rotations in device
    memory (N = 101, 130), both there (N = 150); odd N (a padding index in the round-robin), N = 1 and 2.  Tolerance: 1e-9
    absolute on unit vectors whose eigenvalue gaps are >= 1e-8 of the spectrum (measured 3e-11)."""
    import bench
    from sgdml_amd.utils import perm

    for N, M in ((1, 3), (2, 3), (3, 5), (8, 40), (21, 100), (42, 40), (99, 6), (100, 6), (101, 4), (130, 3), (150, 2)):
        R, _, _ = bench.synth_geometries(max(N, 2), M, seed=1)
        adj = perm._dist_matrices(R.reshape(M, max(N, 2), 3)[:, :N])
        w, v = np.linalg.eigh(adj)
        ref = np.abs(v[:, :, ::-1])
        got = ctx.sym_eig_absv(adj)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-9, (N, np.abs(got - ref).max())
        # columns are unit vectors, mutually orthogonal in absolute value only by accident: check the norms
        assert np.abs(np.linalg.norm(got, axis=1) - 1.0).max() <= 1e-12


def test_device_matching_edge_sizes(ctx):
    """One geometry (no pairs), two geometries, one atom, and a result buffer that is too small (the wrapper repeats the call
    with room for every pair)."""
    from sgdml_amd.utils import perm

    rs = np.random.RandomState(3)
    R = rs.normal(size=(1, 5, 3))
    f, c = perm.bipartite_match(R, np.ones(5, int), ctx=ctx)
    assert f == {} and c.shape == (1, 1)
    R = rs.normal(size=(2, 1, 3))
    f, c = perm.bipartite_match(R, np.ones(1, int), ctx=ctx)
    assert f == {}
    # 12 geometries of a 4-atom molecule, every second one with atoms 0 and 1 exchanged: 36 of 66 pairs are kept; room for 4
    base = rs.normal(size=(4, 3)) * 2
    R = base[None] + 0.01 * rs.normal(size=(12, 4, 3))
    R[::2][:, [0, 1]] = R[::2][:, [1, 0]]
    adj = perm._dist_matrices(R)
    w, v = np.linalg.eigh(adj)
    absv = np.abs(v[:, :, ::-1])
    cost = np.zeros((12, 12))
    ij, pm = np.empty((4, 2), np.int32), np.empty((4, 4), np.int32)
    import ctypes as C
    from sgdml_amd._lib import _ptr
    n = C.c_int64(0)
    sp = np.zeros(4, np.int32)
    rc = ctx._lib.gdml_perm_match(ctx._h, _ptr(np.ascontiguousarray(absv)), _ptr(np.ascontiguousarray(adj)), _ptr(sp), 12, 4,
                                  _ptr(cost), _ptr(ij), _ptr(pm), 4, C.byref(n))
    assert rc == 0 and n.value == 36
    fh, _ = perm.bipartite_match(R, np.ones(4, int))
    assert len(fh) == 36
    for (i, j), p in zip(ij, pm):
        assert np.array_equal(fh[int(i), int(j)], p)
    fd, _ = perm.bipartite_match(R, np.ones(4, int), ctx=ctx)  # the wrapper's capacity (4 M = 48) suffices here
    assert set(fd) == set(fh)


def test_create_task_discovers_the_group_on_the_device():
    """create_task without `perms` (train.py:560-584): the matching runs through gdml_perm_match on the trainer's context and the
    task carries the reference's group (tests/golden/perm_c3.npz)."""
    from sgdml_amd.train import GDMLTrain

    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'perm_c3.npz'))
    n = g['R'].shape[0]
    rs = np.random.RandomState(0)
    ds = {'type': 'd', 'name': np.array('c3'), 'theory': np.array('none'), 'z': g['z'], 'R': g['R'],
          'F': rs.normal(size=g['R'].shape), 'E': rs.normal(size=n)}
    tr = GDMLTrain()
    try:
        np.random.seed(1)
        task = tr.create_task(ds, n, ds, 0, sig=10)
        assert {tuple(p) for p in task['perms']} == {tuple(p) for p in g['perms']}
        assert tr._context().phase_ms('perm_match')[1] >= 1  # the device kernel ran
    finally:
        tr.__del__()


def test_wide_contractions_on_padded_tables_and_query_chunks(ctx):
    """Round 6, csrc/predict_wide.hip (D > 256): tables padded to a multiple of 64 rows, query chunks to a multiple of 128, the
    first contraction overwriting its block (gemm_nt_neg_kernel: nothing cleared, nothing read back), the back contraction loading
    unchecked behind the used columns.  (a) against round 5's shapes (predict.wide_pad = 0) at sizes where nothing is a multiple
    of anything -- M P = 27 x 31 = 837 rows, D = 861, 333 queries; (b) a batch that needs TWO query chunks (M P = 54 000 table
    rows: chunks of 2 304, the second one 696 = 5.4 tiles) against the same queries in single-chunk calls; (c) both against the
    oracle on a few queries."""
    import bench
    from sgdml_amd.utils.desc import Desc

    N = 42
    perms = bench.perm_group(N, 'c3x3')
    tril = np.array([Desc.perm(p_) for p_ in perms])
    rs = np.random.RandomState(11)
    for M, B in ((31, 333), (2000, 3000)):
        R, _, _ = bench.synth_trajectory(N, M, seed=3, n_modes=8, amp=0.15, noise=0.01)
        xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
        alphas = rs.normal(size=(M, 3 * N))
        JA = Desc(N).d_desc_dot_vec(gd, alphas)
        ctx.predict_upload_model(xd, JA, tril, 60.0, None)
        Rq = R.reshape(M, -1)[rs.randint(0, M, size=B)] + 0.02 * rs.normal(size=(B, 3 * N))
        E1, F1 = ctx.predict(Rq, None)
        if M == 31:
            ctx.set_option('predict.wide_pad', 0)
            try:
                E0, F0 = ctx.predict(Rq, None)
            finally:
                ctx.set_option('predict.wide_pad', 1)
            assert np.abs(F1 - F0).max() <= 1e-12 * np.abs(F0).max() and np.abs(E1 - E0).max() <= 1e-12 * np.abs(E0).max()
            xq, gq = orc.desc_from_R(Rq[:4])
            Eo, Fo = orc.predict_from_desc(xq, gq, xd, JA, tril, 60.0)
            assert np.abs(F1[:4] - Fo).max() <= 1e-10 * np.abs(Fo).max() and np.abs(E1[:4] - Eo).max() <= 1e-10 * np.abs(Eo).max()
        else:
            for lo in (0, 1000, 2000):  # one chunk each
                Es, Fs = ctx.predict(Rq[lo:lo + 1000], None)
                assert np.abs(F1[lo:lo + 1000] - Fs).max() <= 1e-12 * np.abs(F1).max()
                assert np.abs(E1[lo:lo + 1000] - Es).max() <= 1e-12 * np.abs(E1).max()


def test_energy_constraint_system_of_several_row_blocks_against_the_reference(ctx):
    """ecstr_n9_p6_m40 (make_golden_r6.py: N = 9, 6-element group, M = 40, use_E_cstr, n = 1120, lam = 1e-8): the energy ROWS of the
    K the reference assembled (train.py:235-248) and, by the symmetry the distributed Cholesky relies on, its energy COLUMNS
    (train.py:250-300) from the device assembly at 1e-12; a slice and an index list that mix force and energy columns; the
    reference's coefficients through the single-GPU analytic solve by residual (1e-10) and value (cond ~ 1e9), and its predictions
    for unseen geometries from the model GDMLTrain.train builds (energy-constraint term of predict.py included)."""
    from sgdml_amd.predict import GDMLPredict
    from sgdml_amd.train import GDMLTrain

    g = load('ecstr_n9_p6_m40')
    M, N = g['R_train'].shape[:2]
    n_ff = 3 * N * M
    sig, lam = float(g['sig']), float(g['lam'])
    xd, gd = ctx.desc_from_R(g['R_train'].reshape(M, -1), N)
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    ctx.train_upload(xd, gd, tp)
    K = ctx.assemble_K(sig, True, to_host=True)
    ref, scale = g['K_E_rows'], np.abs(g['K_E_rows']).max()
    assert K.shape == (n_ff + M, n_ff + M)
    assert np.abs(K[n_ff:] - ref).max() <= 1e-12 * scale
    assert np.abs(K[:, n_ff:] - ref.T).max() <= 1e-12 * scale
    Ks = ctx.assemble_K(sig, True, points=(M - 2, M + 5), to_host=True)  # two force points, five energy columns
    assert np.abs(Ks[n_ff:] - np.hstack((ref[:, (M - 2) * 3 * N:n_ff], ref[:, n_ff:n_ff + 5]))).max() <= 1e-12 * scale
    idx = np.sort(np.concatenate((np.random.RandomState(3).choice(n_ff, 85, replace=False), n_ff + np.array([1, 7, M - 1]))))
    Ki = ctx.assemble_K(sig, True, idx=idx, to_host=True)
    assert np.abs(Ki[n_ff:] - ref[:, idx]).max() <= 1e-12 * scale
    # analytic solve on one GPU (chol.hip: full matrix, negated and regularised by the factorisation)
    ctx.assemble_K(sig, True, alloc_extra_rows=1, for_cholesky=lam)
    ctx.chol_set_rhs(g['y'])
    ctx.chol_factor(lam)
    a = ctx.chol_solve(None)
    a_ref = np.hstack((g['alphas_F'], g['alphas_E']))
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, sig, np.zeros(M))
    assert np.linalg.norm(ctx.kernel_matvec(lam, True, -a) + g['y']) <= 1e-10 * np.linalg.norm(g['y'])
    assert np.abs(a - a_ref).max() <= 1e-5 * np.abs(a_ref).max()
    # drop-in training and prediction
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': g['z'], 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x',
        'sig': int(sig), 'lam': lam, 'use_E': True, 'use_E_cstr': True, 'use_sym': True, 'perms': g['perms'],
    }
    tr = GDMLTrain()
    try:
        model = tr.train(task)
    finally:
        tr.__del__()
    assert model['solver_name'] == 'analytic' and 'alphas_E' in model
    assert abs(float(model['c']) - float(g['model_c'])) <= 1e-6 * max(1.0, abs(float(g['model_c'])))
    nt = len(g['R_test'])
    E, F = GDMLPredict(model).predict(g['R_test'].reshape(nt, -1))
    assert np.abs(F - g['F_test']).max() <= 1e-7 * np.abs(g['F_test']).max()
    assert np.abs(E - g['E_test']).max() <= 1e-7 * max(1.0, np.abs(g['E_test']).max())


@pytest.mark.parametrize('W,f', [(2048, 2.0), (2048, 64.0), (3072, 8.0)])
def test_two_level_factorisation_solves_the_same_system(ctx, W, f):
    """Option chol.block = W (csrc/chol.hip, round 6: column blocks of W columns, each factored with all rows below it carried along,
    then ONE lower update of depth W for everything to its right; measured 0.8 % slower than the one-level schedule --
    profiles/r06_chol_block.txt -- so it stays an A/B option): same system, residual inside the contract, same solution up to the
    rounding of a different summation order.  n = 9450 with the carried right-hand-side row: four / three blocks, the last
    taking the sliver; chol.block_f moves the point where a tall block leaves the fused / paired forms (both ends exercised).
    A failing pivot is reported at its global position."""
    N, M = 21, 150
    ds = orc.synth_dataset(N, M, seed=4, jitter=0.3)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    y = ds['F'].ravel() / np.std(ds['F'])
    ctx.train_upload(xd, gd, tp)
    sols = {}
    try:
        for blk in (0, W):
            ctx.set_option('chol.block', blk)
            ctx.set_option('chol.block_f', f)
            ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
            ctx.chol_set_rhs(y)
            assert ctx.chol_factor(1e-10) == 0
            sols[blk] = ctx.chol_solve(None)
        # not positive definite (lam far below zero): both schedules stop at the same leading minor
        infos = []
        for blk in (0, W):
            ctx.set_option('chol.block', blk)
            ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=-1e-3)
            ctx.chol_set_rhs(y)
            try:
                infos.append(ctx.chol_factor(-1e-3))
            except np.linalg.LinAlgError as e:
                infos.append(str(e))
        assert infos[0] == infos[1] and infos[0] != 0, infos
    finally:
        ctx.set_option('chol.block', 0)
        ctx.set_option('chol.block_f', 2.0)
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, 20.0, None)
    for blk in (0, W):
        r = ctx.kernel_matvec(1e-10, False, -sols[blk]) + y
        assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(y)
    assert np.linalg.norm(sols[W] - sols[0]) <= 1e-6 * np.linalg.norm(sols[0])


def test_fill_aware_split_counts_change_nothing_but_the_order_of_sums(ctx):
    """Round 6, late: two launches whose unit count is chosen by how well it fills the chip's 512 workgroup slots.
    (a) predict.tn_fill -- the split count of the prediction back contraction (D > 256): configs[3]-like shape, 2000 x 27 table
    rows, 1000 queries: forces / energies against round 5's rule at 1e-12 (another partition of the same sum) and against the
    oracle on a few queries.  (b) nys.syrk_split -- the Gram matrix K_nm^T K_nm of the Nystroem build cut along the rows: the
    factor L^-1 K_mn (all rows, host copy), its leverage scores and one preconditioner application against the one-pass Gram
    matrix on the same system (lam = 1e-8: both are the same numbers up to cond x eps)."""
    import bench
    from sgdml_amd.utils.desc import Desc

    N = 42
    perms = bench.perm_group(N, 'c3x3')
    tril = np.array([Desc.perm(p_) for p_ in perms])
    rs = np.random.RandomState(5)
    M, B = 2000, 1000
    R, _, _ = bench.synth_trajectory(N, M, seed=3, n_modes=8, amp=0.15, noise=0.01)
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    JA = Desc(N).d_desc_dot_vec(gd, rs.normal(size=(M, 3 * N)))
    ctx.predict_upload_model(xd, JA, tril, 60.0, None)
    Rq = R.reshape(M, -1)[rs.randint(0, M, size=B)] + 0.02 * rs.normal(size=(B, 3 * N))
    E1, F1 = ctx.predict(Rq, None)
    ctx.set_option('predict.tn_fill', 0)
    try:
        E0, F0 = ctx.predict(Rq, None)
    finally:
        ctx.set_option('predict.tn_fill', 1)
    assert np.abs(F1 - F0).max() <= 1e-12 * np.abs(F0).max() and np.abs(E1 - E0).max() <= 1e-12 * np.abs(E0).max()
    xq, gq = orc.desc_from_R(Rq[:3])
    Eo, Fo = orc.predict_from_desc(xq, gq, xd, JA, tril, 60.0)
    assert np.abs(F1[:3] - Fo).max() <= 1e-10 * np.abs(Fo).max() and np.abs(E1[:3] - Eo).max() <= 1e-10 * np.abs(Eo).max()

    # (b) 21 atoms, 700 points (n = 44 100 rows), 1200 inducing columns: 10 x 11 / 2 = 55 Gram tiles -> 8 row ranges
    N, M, lam = 21, 700, 1e-8
    R, _, _ = bench.synth_trajectory(N, M, seed=4, n_modes=8, amp=0.15, noise=0.01)
    tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tp)
    idx = np.sort(rs.choice(M * 3 * N, 1200, replace=False))
    v = rs.normal(size=M * 3 * N)
    out = {}
    try:
        for split in (1, 0):
            ctx.set_option('nys.syrk_split', split)
            ctx.set_option('pcg.precon_form', 0)
            ctx.assemble_K(20.0, False, idx=idx, alloc_extra_rows=len(idx))
            lev, fac, info = ctx.nystroem_factor(lam, idx, want_factor=True, want_lev=True)
            out[split] = (lev, fac, ctx.precon_apply(lam, v), info)
    finally:
        ctx.set_option('nys.syrk_split', 1)
        ctx.set_option('pcg.precon_form', 2)
    (l1, f1, z1, i1), (l0, f0, z0, i0) = out[1], out[0]
    assert i1 == i0
    assert np.abs(l1 - l0).max() <= 1e-6 * np.abs(l0).max()
    assert np.abs(f1 - f0).max() <= 1e-6 * np.abs(f0).max()
    assert np.abs(z1 - z0).max() <= 1e-5 * np.abs(z0).max()


def test_iterative_solver_with_energy_constraints_vs_reference():
    """The reference's Iterative.solve WITH energy constraints (fixture pcg_ecstr_n9_p6_m150, make_golden_r6.py: N = 9, P = 6,
    M = 150, n = 4200, lam = 1e-8, k = 3 inducing points it drew itself, four of the 81 inducing columns are energy columns): our
    K_nm for its columns equals the one it assembled -- force and energy rows -- (1e-12), gdml_pcg on that preconditioner follows
    scipy's residual history (first 8 steps 1e-5, level crossings), converges in the same number of iterations (+-10 %), both
    coefficient vectors predict alike; then the drop-in path (GDMLTrain.train -> Iterative under the reference's seed: the same
    inducing columns) against the reference's predictions."""
    from sgdml_amd import _lib
    from sgdml_amd.predict import GDMLPredict
    from sgdml_amd.train import GDMLTrain

    g = load('pcg_ecstr_n9_p6_m150')
    M, N = g['R_train'].shape[:2]
    n_ff = 3 * N * M
    sig, lam, y = float(g['sig']), float(g['lam']), g['y']
    idx = g['inducing_pts_idxs']
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    c = _lib.Context()
    try:
        xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
        c.train_upload(xd, gd, tp)
        K_nm = c.assemble_K(sig, True, idx=idx, to_host=True)
        assert np.abs(K_nm[g['K_nm_rows']] - g['K_nm_sample']).max() <= 1e-12 * float(g['K_nm_absmax'])
        assert abs(np.linalg.norm(K_nm) - float(g['K_nm_fro'])) <= 1e-11 * float(g['K_nm_fro'])
        c.set_option('pcg.precon_form', 0)  # the reference's operator
        c.assemble_K(sig, True, idx=idx, alloc_extra_rows=len(idx))
        c.nystroem_factor(lam, idx)
        c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, np.zeros(M))
        hist = []
        x, info, iters, resid = c.pcg(lam, True, y, rtol=1e-4, maxiter=20000,
                                      callback=lambda it, r, fetch_x: hist.append(r) or False)
        assert info == 0
        n_ref = int(g['n_iters'])
        assert abs(iters - n_ref) <= max(2, n_ref // 10), (iters, n_ref)
        ref, ours = g['resid_hist'], np.array(hist)
        np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-5)
        assert_same_convergence(ours, ref, np.linalg.norm(y))
        a_ref = g['alphas']
        assert np.linalg.norm(-x - a_ref) <= 5e-2 * np.linalg.norm(a_ref)  # two solves to tol = 1e-4 of a system with cond ~ 1e8
    finally:
        c.close()
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': g['z'], 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x',
        'sig': int(sig), 'lam': lam, 'use_E': True, 'use_E_cstr': True, 'use_sym': True, 'perms': g['perms'],
    }
    tr = GDMLTrain()
    try:
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = int(g['k'])
        tr.emulate_reference_rng = True
        np.random.seed(int(g['seed']))
        model = tr.train(task)
    finally:
        tr.__del__()
    assert np.array_equal(model['inducing_pts_idxs'], idx)
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
    assert abs(int(model['solver_iters']) - n_ref) <= max(3, n_ref // 5)
    nt = len(g['R_test'])
    E, F = GDMLPredict(model).predict(g['R_test'].reshape(nt, -1))
    assert np.abs(F - g['F_test']).max() <= 2e-2 * np.abs(g['F_test']).max()


@pytest.mark.parametrize('N,M,B', [(26, 30, 200), (26, 30, 3), (50, 12, 150), (50, 12, 2), (100, 6, 130)])
def test_energy_constraint_coefficients_beyond_256_descriptor_entries(ctx, N, M, B):
    """alphas_E (the energy-constraint term of predict.py:226-245) in the prediction paths for D > 256 -- the GEMM pipeline of
    csrc/predict_wide.hip for batches, predict_big_kernel for a few geometries -- and in the matrix-free operator with energy
    constraints there (kernel_matvec, use_E_cstr): against the oracle with a two-element group, 1e-10."""
    swp = list(range(N))
    swp[3], swp[4] = 4, 3
    perms = np.array([list(range(N)), swp])
    tp = orc.tril_perms_from_atom_perms(perms)
    ds = orc.synth_dataset(N, M + B, seed=71, jitter=0.3)
    rs = np.random.RandomState(8)
    xd, gd = orc.desc_from_R(ds['R'][:M].reshape(M, -1))
    alphas, aE = rs.normal(size=(M, 3 * N)), rs.normal(size=M)
    JA = orc.d_desc_dot_vec(gd, alphas)
    ctx.predict_upload_model(xd, JA, tp, 40.0, aE)
    Rq = ds['R'][M:].reshape(B, -1)
    E, F = ctx.predict(Rq, None)
    xq, gq = orc.desc_from_R(Rq)
    Eo, Fo = orc.predict_from_desc(xq, gq, xd, JA, tp, 40.0, alphas_E=aE)
    assert np.abs(F - Fo).max() <= 1e-10 * np.abs(Fo).max()
    assert np.abs(E - Eo).max() <= 1e-10 * max(1.0, np.abs(Eo).max())
    if B > 100:  # the operator of the iterative solver on the same training set
        ctx.train_upload(xd, gd, tp)
        ctx.predict_upload_model(xd, np.zeros_like(xd), tp, 40.0, np.zeros(M))
        v = rs.normal(size=M * 3 * N + M)
        Kv = ctx.kernel_matvec(1e-10, True, v)
        Kv0 = orc.kernel_matvec(xd, gd, tp, 40.0, 1e-10, v, True)
        assert np.abs(Kv - Kv0).max() <= 1e-11 * np.abs(Kv0).max()
