"""GPU parity, round 4 (all through the C ABI):
  * the mode matrix of assemble_perm_kernel (csrc/assemble_perm.hip) that round 3 only exercised through
    tools/asm_perm_check.py: full / lower form / energy constraints / index list / point range x LDS residency levels
    0-3 x workgroup shapes x N in {24, 42, 65, 100} x P in {1, 6, 27}, 1e-12 of max|K| against the oracle
    (sgdml/train.py:148-159, :235-300, :1376-1407);
  * the reference's iterative solver with a real permutation group (fixture pcg_n12_p6_m200, make_golden_r4.py): its
    K_nm (index-list assembly), PCG residual history, iteration count, predictions (sgdml/solvers/iterative.py:473-825);
  * the reference's column modes on a 24-atom molecule with a 6-element group (fixture cols_n24_p6);
  * predict_big_kernel<8> (1024 < D <= 4096) and <16> against the oracle."""
import functools
import os

import numpy as np
import pytest

from oracle import gdml_oracle as orc
from _pcg_compare import assert_same_convergence

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def group_perms(N, kind):
    """Closed permutation groups on N atoms: 'c3xc2' (6 elements), 'c3^3' (27), 'id' (1); identity first."""
    idt = tuple(range(N))

    def rot3(a):
        p = list(idt)
        p[a], p[a + 1], p[a + 2] = a + 1, a + 2, a
        return tuple(p)

    def swap(a):
        p = list(idt)
        p[a], p[a + 1] = a + 1, a
        return tuple(p)

    gens = {'id': [], 'c3xc2': [rot3(0), swap(N - 2)], 'c3^3': [rot3(0), rot3(3), rot3(N - 3)]}[kind]
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
    return np.array([idt] + sorted(G - {idt}))


@functools.lru_cache(maxsize=None)
def _oracle_case(N, M, kind):
    ds = orc.synth_dataset(N, M, seed=N + 7 * M, jitter=0.3)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    perms = group_perms(N, kind)
    tp = orc.tril_perms_from_atom_perms(perms)
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    sig = 13.0
    KoE = orc.assemble_K(xo, go, lin, sig, use_E_cstr=True)
    n = M * 3 * N
    Ko = np.ascontiguousarray(KoE[:n, :n])  # force rows / columns of the extended matrix are the plain K
    return xo, go, tp, sig, Ko, KoE


def _oracle_case_m(N, M, kind):
    return _oracle_case(N, M, kind)


M_OF_N = {24: 4, 42: 3, 65: 3, 100: 2}
# residency levels of the row-point image / G_j strip / x_j tables in LDS (asm.perm_level), the automatic choice, and the
# two workgroup shapes (4 / 8 wavefronts), one LDS image buffer, one permutation per group, the slow store path
OPTION_SETS = [{}, {'asm.perm_level': 0}, {'asm.perm_level': 1}, {'asm.perm_level': 2}, {'asm.perm_level': 3},
               {'asm.perm_w': 4}, {'asm.perm_w': 8}, {'asm.perm_nimg': 1, 'asm.perm_pg': 1}, {'asm.perm_fast_store': 0},
               {'asm.perm_i_chunk': 2}]


@pytest.mark.parametrize('kind', ['id', 'c3xc2', 'c3^3'])
@pytest.mark.parametrize('N', [24, 42, 65, 100])
def test_assemble_perm_mode_matrix(N, kind):
    # assemble_perm_kernel itself (asm.perm2 = 0: N = 42, P = 27 would otherwise go to assemble_perm2_kernel) + the default dispatch
    _check_assembly_modes(N, M_OF_N[N], kind, [{}] + [dict(o, **{'asm.perm2': 0}) for o in OPTION_SETS])


@pytest.mark.parametrize('N,kind', [(26, 'c3xc2'), (33, 'c3xc2'), (36, 'c3^3'), (42, 'c3^3'), (42, 'c3xc2'), (25, 'c3^3')])
def test_assemble_perm2_mode_matrix(N, kind):
    """csrc/assemble_perm2.hip (fp64-MFMA outer products, fixed-atom split, once-per-block single / diagonal terms of fixed atoms)
    forced on every size it supports (by default it only takes N >= 40, P >= 16, where it is faster): the dense modes (full, lower,
    point ranges) run on it, the others on assemble_perm_kernel; without the split, without the once-per-block pass, short and
    unchunked rows, two row points per workgroup.  1e-12 of max|K| against the independent restatement (train.py:97-302)."""
    force = {'asm.perm2_min_n': 25, 'asm.perm2_min_p': 2}
    sets = [{}, {'asm.perm2_split': 0}, {'asm.perm2_post': 0}, {'asm.perm2_ed': 0}, {'asm.perm2_es': 0}, {'asm.perm2_direct': 0}, {'asm.perm2_chunk': 5}, {'asm.perm2_chunk': 100}, {'asm.perm2_i_chunk': 2}]
    _check_assembly_modes(N, 4 if N < 30 else 3, kind, [dict(o, **force) for o in sets])


@pytest.mark.parametrize('N,kind', [(130, 'c3xc2'), (150, 'id'), (172, 'id'), (172, 'c3xc2')])
def test_assembly_beyond_128_atoms(N, kind):
    """Molecules larger than the 128 atoms rounds 1-3 stopped at (the reference has no size limit, train.py:97-302):
    permutation entries from the LDS copy instead of lane-held rows (assemble_perm_kernel<..., 2>), energy-constraint
    columns with 3N > 512 through ecol_big_kernel (N = 172: 3N = 516).  Same mode matrix, 1e-12 of max|K| vs the oracle."""
    _check_assembly_modes(N, 2, kind, [{}, {'asm.perm_fast_store': 0}, {'asm.perm_i_chunk': 2}, {'asm.perm_pg': 1}])


def test_assemble_perm2_options_apply_to_the_next_call():
    """Options are read at the point of use (include/gdml_hip.h): asm.perm2_split / _chunk / _post / _es set between two
    calls on ONE context rebuild the kernel's plan (atom renumbering, tasks, tables) instead of being ignored."""
    from sgdml_amd import _lib

    xo, go, tp, sig, Ko, KoE = _oracle_case(42, 3, 'c3^3')
    scale = np.abs(Ko).max()
    c = _lib.Context()
    try:
        for k, v in {'asm.wave': 0, 'asm.strip': 0, 'asm.pts': 0}.items():
            c.set_option(k, v)
        c.train_upload(xo, go, tp)
        for opts in [{}, {'asm.perm2_split': 0}, {'asm.perm2_split': 1, 'asm.perm2_chunk': 5}, {'asm.perm2_post': 0}, {'asm.perm2_post': 1, 'asm.perm2_es': 0},
                     {'asm.perm2_es': 1, 'asm.perm2_chunk': 12}, {'asm.perm2_direct': 0}]:
            for k, v in opts.items():
                c.set_option(k, v)
            K = c.assemble_K(sig, False, to_host=True)
            assert np.abs(K - Ko).max() <= 1e-12 * scale, opts
    finally:
        c.close()


def _check_assembly_modes(N, M, kind, option_sets):
    from sgdml_amd import _lib

    xo, go, tp, sig, Ko, KoE = _oracle_case(N, M, kind)
    N3, n = 3 * N, M * 3 * N
    scale = np.abs(Ko).max()
    lam = 1e-7
    low = np.kron(np.tril(np.ones((M, M))), np.ones((N3, N3))).astype(bool)
    Ao = -Ko + lam * np.eye(n)
    rng = np.random.default_rng(N)
    # arbitrary sorted columns incl. energy-constraint columns (>= n) in the E-constraint variant
    idx = np.sort(rng.choice(n, size=min(n, 2 * N3 + 7), replace=False))
    idxE = np.sort(np.concatenate([rng.choice(n, size=N3 + 5, replace=False), n + np.arange(M)[::2]]))
    p0, p1 = M // 3, M // 3 + max(1, M // 2)
    for opts in option_sets:
        c = _lib.Context()
        try:
            c.set_option('asm.wave', 0)
            c.set_option('asm.strip', 0)
            c.set_option('asm.pts', 0)  # the general kernel also where a specialised one exists
            for k, v in opts.items():
                c.set_option(k, v)
            c.train_upload(xo, go, tp)
            tag = '%s N=%d %s' % (opts, N, kind)
            K = c.assemble_K(sig, False, to_host=True)
            assert np.abs(K - Ko).max() <= 1e-12 * scale, 'full ' + tag
            c.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
            A = c.K_to_host()[:n]
            assert np.abs((A - Ao)[low]).max() <= 1e-12 * scale, 'lower ' + tag
            KE = c.assemble_K(sig, True, to_host=True)
            assert np.abs(KE - KoE).max() <= 1e-12 * np.abs(KoE).max(), 'ecstr ' + tag
            Kc = c.assemble_K(sig, False, idx=idx, to_host=True)
            assert np.abs(Kc - Ko[:, idx]).max() <= 1e-12 * scale, 'index ' + tag
            KcE = c.assemble_K(sig, True, idx=idxE, to_host=True)
            assert np.abs(KcE - KoE[:, idxE]).max() <= 1e-12 * np.abs(KoE).max(), 'index+ecstr ' + tag
            Kp = c.assemble_K(sig, False, points=(p0, p1), to_host=True)
            assert np.abs(Kp - Ko[:, p0 * N3:p1 * N3]).max() <= 1e-12 * scale, 'points ' + tag
            Kx = c.assemble_K(sig, False, idx=idx, alloc_extra_rows=5, to_host=True)  # the Nystroem call shape
            assert Kx.shape == (n + 5, len(idx)) and np.array_equal(Kx[:n], Kc), 'extra rows ' + tag
        finally:
            c.close()


def test_column_modes_n24_p6_vs_reference():
    """_assemble_kernel_mat of the reference with an index list (partial blocks of most points, train.py:1376-1407) and
    with a slice of whole points (train.py:1357-1374) on N = 24, P = 6: sampled rows, Frobenius norms; through every
    assembly kernel that serves this shape (assemble_pts: whole-point ranges; assemble_perm: everything)."""
    from sgdml_amd import _lib

    g = load('cols_n24_p6')
    M, N = g['R_train'].shape[:2]
    sig = float(g['sig'])
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    rows, scale = g['rows'], float(g['K_absmax'])
    p0, p1 = [int(v) for v in g['points']]
    for opts in ({}, {'asm.pts': 0}, {'asm.pts': 0, 'asm.perm_level': 1}):
        c = _lib.Context()
        try:
            for k, v in opts.items():
                c.set_option(k, v)
            xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
            c.train_upload(xd, gd, tp)
            Ki = c.assemble_K(sig, False, idx=g['col_idxs'], to_host=True)
            assert np.abs(Ki[rows] - g['K_idx_sample']).max() <= 1e-12 * scale, opts
            assert abs(np.linalg.norm(Ki) - float(g['K_idx_fro'])) <= 1e-11 * float(g['K_idx_fro'])
            Kp = c.assemble_K(sig, False, points=(p0, p1), to_host=True)
            assert np.abs(Kp[rows] - g['K_pts_sample']).max() <= 1e-12 * scale, opts
            assert abs(np.linalg.norm(Kp) - float(g['K_pts_fro'])) <= 1e-11 * float(g['K_pts_fro'])
        finally:
            c.close()


def test_iterative_solver_with_permutation_group_vs_reference():
    """The reference's Iterative.solve on N = 12, P = 6, M = 200 (n = 7200), k = 12 inducing points it drew itself: our
    K_nm for its columns equals the one it assembled (1e-12), gdml_pcg on that preconditioner follows scipy's residual
    history (first 8 steps 1e-6, whole history within the drift of two correct PCG runs), converges in the same number
    of iterations (+-10 %), and both coefficient vectors predict alike; then the drop-in path end to end (GDMLTrain.train
    -> Iterative with its own leverage sampling under the reference's seed: the same inducing columns)."""
    from sgdml_amd import _lib
    from sgdml_amd.train import GDMLTrain
    from sgdml_amd.utils.desc import Desc

    g = load('pcg_n12_p6_m200')
    M, N = g['R_train'].shape[:2]
    sig, lam, y = float(g['sig']), float(g['lam']), g['y']
    idx = g['inducing_pts_idxs']
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    c = _lib.Context()
    try:
        xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
        c.train_upload(xd, gd, tp)
        K_nm = c.assemble_K(sig, False, idx=idx, to_host=True)
        assert np.abs(K_nm[g['K_nm_rows']] - g['K_nm_sample']).max() <= 1e-12 * float(g['K_nm_absmax'])
        assert abs(np.linalg.norm(K_nm) - float(g['K_nm_fro'])) <= 1e-11 * float(g['K_nm_fro'])
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        c.nystroem_factor(lam, idx)
        c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
        hist = []
        x, info, iters, resid = c.pcg(lam, False, y, rtol=1e-4, maxiter=20000,
                                      callback=lambda it, r, fetch_x: hist.append(r) or False)
        assert info == 0
        n_ref = int(g['n_iters'])
        assert abs(iters - n_ref) <= max(2, n_ref // 10), (iters, n_ref)
        ref, ours = g['resid_hist'], np.array(hist)
        # (the residual moves by 1e-4 of its norm per step at first; two fp64 evaluations of the P = 6 operator differ by 1e-6)
        np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-5)
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
    # ---- drop-in: the same seed draws the same inducing columns through our leverage-score estimate
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': g['z'], 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x',
        'sig': int(sig), 'lam': lam, 'use_E': True, 'use_E_cstr': False, 'use_sym': True, 'perms': g['perms'],
    }
    tr = GDMLTrain()
    try:
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = int(g['k'])
        np.random.seed(int(g['seed']))
        model = tr.train(task)
    finally:
        tr.__del__()
    assert np.array_equal(model['inducing_pts_idxs'], idx)
    assert model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
    assert abs(int(model['solver_iters']) - n_ref) <= max(2, n_ref // 10)
    from sgdml_amd.predict import GDMLPredict

    E, F = GDMLPredict(model).predict(g['R_test'].reshape(len(g['R_test']), -1))
    assert np.abs(F - g['F_test']).max() <= 5e-3 * np.abs(g['F_test']).max()
    # (the fixture's model came from create_model, c = 0: energies are not compared)


@pytest.mark.parametrize('n_atoms,n_train,n_query,n_perms', [(80, 5, 9, 1), (66, 4, 70, 6), (91, 3, 5, 1), (92, 3, 6, 1),
                                                             (128, 2, 4, 1), (129, 3, 1, 1), (140, 2, 5, 6), (200, 2, 3, 1)])
def test_predict_big_kernels_vs_oracle(n_atoms, n_train, n_query, n_perms):
    """predict_big_kernel<8> (1024 < D <= 4096: N = 46 ... 91) and <16> (D <= 8192: N = 92 ... 128) for batches below the
    GEMM pipeline: forces 1e-10 of the largest force, energies 1e-10 max(1, |E|) against the oracle (predict.py:84-245).
    Beyond 128 atoms (D > 8192) every batch -- a single query too -- takes the GEMM pipeline, whatever the option says."""
    from sgdml_amd import _lib

    N, M, B = n_atoms, n_train, n_query
    ds = orc.synth_dataset(N, M + B, seed=N, jitter=0.3)
    perms = group_perms(N, 'c3xc2' if n_perms == 6 else 'id')
    tp = orc.tril_perms_from_atom_perms(perms)
    xo, go = orc.desc_from_R(ds['R'][:M].reshape(M, -1))
    rs = np.random.RandomState(2)
    ja = orc.d_desc_dot_vec(go, rs.normal(size=(M, 3 * N)))
    xq, gq = orc.desc_from_R(ds['R'][M:].reshape(B, -1))
    sig = 25.0
    E_ref, F_ref = orc.predict_from_desc(xq, gq, xo, ja, tp, sig)
    c = _lib.Context()
    try:
        c.set_option('predict.mfma_wide', 0)  # stay on the register-tiled kernels whatever the batch
        c.predict_upload_model(xo, ja, tp, sig, None)
        E, F = c.predict(ds['R'][M:].reshape(B, -1))
    finally:
        c.close()
    assert np.abs(F - F_ref).max() <= 1e-10 * np.abs(F_ref).max()
    assert np.abs(E - E_ref).max() <= 1e-10 * max(1.0, np.abs(E_ref).max())


@pytest.mark.parametrize('N,M,kind,n_cols', [(42, 10, 'c3^3', 40), (100, 6, 'id', 30), (24, 12, 'c3xc2', 25), (65, 5, 'c3xc2', 20),
                                             (12, 40, 'c3xc2', 90), (42, 9, 'id', 3)])
def test_assemble_perm_compact_column_lists(N, M, kind, n_cols):
    """Sparse index lists -- a few columns per training point, what the iterative solver's leverage sampling asks for
    (iterative.py:372-379, :401-411): the general kernel packs the REQUESTED column atoms 64 to a strip (asm.perm_compact)
    instead of computing all 3N columns of every point it touches.  Against the oracle (1e-12 of max|K|), bit-identical to
    the dense strips, with and without energy-constraint rows / columns, with extra rows."""
    from sgdml_amd import _lib

    xo, go, tp, sig, Ko, KoE = _oracle_case_m(N, M, kind)
    n = M * 3 * N
    scale = np.abs(Ko).max()
    rng = np.random.default_rng(N + M)
    idx = np.sort(rng.choice(n, size=n_cols, replace=False))
    idxE = np.sort(np.concatenate([idx, n + rng.choice(M, size=min(3, M), replace=False)]))
    out = {}
    for compact in (1, 0):
        c = _lib.Context()
        try:
            c.set_option('asm.wave', 0)
            c.set_option('asm.strip', 0)
            c.set_option('asm.perm_compact', compact)
            c.train_upload(xo, go, tp)
            Kc = c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx), to_host=True)
            KcE = c.assemble_K(sig, True, idx=idxE, to_host=True)
            out[compact] = (Kc[:n], KcE)
        finally:
            c.close()
    assert np.abs(out[1][0] - Ko[:, idx]).max() <= 1e-12 * scale
    assert np.abs(out[1][1] - KoE[:, idxE]).max() <= 1e-12 * np.abs(KoE).max()
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])
