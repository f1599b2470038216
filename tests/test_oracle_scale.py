"""Pin the oracle against the round-2 fixtures: outputs of the reference at BASELINE.json's configuration
shapes (tests/golden/make_golden_r2.py) -- configs[0] (N=9, P=6, M=200), configs[3] (N=42, P=27),
configs[4] (N=60), the reference's PCG residual history and its LU branch.  CPU only."""
import os

import numpy as np
import pytest

from oracle import gdml_oracle as orc
from _pcg_compare import assert_same_convergence

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def setup_case(g):
    M, N = g['R_train'].shape[:2]
    xd, gd = orc.desc_from_R(g['R_train'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    return M, N, xd, gd, tp, orc.tril_perms_lin_from_tril_perms(tp)


@pytest.mark.parametrize('name', ['cfg0_n9_p6', 'cfg3_n42_p27', 'cfg4_n60_p1', 'cfg1_n21_m100', 'cfg3_n42_p27_m60', 'n100_m3', 'n150_p2_m3'])
def test_K_samples_and_solve_at_config_shapes(name):
    g = load(name)
    M, N, xd, gd, tp, lin = setup_case(g)
    sig, lam = float(g['sig']), float(g['lam'])
    K = orc.assemble_K(xd, gd, lin, sig)
    assert K.shape == (3 * N * M, 3 * N * M)
    assert np.abs(K[np.ix_(g['rows'], g['cols'])] - g['K_sample']).max() <= 1e-12 * float(g['K_absmax'])
    assert abs(np.linalg.norm(K) - float(g['K_fro'])) <= 1e-11 * float(g['K_fro'])
    assert abs(np.abs(K).max() - float(g['K_absmax'])) <= 1e-13 * float(g['K_absmax'])
    # solve: residual of the reference's coefficients and of the oracle's, predictions of both
    A = -K + lam * np.eye(K.shape[0])
    y = g['y']
    assert np.linalg.norm(A @ (-g['alphas']) - y) <= 1e-7 * np.linalg.norm(y)
    al, used_lu = orc.analytic_solve(K, y, lam)
    assert not used_lu and int(g['used_lu']) == 0
    xq, gq = orc.desc_from_R(g['R_test'].reshape(len(g['R_test']), -1))
    for coeffs, tol in ((g['alphas'], 1e-9), (al, 2e-4)):  # own alphas: conditioning-limited (lam = 1e-10)
        JA = orc.d_desc_dot_vec(gd, coeffs.reshape(M, -1))
        E, F = orc.predict_from_desc(xq, gq, xd, JA, tp, sig)
        F, E = F * float(g['model_std']), E * float(g['model_std']) + float(g['model_c'])
        assert np.abs(F - g['F_test']).max() <= tol * np.abs(g['F_test']).max()
        assert np.abs(E - g['E_test']).max() <= tol * np.abs(g['E_test']).max()


def test_validation_errors_recipe_cfg0():
    """cli.py:1564-1605 on the reference's predictions of the 100 validation geometries."""
    g = load('cfg0_n9_p6')
    M, N, xd, gd, tp, lin = setup_case(g)
    JA = orc.d_desc_dot_vec(gd, g['alphas'].reshape(M, -1))
    nv = len(g['R_valid'])
    xq, gq = orc.desc_from_R(g['R_valid'].reshape(nv, -1))
    E, F = orc.predict_from_desc(xq, gq, xd, JA, tp, float(g['sig']))
    F, E = F * float(g['model_std']), E * float(g['model_std']) + float(g['model_c'])
    de, df = g['E_valid_ref'] - E, (g['F_valid_ref'].reshape(nv, -1) - F).ravel()
    errs = np.array([np.abs(de).mean(), np.sqrt((de**2).mean()), np.abs(df).mean(), np.sqrt((df**2).mean())])
    np.testing.assert_allclose(errs, g['valid_errors'], rtol=1e-7)


def test_pcg_history_matches_reference():
    """The oracle's PCG on the inducing columns the reference drew: the residual norm after every iteration
    follows scipy's cg inside Iterative.solve (iterative.py:740-752) -- same count, same history."""
    g = load('pcg_n9_m400')
    M, N, xd, gd, tp, lin = setup_case(g)
    sig, lam, y = float(g['sig']), float(g['lam']), g['y']
    fac = orc.nystroem_factor(xd, gd, lin, sig, lam, g['inducing_pts_idxs'])
    def A(v):
        return -orc.kernel_matvec(xd, gd, tp, sig, lam, v)

    # residual norms ||r_k||: the preconditioner is applied to r_k once per iteration
    r_hist = []
    x, info, iters, resid = orc.pcg(A, y, M_mv=lambda r: (r_hist.append(np.linalg.norm(r)), orc.precon_apply(fac, lam, r))[1],
                                    rtol=1e-4, maxiter=2000)
    ref_hist = g['resid_hist']
    assert info == 0
    # scipy calls the callback after the update: ref_hist[k] = ||r_{k+1}||; ours records ||r_k|| before iteration k
    ours = np.array(r_hist[1:] + [resid])
    assert abs(iters - int(g['n_iters'])) <= max(2, int(0.1 * int(g['n_iters'])))
    # the first steps agree to rounding; afterwards two correct PCG runs on a system with cond ~ 1e10 drift
    # apart (measured: <= 7 % pointwise, 145 vs 146 iterations), so the bound on the whole history is loose
    np.testing.assert_allclose(ours[:8], ref_hist[:8], rtol=1e-6)
    k = min(len(ours), len(ref_hist))
    np.testing.assert_allclose(ours[:k], ref_hist[:k], rtol=0.15)
    # converged solutions predict alike
    JA0 = orc.d_desc_dot_vec(gd, g['alphas'].reshape(M, -1))
    JA1 = orc.d_desc_dot_vec(gd, (-x).reshape(M, -1))
    xq, gq = orc.desc_from_R(g['R_test'].reshape(len(g['R_test']), -1))
    F0 = orc.predict_from_desc(xq, gq, xd, JA0, tp, sig)[1]
    F1 = orc.predict_from_desc(xq, gq, xd, JA1, tp, sig)[1]
    assert np.abs(F1 - F0).max() <= 5e-3 * np.abs(F0).max()


def test_lu_branch_fixture():
    """A system on which scipy's Cholesky fails: the reference took its LU branch (analytic.py:101-114); the
    oracle takes it as well and the predictions agree."""
    g = load('lu_branch')
    M, N, xd, gd, tp, lin = setup_case(g)
    sig, lam = float(g['sig']), float(g['lam'])
    K = orc.assemble_K(xd, gd, lin, sig)
    al, used_lu = orc.analytic_solve(K, g['y'], lam)
    assert used_lu
    xq, gq = orc.desc_from_R(g['R_test'].reshape(len(g['R_test']), -1))
    F = []
    for coeffs in (g['alphas'], al):
        JA = orc.d_desc_dot_vec(gd, coeffs.reshape(M, -1))
        F.append(orc.predict_from_desc(xq, gq, xd, JA, tp, sig)[1] * float(g['model_std']))
    assert np.abs(F[0] - g['F_test']).max() <= 1e-9 * np.abs(g['F_test']).max()
    assert np.abs(F[1] - g['F_test']).max() <= 1e-6 * np.abs(g['F_test']).max()


def test_restart_fixture_first_stage_history():
    """Round-3 fixture of the reference's restarting run (make_golden_r3.py: k = 1 inducing point, restart after 100
    stagnating steps): the oracle's PCG on the first stage's inducing columns reproduces the reference's residual
    history up to the restart, and the restart criterion of iterative.py:614-735 (efficiency <= 0 over a window of
    100 steps) evaluated on the REFERENCE's own history fires at step 100 and never in the second stage."""
    g = load('pcg_restart')
    M = int(g['n_train'])
    N = g['R_all'].shape[1]
    xd, gd = orc.desc_from_R(g['R_all'][:M].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    sig, lam = float(g['sig']), float(g['lam'])
    y = g['F_all'][:M].ravel() / np.std(g['F_all'][:M].ravel())
    assert abs(np.linalg.norm(y) - float(g['norm_y_train'])) <= 1e-9 * np.linalg.norm(y)
    fac = orc.nystroem_factor(xd, gd, lin, sig, lam, g['inducing_stage0'])
    r_hist = []
    orc.pcg(lambda v: -orc.kernel_matvec(xd, gd, tp, sig, lam, v), y,
            M_mv=lambda r: (r_hist.append(np.linalg.norm(r)), orc.precon_apply(fac, lam, r))[1], rtol=1e-4, maxiter=100)
    ref = g['resid_hist']
    np.testing.assert_allclose(np.array(r_hist[1:9]), ref[:8], rtol=1e-6)
    np.testing.assert_allclose(np.array(r_hist[1:100]), ref[:99], rtol=0.15)

    def eff_after(hist):  # iterative.py:640-655
        steps = np.diff(np.concatenate([[hist[0]], hist]))
        steps[0] = 0.0
        tot = np.abs(steps).sum()
        ratio = -steps.clip(max=0).sum() / tot if tot > 0 else 1.0
        return (int(100 * ratio) - 50) * 2

    starts = list(g['cg_starts'])
    assert starts == [0, 100]
    assert eff_after(ref[:100]) <= 0
    second = ref[100:]
    assert all(eff_after(second[k - 100:k]) > 0 for k in range(100, len(second), 50))


def test_column_modes_n24_p6_fixture():
    """Round 4: the reference's index-list and point-slice assembly on N = 24 with a 6-element group (make_golden_r4.py)."""
    g = load('cols_n24_p6')
    M, N, xd, gd, tp, lin = setup_case(g)
    sig, scale = float(g['sig']), float(g['K_absmax'])
    Ki = orc.assemble_K(xd, gd, lin, sig, col_idxs=g['col_idxs'])
    assert np.abs(Ki[g['rows']] - g['K_idx_sample']).max() <= 1e-12 * scale
    assert abs(np.linalg.norm(Ki) - float(g['K_idx_fro'])) <= 1e-11 * float(g['K_idx_fro'])
    p0, p1 = [int(v) for v in g['points']]
    Kp = orc.assemble_K(xd, gd, lin, sig, col_idxs=np.s_[p0 * 3 * N:p1 * 3 * N])
    assert np.abs(Kp[g['rows']] - g['K_pts_sample']).max() <= 1e-12 * scale
    assert abs(np.linalg.norm(Kp) - float(g['K_pts_fro'])) <= 1e-11 * float(g['K_pts_fro'])


def test_pcg_with_permutation_group_matches_reference():
    """Round 4: the reference's Iterative.solve with P = 6 (N = 12, M = 200, k = 40): the oracle reproduces the K_nm it
    assembled for its inducing columns and, preconditioned with them, its residual history and iteration count."""
    g = load('pcg_n12_p6_m200')
    M, N, xd, gd, tp, lin = setup_case(g)
    sig, lam, y, idx = float(g['sig']), float(g['lam']), g['y'], g['inducing_pts_idxs']
    K_nm = orc.assemble_K(xd, gd, lin, sig, col_idxs=idx)
    assert np.abs(K_nm[g['K_nm_rows']] - g['K_nm_sample']).max() <= 1e-12 * float(g['K_nm_absmax'])
    assert abs(np.linalg.norm(K_nm) - float(g['K_nm_fro'])) <= 1e-11 * float(g['K_nm_fro'])
    fac = orc.nystroem_factor(xd, gd, lin, sig, lam, idx)
    # the 523 operator applications run on the assembled matrix (0.03 s each instead of 0.14 s matrix-free: this test was a
    # quarter of the CPU suite); the matrix-free operator the reference iterates with (iterative.py:183-204) is checked against it
    K = orc.assemble_K(xd, gd, lin, sig)
    probe = np.random.RandomState(0).normal(size=K.shape[0])
    mf = orc.kernel_matvec(xd, gd, tp, sig, lam, probe)
    assert np.abs(mf - (K @ probe - lam * probe)).max() <= 1e-12 * np.abs(mf).max()
    r_hist = []
    x, info, iters, resid = orc.pcg(lambda v: -(K @ v - lam * v), y,
                                    M_mv=lambda r: (r_hist.append(np.linalg.norm(r)), orc.precon_apply(fac, lam, r))[1],
                                    rtol=1e-4, maxiter=5000)
    assert info == 0
    n_ref = int(g['n_iters'])
    assert abs(iters - n_ref) <= max(2, n_ref // 10), (iters, n_ref)
    ours, ref = np.array(r_hist[1:] + [resid]), g['resid_hist']
    np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-6)
    assert_same_convergence(ours, ref, np.linalg.norm(y))


def test_pcg_with_energy_constraints_matches_reference():
    """Round 6: the reference's Iterative.solve WITH energy constraints (fixture pcg_ecstr_n9_p6_m150, make_golden_r6.py: the
    (3N + 1) M system of train.py:235-300, N = 9, P = 6, M = 150, lam = 1e-8, k = 3 inducing points, four of its 81 inducing columns
    are energy columns, 74 iterations): the oracle reproduces the K_nm it assembled -- force AND energy rows -- and, preconditioned
    with it, its residual history and iteration count."""
    g = load('pcg_ecstr_n9_p6_m150')
    M, N, xd, gd, tp, lin = setup_case(g)
    sig, lam, y, idx = float(g['sig']), float(g['lam']), g['y'], g['inducing_pts_idxs']
    n_ff = M * 3 * N
    assert len(y) == n_ff + M and (idx >= n_ff).sum() == 4 and (g['K_nm_rows'] >= n_ff).sum() == 16
    K_nm = orc.assemble_K(xd, gd, lin, sig, True, col_idxs=idx)
    assert np.abs(K_nm[g['K_nm_rows']] - g['K_nm_sample']).max() <= 1e-12 * float(g['K_nm_absmax'])
    assert abs(np.linalg.norm(K_nm) - float(g['K_nm_fro'])) <= 1e-11 * float(g['K_nm_fro'])
    fac = orc.nystroem_factor(xd, gd, lin, sig, lam, idx, use_E_cstr=True)
    K = orc.assemble_K(xd, gd, lin, sig, True)
    probe = np.random.RandomState(0).normal(size=K.shape[0])
    mf = orc.kernel_matvec(xd, gd, tp, sig, lam, probe, True)
    assert np.abs(mf - (K @ probe - lam * probe)).max() <= 1e-12 * np.abs(mf).max()
    r_hist = []
    x, info, iters, resid = orc.pcg(lambda v: -(K @ v - lam * v), y,
                                    M_mv=lambda r: (r_hist.append(np.linalg.norm(r)), orc.precon_apply(fac, lam, r))[1],
                                    rtol=1e-4, maxiter=5000)
    assert info == 0
    n_ref = int(g['n_iters'])
    assert abs(iters - n_ref) <= max(2, n_ref // 10), (iters, n_ref)
    ours, ref = np.array(r_hist[1:] + [resid]), g['resid_hist']
    np.testing.assert_allclose(ours[:8], ref[:8], rtol=1e-6)
    assert_same_convergence(ours, ref, np.linalg.norm(y))


def test_fixed_entry_split_of_the_perm_summed_block_matches_the_oracle():
    """Groundwork for the permutation-group assembly redesign: descriptor entries fixed by every
    permutation of the group contribute once per (i, j), the others once per permutation -- same K to rounding
    (tools/perm_split_check.py; train.py:165-232)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from perm_split_check import check

    perms = np.array([[0, 1, 2, 3, 4, 5, 6, 7], [1, 2, 0, 3, 4, 5, 6, 7], [2, 0, 1, 3, 4, 5, 6, 7],
                      [0, 1, 2, 4, 3, 5, 6, 7], [1, 2, 0, 4, 3, 5, 6, 7], [2, 0, 1, 4, 3, 5, 6, 7]])
    dev, ratio, n_fixed, D = check(8, 3, perms)
    assert D == 28 and n_fixed == 4  # the pairs among atoms 5, 6, 7 and the swapped pair {3, 4}, which maps to itself
    assert dev <= 1e-14 and ratio < 1.0


def test_planned_perm_assembly_data_flow_matches_the_oracle():
    """Groundwork (tools/perm_mfma_emulate.py): the atom-pair forms of u_p, v_p, the diagonal and single terms of J_i^T P J_j, and
    the outer products accumulated as 16 x 16 x 4 MFMA tiles in C layout (lane / register -> row, column), emulated in NumPy for a
    6-element group on 8 atoms and a ragged strip of column atoms, against the oracle's K (train.py:165-232)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from perm_mfma_emulate import check

    perms = np.array([[0, 1, 2, 3, 4, 5, 6, 7], [1, 2, 0, 3, 4, 5, 6, 7], [2, 0, 1, 3, 4, 5, 6, 7],
                      [0, 1, 2, 4, 3, 5, 6, 7], [1, 2, 0, 4, 3, 5, 6, 7], [2, 0, 1, 4, 3, 5, 6, 7]])
    assert check(8, perms, strip=[7, 0, 3, 4, 1]) <= 1e-14


def test_perm2_kernel_data_flow_matches_the_oracle():
    """The arithmetic csrc/assemble_perm2.hip was written against (tools/perm2_emulate.py): atoms renumbered with the ones no
    permutation moves first, their mutual descriptor entries summed once per block, one fused pass per permutation over the
    entries that touch a moved atom (chunked rows, fixed-order partial sums), outer products per tile group, single terms per
    permutation only where a moved atom is involved -- and the variant that sums the single / diagonal terms of fixed atoms once
    per block through W[x][y] = sum of c_p over the permutations with pi_p x = y.  Against the oracle's K (train.py:165-232)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from perm2_emulate import check, plan

    # two rotors in the middle of a 20-atom molecule: 6 moved atoms, 14 fixed ones; 16 x 16 tile groups: (0, 0) is not all-fixed
    idt = np.arange(20)
    def rot(a):
        p = idt.copy(); p[[a, a + 1, a + 2]] = [a + 1, a + 2, a]; return p
    r1, r2 = rot(4), rot(11)
    perms = np.array([idt, r1, r1[r1], r2, r2[r1], r2[r1[r1]], r2[r2], r2[r2][r1], r2[r2][r1[r1]]])
    sigma, nF, permI, pinvI = plan(perms)
    assert nF == 14 and sorted(sigma[nF:]) == [4, 5, 6, 11, 12, 13]
    for kw in [{}, {'split': False}, {'nchk_e': 2}, {'post': True}, {'post': True, 'es': True, 'ed': True, 'n_extra': 3}]:
        assert check(20, perms, **kw) <= 1e-14, kw
