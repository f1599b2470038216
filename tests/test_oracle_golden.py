"""Pin the NumPy oracle against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np

from oracle import gdml_oracle as orc
from tests._tol import solve_tol


def _lat(g):
    if 'lattice' in g:
        lat = g['lattice']
        return (lat, np.linalg.inv(lat))
    return None


def _tril_perms(g):
    return orc.tril_perms_from_lin(g['tril_perms_lin'], g['R_desc'].shape[1])


def test_desc(golden):
    g = golden
    M = g['R_train'].shape[0]
    xd, jd = orc.desc_from_R(g['R_train'].reshape(M, -1), _lat(g))
    np.testing.assert_allclose(xd, g['R_desc'], rtol=1e-13, atol=0)
    np.testing.assert_allclose(jd, g['R_d_desc'], rtol=1e-12, atol=1e-15)


def test_perm_linearisation(golden):
    g = golden
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    assert np.array_equal(orc.tril_perms_lin_from_tril_perms(tp), g['tril_perms_lin'])
    assert np.array_equal(_tril_perms(g), tp)


def test_K_full(golden):
    g = golden
    K = orc.assemble_K(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], g['sig'], bool(g['use_E_cstr']))
    scale = np.abs(g['K']).max()
    assert np.abs(K - g['K']).max() <= 1e-12 * scale


def test_K_columns(golden):
    g = golden
    n = g['K'].shape[0]
    ex = len(g['col_idxs'])
    Kc = orc.assemble_K(
        g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], g['sig'], bool(g['use_E_cstr']),
        col_idxs=g['col_idxs'], alloc_extra_rows=ex,
    )
    assert Kc.shape == (n + ex, ex)
    scale = np.abs(g['K']).max()
    assert np.abs(Kc[:n] - g['K_cols']).max() <= 1e-12 * scale
    Ks = orc.assemble_K(
        g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], g['sig'], bool(g['use_E_cstr']),
        col_idxs=np.s_[: g['K_slice'].shape[1]],
    )
    assert np.abs(Ks - g['K_slice']).max() <= 1e-12 * scale


def test_analytic_solve_residual(golden):
    g = golden
    alphas, _ = orc.analytic_solve(g['K'], g['y'], float(g['lam']))
    A = -g['K'] + float(g['lam']) * np.eye(g['K'].shape[0])
    r = A @ (-alphas) - g['y']
    assert np.linalg.norm(r) / np.linalg.norm(g['y']) < solve_tol(A, alphas, g['y'])
    # prediction parity with the reference's alphas (not elementwise alpha parity: cond ~ 1/lam)
    M = g['R_desc'].shape[0]
    tp = _tril_perms(g)
    use_E = bool(g['use_E_cstr'])
    aF = alphas[:-M] if use_E else alphas
    aE = alphas[-M:] if use_E else None
    raF = g['alphas'][:-M] if use_E else g['alphas']
    raE = g['alphas'][-M:] if use_E else None
    N3 = g['R_train'].shape[1] * 3
    JA = orc.d_desc_dot_vec(g['R_d_desc'], aF.reshape(M, N3))
    rJA = orc.d_desc_dot_vec(g['R_d_desc'], raF.reshape(M, N3))
    xq, jq = orc.desc_from_R(g['R_test'].reshape(len(g['R_test']), -1), _lat(g))
    E1, F1 = orc.predict_from_desc(xq, jq, g['R_desc'], JA, tp, g['sig'], aE)
    E2, F2 = orc.predict_from_desc(xq, jq, g['R_desc'], rJA, tp, g['sig'], raE)
    assert np.abs(F1 - F2).max() <= 1e-5 * np.abs(F2).max()


def _model(g):
    m = {
        'sig': g['sig'], 'R_desc': g['model_R_desc'], 'R_d_desc_alpha': g['model_R_d_desc_alpha'],
        'tril_perms_lin': g['tril_perms_lin'], 'c': float(g['model_c']), 'std': float(g['model_std']),
    }
    if 'model_alphas_E' in g:
        m['alphas_E'] = g['model_alphas_E']
    if 'lattice' in g:
        m['lattice'] = g['lattice']
    return m


def cancel_floor(g):
    """Round-off floor of the prediction sum: the model coefficients of an ill-conditioned
    fit (lam=1e-10) reach 1e9 and cancel down to O(1) forces, so two correct summation
    orders differ by ~eps * (size of the summands).  50 eps * std * max|J alpha| * 5/(3 sig^2)
    * sqrt(M P) * max|J_x|."""
    M = g['R_desc'].shape[0]
    P = g['perms'].shape[0]
    sig = float(g['sig'])
    return (50 * np.finfo(float).eps * float(g['model_std']) * np.abs(g['model_R_d_desc_alpha']).max()
            * 5.0 / (3 * sig**2) * np.sqrt(M * P) * max(1.0, np.abs(g['R_d_desc']).max()))


def test_predict(golden):
    g = golden
    fl = cancel_floor(g)
    E, F = orc.predict(_model(g), g['R_test'].reshape(len(g['R_test']), -1))
    assert np.abs(F - g['F_test']).max() <= 1e-10 * np.abs(g['F_test']).max() + fl
    assert np.abs(E - g['E_test']).max() <= 1e-10 * max(1.0, np.abs(g['E_test']).max()) + fl * float(g['sig'])
    E, F = orc.predict(_model(g), None, g['R_desc'], g['R_d_desc'])
    assert np.abs(F - g['F_train_pred']).max() <= 1e-10 * np.abs(g['F_train_pred']).max() + fl
    assert np.abs(E - g['E_train_pred']).max() <= 1e-10 * max(1.0, np.abs(g['E_train_pred']).max()) + fl * float(g['sig'])


def test_model_R_d_desc_alpha(golden):
    g = golden
    M = g['R_desc'].shape[0]
    use_E = bool(g['use_E_cstr'])
    aF = g['alphas'][:-M] if use_E else g['alphas']
    JA = orc.d_desc_dot_vec(g['R_d_desc'], aF.reshape(M, -1))
    np.testing.assert_allclose(JA, g['model_R_d_desc_alpha'], rtol=1e-12, atol=1e-14 * np.abs(JA).max())


def test_kernel_matvec(golden):
    g = golden
    Kv = orc.kernel_matvec(g['R_desc'], g['R_d_desc'], _tril_perms(g), g['sig'], float(g['lam']),
                           g['v'], bool(g['use_E_cstr']))
    assert np.abs(Kv - g['Kv']).max() <= 1e-11 * np.abs(g['Kv']).max()
    # and it is the same operator as the assembled matrix
    Kv2 = g['K'] @ g['v'] - float(g['lam']) * g['v']
    assert np.abs(Kv - Kv2).max() <= 1e-10 * np.abs(Kv2).max()


def test_nystroem_factor(golden):
    g = golden
    L = orc.nystroem_factor(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], g['sig'], float(g['lam']),
                            g['col_idxs'], bool(g['use_E_cstr']))
    # the factor itself is conditioning-sensitive; the preconditioner action L^T L is what matters
    P1 = L.T @ L
    P2 = g['L_inv_K_mn'].T @ g['L_inv_K_mn']
    assert np.abs(P1 - P2).max() <= 1e-6 * np.abs(P2).max()


def test_pcg_matches_direct(golden):
    g = golden
    lam = float(g['lam'])
    A = -g['K'] + lam * np.eye(g['K'].shape[0])
    L = orc.nystroem_factor(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], g['sig'], lam,
                            g['col_idxs'], bool(g['use_E_cstr']))
    x, info, iters, resid = orc.pcg(lambda v: A @ v, g['y'], M_mv=lambda v: -orc.precon_apply(L, lam, v),
                                    rtol=1e-6, maxiter=5000)
    assert info == 0
    assert np.linalg.norm(A @ x - g['y']) <= 1e-5 * np.linalg.norm(g['y'])


def test_periodic_fixture_without_roundoff_floor():
    """n10_p2_pbc (round 6, lam = 1e-4): the oracle's minimum-image descriptors and its predictions from the reference's
    model against the reference's outputs with NO cancellation floor (the floor is 1.3e-12 on this fixture)."""
    import os

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'n10_p2_pbc.npz')))
    assert cancel_floor(g) <= 2e-12
    M = g['R_train'].shape[0]
    xd, jd = orc.desc_from_R(g['R_train'].reshape(M, -1), _lat(g))
    xo, _ = orc.desc_from_R(g['R_train'].reshape(M, -1))
    assert np.mean(np.abs(xo - xd) > 1e-9) > 0.3  # periodic in earnest
    np.testing.assert_allclose(xd, g['R_desc'], rtol=1e-13, atol=0)
    E, F = orc.predict(_model(g), g['R_test'].reshape(len(g['R_test']), -1))
    assert np.abs(F - g['F_test']).max() <= 1e-10 * np.abs(g['F_test']).max()
    assert np.abs(E - g['E_test']).max() <= 1e-10 * max(1.0, np.abs(g['E_test']).max())
