"""Where does the fp32 + Gram-correction preconditioner (pcg.precon_form = 3) lose accuracy on the dominant subspace?
Fixture pcg_n12_p6_m200.  The device build leaves the rounded factor X32 (as doubles) and L (K_nm^T K_nm + lam I = L L^T) in the
resident matrix; T0 is rebuilt from them in NumPy, and both operators are compared on vectors z = X32 a of the factor's column
space against the cancellation-free expression  P z = -X32 L_G^-T (L^-1 L^-T) L_G^T a.

    python tools/f32_diag.py > gpurun_out/f32_diag.txt
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _pcg_compare import crossings  # noqa: E402
from oracle import gdml_oracle as orc  # noqa: E402
from sgdml_amd import _lib  # noqa: E402


def main():
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'pcg_n12_p6_m200.npz'), allow_pickle=True))
    M, N = g['R_train'].shape[:2]
    sig, lam, y, idx = float(g['sig']), float(g['lam']), g['y'], g['inducing_pts_idxs']
    n, m = len(y), len(idx)
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    c = _lib.Context()
    xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
    c.train_upload(xd, gd, tp)
    K = c.assemble_K(sig, False, to_host=True)
    c.set_option('pcg.precon_form', 3)
    c.assemble_K(sig, False, idx=idx, alloc_extra_rows=m)
    _, _, info = c.nystroem_factor(lam, idx, want_lev=False)
    c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
    buf = c.K_to_host()
    Xt, L = buf[:n].astype(np.float32).astype(np.float64), np.tril(buf[n:n + m])  # the device's X32 = float(X), X resident in fp64
    Y = sla.solve_triangular(L, np.eye(m), lower=True)  # L^-1
    W = Y.T @ Y  # L^-T L^-1 ... careful: X = B L^-T, X^T X = L^-1 (L L^T - lam) L^-T = I - lam L^-1 L^-T
    Wc = Y @ Y.T  # L^-1 L^-T
    G0 = np.eye(m) - lam * Wc
    G = Xt.T @ Xt
    LG = np.linalg.cholesky(G)
    Li = sla.solve_triangular(LG, np.eye(m), lower=True)
    T0 = Li.T @ G0 @ Li
    print('info %d; cond(G) %.3f; min eig G0 %.3e; |G - G0| max %.2e' % (info, np.linalg.cond(G), np.linalg.eigvalsh(G0)[0], np.abs(G - G0).max()))
    P_np = lambda v: (Xt @ (T0 @ (Xt.T @ v)) - v) / lam
    P_gpu = lambda v: c.precon_apply(lam, v)
    rng = np.random.default_rng(0)
    # top eigen-directions of the inner matrix expressed through the factor: a = L_G^-T e (columns of the orthonormal basis)
    evals, evecs = np.linalg.eigh(Wc)  # small eigenvalues of Wc = large sigma
    for name, a_orth in (('random in range', rng.standard_normal(m)), ('top direction (largest sigma)', evecs[:, 0]),
                         ('10th direction', evecs[:, 9]), ('smallest-sigma direction', evecs[:, -1])):
        a = Li.T @ a_orth  # z = X32 L_G^-T a_orth = Q a_orth, |z| = |a_orth|
        z = Xt @ a
        truth = -Xt @ (Li.T @ (Wc @ a_orth))
        pn, pg = P_np(z), P_gpu(z)
        s = np.abs(truth).max()
        print('%-32s |truth| %.3e   NumPy T0: %.2e   GPU: %.2e   (relative to |truth|)' %
              (name, s, np.abs(pn - truth).max() / s, np.abs(pg - truth).max() / s))
    ny = np.linalg.norm(y)
    lv = (0.3, 0.1, 0.03, 0.01, 3e-3, 1e-3)
    def dev_loop():
        h = []
        x, inf, it, res = c.pcg(lam, False, y, rtol=1e-4, maxiter=1500, callback=lambda i, r, f: h.append(r) or False)
        print('%-56s iters %4d crossings %s' % ('device loop gdml_pcg, fp32 + Gram correction', it, crossings(np.array(h), ny, lv).tolist()), flush=True)

    dev_loop()
    for name, P in (('host loop, dense K, NumPy T0 on the GPU-built X32 / L', P_np), ('host loop, dense K, GPU operator', P_gpu)):
        h = []
        x, inf, it, res = orc.pcg(lambda v: -(K @ v - lam * v), y, M_mv=lambda r: (h.append(np.linalg.norm(r)), P(r))[1], rtol=1e-4, maxiter=1500)
        print('%-56s iters %4d crossings %s' % (name, it, crossings(np.array(h[1:] + [res]), ny, lv).tolist()), flush=True)
    c.close()


if __name__ == '__main__':
    main()
