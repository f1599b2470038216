"""NumPy emulation of the index arithmetic of csrc/assemble_perm.hip (column-atom lanes, row atoms split over
wavefronts, permutation loop), checked against the oracle on the CPU.  Development aid: it pins the conventions
(perm vs inverse perm, dense-table layout, signs) before GPU time is spent.
    python tools/asm_perm_emulate.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gdml_oracle as orc


def pair_idx(a, b):
    hi, lo = max(a, b), min(a, b)
    return hi * (hi - 1) // 2 + lo


def dense_tables(x, g, N):
    """XF[m][b] = x[pair(b,m)], GD[m][b][al] = G(b,m)[al] = (r_m - r_b)/d^3 (assemble_wave.hip dense_tables_kernel)."""
    XF = np.zeros((N, N)); GD = np.zeros((N, N, 3))
    for m in range(N):
        for b in range(N):
            if m == b: continue
            k = pair_idx(b, m)
            XF[m, b] = x[k]
            GD[m, b] = (1.0 if b < m else -1.0) * g[k]
    return XF, GD


def atom_perm_from_tril(tp, N):
    """pi with tp[pair(a,m)] = pair(pi a, pi m) (desc.hip atom_perm_from_tril_perm)."""
    D = N * (N - 1) // 2
    ii, jj = np.tril_indices(N, -1)
    pi = np.zeros(N, dtype=int)
    if N == 2:
        return np.array([0, 1])
    for a in range(N):
        m1, m2 = (a + 1) % N, (a + 2) % N
        k1, k2 = tp[pair_idx(a, m1)], tp[pair_idx(a, m2)]
        s1, s2 = {ii[k1], jj[k1]}, {ii[k2], jj[k2]}
        (pi[a],) = tuple(s1 & s2)
    return pi


def block_emulated(XFi, GDi, XFj, GDj, perms, pinvs, sig, N):
    """K block (i, j), un-negated, by the kernel's decomposition."""
    N3 = 3 * N
    out = np.zeros((N3, N3))
    erow = np.zeros(N3)
    for p in range(len(perms)):
        perm, pinv = perms[p], pinvs[p]
        # V12: lane = row atom a
        v = np.zeros((N, 3)); nrm2 = 0.0
        for a in range(N):
            pa = perm[a]
            for m in range(N):
                pm = perm[m]
                d = XFi[m, a] - XFj[pm, pa]
                v[a] += d * GDi[m, a]
                nrm2 += d * d
        nrm = np.sqrt(5.0) * np.sqrt(0.5 * nrm2)
        ex = np.exp(-nrm / sig)
        bp = ex * 5.0 / (3.0 * sig ** 4)
        beta, c = 5.0 * bp, (sig * sig + sig * nrm) * bp
        # V3: lane = column atom b
        u = np.zeros((N, 3)); dg = np.zeros((N, 3, 3))
        for b in range(N):
            ap = pinv[b]
            for mp in range(N):
                mi = pinv[mp]
                d = XFi[mi, ap] - XFj[mp, b]
                rj = GDj[mp, b]
                gi = GDi[mi, ap]
                u[b] += d * rj
                dg[b] += np.outer(gi, rj)
        # O
        for a in range(N):
            pa = perm[a]
            for b in range(N):
                ap = pinv[b]
                gi = GDi[ap, a]          # G_i(a, a')
                gj = GDj[pa, b]          # G_j(b, pi a)
                blk = beta * np.outer(v[a], u[b])
                if a == ap:
                    blk -= c * dg[b]
                else:
                    blk -= c * np.outer(gi, gj)
                out[3 * a:3 * a + 3, 3 * b:3 * b + 3] += blk
        erow -= (5.0 / (3.0 * sig ** 3)) * (nrm + sig) * ex * u.reshape(-1)
    return out, erow


def main():
    rng = np.random.default_rng(0)
    for N, M, perm_list in [(5, 3, [[0, 1, 2, 3, 4], [1, 0, 2, 3, 4], [0, 1, 3, 4, 2], [1, 0, 4, 2, 3]]),
                            (6, 3, [[0, 1, 2, 3, 4, 5]]),
                            (7, 2, [[0, 1, 2, 3, 4, 5, 6], [2, 0, 1, 3, 4, 6, 5], [1, 2, 0, 3, 4, 5, 6]])]:
        R = rng.normal(size=(M, N, 3)) * 1.5
        x, g = orc.desc_from_R(R.reshape(M, -1))
        tp = orc.tril_perms_from_atom_perms(np.array(perm_list))
        sig = 7.0
        K = orc.assemble_K(x, g, orc.tril_perms_lin_from_tril_perms(tp), sig, use_E_cstr=True)
        perms = [atom_perm_from_tril(tp[p], N) for p in range(len(perm_list))]
        pinvs = [np.argsort(pp) for pp in perms]
        tabs = [dense_tables(x[i], g[i], N) for i in range(M)]
        err = 0.0
        N3 = 3 * N
        for i in range(M):
            for j in range(M):
                blk, erow = block_emulated(tabs[i][0], tabs[i][1], tabs[j][0], tabs[j][1], perms, pinvs, sig, N)
                err = max(err, np.abs(blk - K[i * N3:(i + 1) * N3, j * N3:(j + 1) * N3]).max())
                err = max(err, np.abs(erow - K[M * N3 + i, j * N3:(j + 1) * N3]).max())
        print('N=%d P=%d: max|dK| = %.3e (max|K| = %.3e)' % (N, len(perm_list), err, np.abs(K).max()))
        assert err < 1e-12 * np.abs(K).max()


if __name__ == '__main__':
    main()
