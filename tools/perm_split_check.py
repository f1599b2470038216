"""Round 4 (groundwork for the permutation-group assembly redesign): the FIXED-ENTRY SPLIT of the
perm-summed kernel block, restated in NumPy and checked against the oracle (train.py:165-232 is what both restate).

For a block (i, j):   K_ij = sum_p [ 5 b_p (J_i^T d_p) (J_j^p^T d_p)^T  -  (sig^2 + sig n_p) b_p  J_i^T J_j^p ],
d_p = x_i - x_j[tp_p],  J_j^p = J_j[tp_p].  Descriptor entries k with tp_p[k] = k for EVERY permutation of the group (both atoms
of the pair are fixed points of the group: F) contribute the same to every p:
    |d_p|^2      = |d[F]|^2          + |d_p[R]|^2
    J_i^T d_p    = J_i[F]^T d[F]     + J_i[R]^T d_p[R]
    J_j^p^T d_p  = J_j[F]^T d[F]     + J_j[tp_p[R]]^T d_p[R]
    J_i^T J_j^p  = J_i[F]^T J_j[F]   + J_i[R]^T J_j[tp_p[R]]          (weighted: (sum_p c2_p) x the first term)
so the F parts are computed once per (i, j) and only the R parts (entries touching a moved atom) per permutation.
  python tools/perm_split_check.py        -> max deviation from the oracle and the work ratio per shape"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gdml_oracle as orc  # noqa: E402  (checker)

SQRT5 = np.sqrt(5.0)


def fixed_entries(tril_perms):
    """Boolean mask over descriptor entries: True where every permutation maps the entry to itself."""
    return np.all(tril_perms == np.arange(tril_perms.shape[1])[None, :], axis=0)


def K_fixed_split(R_desc, R_d_desc, tril_perms, sig):
    """Un-negated K (3NM square, no energy constraints) through the split; returns (K, multiply-add counts (split, plain))."""
    M, D = R_desc.shape
    N = orc.n_atoms_from_dim_d(D)
    N3 = 3 * N
    P = tril_perms.shape[0]
    sig = float(sig)
    J = orc.d_desc_from_comp(R_d_desc)  # (M, D, 3N)
    F = fixed_entries(tril_perms)
    Rm = ~F
    nF, nR = int(F.sum()), int(Rm.sum())
    K = np.zeros((M * N3, M * N3))
    for j in range(M):
        xj, Jj = R_desc[j], J[j]
        JjF = Jj[F]  # (nF, 3N)
        for i in range(M):
            xi, Ji = R_desc[i], J[i]
            # ---- once per (i, j): the part over fixed entries
            dF = xi[F] - xj[F]
            s2F = dF @ dF
            uF = Ji[F].T @ dF  # J_i^T d   (3N)
            vF = JjF.T @ dF    # J_j^T d   (3N)
            GF = Ji[F].T @ JjF  # J_i^T J_j (3N, 3N)
            blk = np.zeros((N3, N3))
            c2sum = 0.0
            # ---- per permutation: entries touching a moved atom
            for p in range(P):
                tpR = tril_perms[p][Rm]
                dR = xi[Rm] - xj[tpR]
                nrm = SQRT5 * np.sqrt(s2F + dR @ dR)
                b = 5.0 * np.exp(-nrm / sig) / (3.0 * sig**4)
                c2 = (sig**2 + sig * nrm) * b
                u = uF + Ji[Rm].T @ dR
                v = vF + Jj[tpR].T @ dR
                blk += 5.0 * b * np.outer(u, v) - c2 * (Ji[Rm].T @ Jj[tpR])
                c2sum += c2
            blk -= c2sum * GF
            K[i * N3:(i + 1) * N3, j * N3:(j + 1) * N3] = blk
    # multiply-adds of the descriptor-entry sums per (i, j) (6 non-zeros per Jacobian row: u, v 6 each, J^T J 36, norm 1)
    per_entry = 1 + 6 + 6 + 36
    return K, (per_entry * (nF + P * nR), per_entry * P * D)


def check(N, M, perms, sig=20.0, seed=5):
    ds = orc.synth_dataset(N, M, seed=seed, jitter=0.25)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.asarray(perms))
    Ko = orc.assemble_K(xd, gd, orc.tril_perms_lin_from_tril_perms(tp), sig)
    Ks, (w_split, w_plain) = K_fixed_split(xd, gd, tp, sig)
    dev = np.abs(Ks - Ko).max() / np.abs(Ko).max()
    return dev, w_split / w_plain, int(fixed_entries(tp).sum()), tp.shape[1]


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from asm_perm_check import group_perms

    for N, M, kind in [(12, 4, 'c3xc2'), (24, 3, 'c3^3'), (42, 2, 'c3^3'), (9, 4, 'c3xc2'), (21, 3, 'c2xc2')]:
        perms = group_perms(N, kind)
        dev, ratio, nF, D = check(N, M, perms)
        print('N=%-3d P=%-2d %-6s: %3d of %3d descriptor entries fixed by the whole group; max |K_split - K_oracle| / max|K| = %.1e; '
              'entry-sum work %.2f of the plain loop' % (N, len(perms), kind, nF, D, dev, ratio), flush=True)
