"""Round 4 (groundwork, CPU only): index arithmetic of the planned permutation-group assembly kernel (DESIGN.md section 8
item 2) emulated in NumPy and checked against the oracle -- the dense per-point tables, the per-permutation u / v / diagonal /
single terms in atom-pair form, and the outer products sum_p c1_p u_p v_p^T as v_mfma_f64_16x16x4 tiles with the accumulators
in MFMA C layout (the layouts the GEMM / rank64_update kernels of csrc/chol.hip use on the GPU).

Conventions (train.py:165-232, desc.py:193-205, 468-469, 531-539):
  pair(a, m) = descriptor index of the atom pair; x(a, m) = x[pair];   Gs(a, m) = d x(a, m) / d r_a = (+g if a < m else -g)[pair]
  permutation p acts on atoms as pi;  tp_p[pair(a, m)] = pair(pi a, pi m)
  d_p(a, m) = x_i(a, m) - x_j(pi a, pi m)
  u_p[a]   = sum_{m != a} d_p(a, m) Gs_i(a, m)                                   (J_i^T d_p, 3 components per atom)
  v_p[b]   = sum_{m != a} d_p(a, m) Gs_j(b, pi m),  a = pi^-1 b                  ((J_j^p)^T d_p)
  S_p[(a, c), (b, e)] = sum_{m != a} Gs_i(a, m)[c] Gs_j(b, pi m)[e]   if b == pi a     ("diagonal" term)
                      = Gs_i(a, pi^-1 b)[c] Gs_j(b, pi a)[e]           otherwise        (single term)
  K_ij = sum_p [ 5 b_p u_p v_p^T - (sig^2 + sig n_p) b_p S_p ],  n_p = sqrt(5) |d_p|,  b_p = 5 exp(-n_p / sig) / (3 sig^4)

MFMA v_mfma_f64_16x16x4 (one wavefront): A operand lane l = A[row l & 15][k l >> 4], B operand lane l = B[k l >> 4][col l & 15],
accumulator lane l, register r = C[row (l >> 4) + 4 r][col l & 15].
  python tools/perm_mfma_emulate.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gdml_oracle as orc  # noqa: E402  (checker)

SQRT5 = np.sqrt(5.0)


def dense_tables(x, g, N):
    """XF[a][m] = x(a, m) (0 on the diagonal), GS[a][m][:] = Gs(a, m)."""
    ii, jj = orc.tril_pairs(N)  # ii > jj
    XF = np.zeros((N, N))
    GS = np.zeros((N, N, 3))
    XF[ii, jj] = XF[jj, ii] = x
    GS[jj, ii] = g   # a = j_k (the smaller index): +g
    GS[ii, jj] = -g  # a = i_k: -g
    return XF, GS


def mfma_16x16x4(a_lane, b_lane, c_lane):
    """c_lane[l][r] += sum_k A[row][k] B[k][col] with the lane layouts above."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a_lane[l]
        B[l >> 4, l & 15] = b_lane[l]
    C = A @ B
    for l in range(64):
        for r in range(4):
            c_lane[l, r] += C[(l >> 4) + 4 * r, l & 15]


def block_emulated(xi, gi, xj, gj, perms, sig, strip_atoms):
    """K_ij[:, columns of the atoms in strip_atoms] through the planned data flow; returns (3N, 3 len(strip))."""
    N = perms.shape[1]
    N3, P = 3 * N, perms.shape[0]
    XFi, GSi = dense_tables(xi, gi, N)
    XFj, GSj = dense_tables(xj, gj, N)
    nb = len(strip_atoms)
    ncol = 3 * nb
    U = np.zeros((P, N3))      # c1_p u_p
    V = np.zeros((P, ncol))    # v_p on the strip's columns
    c2 = np.zeros(P)
    offd = ~np.eye(N, dtype=bool)
    for p in range(P):
        pi = perms[p]
        pinv = np.argsort(pi)
        Dp = XFi - XFj[np.ix_(pi, pi)]            # d_p(a, m)
        n2 = 0.5 * np.sum(Dp[offd] ** 2)          # every pair twice
        nrm = SQRT5 * np.sqrt(n2)
        b = 5.0 * np.exp(-nrm / sig) / (3.0 * sig**4)
        c2[p] = (sig**2 + sig * nrm) * b
        u = np.einsum('am,amc->ac', Dp, GSi)      # (N, 3); the diagonal of Dp is 0
        U[p] = 5.0 * b * u.reshape(-1)
        for t, bb in enumerate(strip_atoms):
            a = pinv[bb]
            V[p, 3 * t:3 * t + 3] = np.einsum('m,mc->c', Dp[a], GSj[bb][pi])  # sum_m d_p(a, m) Gs_j(b, pi m)
    # ---- outer products on emulated MFMA tiles: rows padded to 16, columns to 16, k = p padded to 4
    RT, CT, KS = (N3 + 15) // 16, (ncol + 15) // 16, (P + 3) // 4
    Up = np.zeros((4 * KS, 16 * RT))
    Vp = np.zeros((4 * KS, 16 * CT))
    Up[:P, :N3] = U
    Vp[:P, :ncol] = V
    acc = np.zeros((RT, CT, 64, 4))
    lanes = np.arange(64)
    for ti in range(RT):
        for tj in range(CT):
            for ks in range(KS):
                a_lane = Up[4 * ks + (lanes >> 4), 16 * ti + (lanes & 15)]
                b_lane = Vp[4 * ks + (lanes >> 4), 16 * tj + (lanes & 15)]
                mfma_16x16x4(a_lane, b_lane, acc[ti, tj])
    # ---- single / diagonal terms added in C layout, then the store mapping lane / register -> (row, column)
    out = np.zeros((N3, ncol))
    for ti in range(RT):
        for tj in range(CT):
            for l in range(64):
                for r in range(4):
                    row, col = 16 * ti + (l >> 4) + 4 * r, 16 * tj + (l & 15)
                    if row >= N3 or col >= ncol:
                        continue
                    a, c = divmod(row, 3)
                    t, e = divmod(col, 3)
                    bb = strip_atoms[t]
                    s = 0.0
                    for p in range(P):
                        pi = perms[p]
                        if pi[a] == bb:
                            s += c2[p] * sum(GSi[a, m, c] * GSj[bb, pi[m], e] for m in range(N) if m != a)
                        else:
                            m = int(np.argsort(pi)[bb])
                            s += c2[p] * GSi[a, m, c] * GSj[bb, pi[a], e]
                    out[row, col] = acc[ti, tj, l, r] - s
    return out


def check(N, perms, sig=20.0, seed=11, strip=None):
    ds = orc.synth_dataset(N, 2, seed=seed, jitter=0.25)
    xd, gd = orc.desc_from_R(ds['R'].reshape(2, -1))
    perms = np.asarray(perms)
    tp = orc.tril_perms_from_atom_perms(perms)
    Ko = orc.assemble_K(xd, gd, orc.tril_perms_lin_from_tril_perms(tp), sig)
    N3 = 3 * N
    strip = list(range(N)) if strip is None else list(strip)
    cols = np.concatenate([3 * b + np.arange(3) for b in strip])
    dev = 0.0
    for i in range(2):
        for j in range(2):
            blk = block_emulated(xd[i], gd[i], xd[j], gd[j], perms, sig, strip)
            ref = Ko[i * N3:(i + 1) * N3, j * N3:(j + 1) * N3][:, cols]
            dev = max(dev, np.abs(blk - ref).max())
    return dev / np.abs(Ko).max()


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from asm_perm_check import group_perms

    for N, kind, strip in [(7, 'id', None), (9, 'c3xc2', None), (12, 'c3xc2', [2, 3, 4, 5, 6, 11, 0])]:
        perms = group_perms(N, kind)
        print('N=%-3d P=%-2d %-6s strip=%s: max |K_emulated - K_oracle| / max|K| = %.1e' % (
            N, len(perms), kind, 'all atoms' if strip is None else strip, check(N, perms, strip=strip)), flush=True)
