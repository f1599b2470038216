"""Round 5 (CPU only): the data flow of assemble_perm2_kernel (csrc/assemble_perm2.hip) restated in NumPy and checked against the
oracle -- the arithmetic the kernel was written against.

What the kernel does differently from assemble_perm_kernel (train.py:97-302, torchtools.py:110-392 are the reference):
  * atoms are renumbered so that the atoms no permutation of the group moves (F, "fixed") come first, the others (E) last;
  * per (i, j) block the contributions of descriptor entries between two fixed atoms do not depend on the permutation: they are
    summed ONCE (base values u0, v0, dg0, nn0), every permutation then only adds the entries that touch a moved atom;
  * per permutation one fused pass over ordered atom pairs (b, m') gives |d_p|^2, u_p (column side), v_p (row side) and the
    'diagonal' 3x3 terms dg_p;  long rows are cut into chunks whose partial sums are added in a fixed order;
  * the outer products sum_p beta_p v_p u_p^T run on v_mfma_f64_16x16x4 with tile rows / columns relabelled so that one lane
    owns whole 3x3 atom blocks: tile group (s, t) = 16 row atoms x 16 column atoms, its 9 tiles are the (al, be) components;
    lane l: column atom 16 t + (l & 15), row atoms 16 s + (l >> 4) + 4 r;
  * the single terms -c_p G_i(a, pi^-1 b) (x) G_j(b, pi a) are added per permutation only in tile groups that contain a moved atom;
    tile groups of fixed atoms only get them once with sum_p c_p;
  * post: everything a fixed atom takes part in summed over the permutations first (W[x][y] = sum of c_p with pi_p x = y, A1, B1);
    es: the moved x moved single terms once per block as four partial sums; ed: the diagonal terms of moved atoms summed per (row,
    column) pair over a group of 8 permutations; n_extra: fixed rows that ride in full-range tasks (their pairs leave the base norm).
  python tools/perm2_emulate.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import gdml_oracle as orc  # noqa: E402  (checker)
from perm_mfma_emulate import dense_tables  # noqa: E402

SQRT5 = np.sqrt(5.0)


def plan(perms):
    """Internal atom order (fixed atoms first), permutations in internal numbering."""
    P, N = perms.shape
    moved = (perms != np.arange(N)[None]).any(axis=0)
    sigma = np.concatenate([np.nonzero(~moved)[0], np.nonzero(moved)[0]])
    nF = int((~moved).sum())
    inv = np.argsort(sigma)
    permI = inv[perms[:, sigma]]          # permI[p][a] = sigma^-1 pi_p sigma a
    pinvI = np.argsort(permI, axis=1)
    assert (permI[:, :nF] == np.arange(nF)[None]).all()
    return sigma, nF, permI, pinvI


def block_perm2(xi, gi, xj, gj, perms, sig, nchk_e=4, split=True, post=False, es=False, ed=False, n_extra=0):
    P, N = perms.shape
    sigma, nF, permI, pinvI = plan(perms)
    if not split:
        nF_eff = 0
    else:
        nF_eff = nF
    XFi, GSi = dense_tables(xi, gi, N)
    XFj, GSj = dense_tables(xj, gj, N)
    # tables in internal order: T[a][m] = (x(a, m), G(a, m))
    XI, GI = XFi[np.ix_(sigma, sigma)], GSi[np.ix_(sigma, sigma)]
    XJ, GJ = XFj[np.ix_(sigma, sigma)], GSj[np.ix_(sigma, sigma)]
    # ---- base pass: pairs of fixed atoms (identity action)
    u0 = np.zeros((N, 3)); v0 = np.zeros((N, 3)); dg0 = np.zeros((N, 3, 3)); nn0 = 0.0
    nFb = nF_eff - n_extra   # the last n_extra fixed rows ride in full-range tasks: their pairs are counted there, not in nn0
    for b in range(nF_eff):
        for m in range(nF_eff):
            d = XI[b, m] - XJ[b, m]
            if b < nFb:
                nn0 += d * d
            u0[b] += d * GJ[b, m]
            v0[b] += d * GI[b, m]
            dg0[b] += np.outer(GI[b, m], GJ[b, m])
    acc = np.zeros((N, N, 3, 3))
    ctot = 0.0
    cns = []
    ED = {}
    NG = (N + 15) // 16
    for p in range(P):
        pin = pinvI[p]
        U = np.zeros((N, 3)); V = np.zeros((N, 3)); DG = np.zeros((N, 3, 3)); nn = nn0
        for b in range(N):
            ap = pin[b]
            fixed_row = b < nFb
            m0 = nF_eff if fixed_row else 0
            nchk = 1 if fixed_row else nchk_e
            cnt = N - m0
            per = (cnt + nchk - 1) // nchk
            pu = np.zeros((nchk, 3)); pv = np.zeros((nchk, 3)); pd = np.zeros((nchk, 3, 3)); pn = np.zeros(nchk)
            for c in range(nchk):
                for m in range(m0 + c * per, min(N, m0 + (c + 1) * per)):
                    mi = pin[m]
                    d = XI[ap, mi] - XJ[b, m]
                    pn[c] += d * d
                    pu[c] += d * GJ[b, m]
                    pv[c] += d * GI[ap, mi]
                    pd[c] += np.outer(GI[ap, mi], GJ[b, m])
            # xor-tree over the chunks (what the shuffles do)
            step = 1
            while step < nchk:
                for c in range(0, nchk, 2 * step):
                    pu[c] += pu[c + step]; pv[c] += pv[c + step]; pd[c] += pd[c + step]; pn[c] += pn[c + step]
                step *= 2
            nn += pn[0]
            U[b] = pu[0] + (u0[b] if fixed_row else 0.0)
            V[ap] = pv[0] + (v0[b] if fixed_row else 0.0)   # ap = b for fixed rows
            DG[b] = pd[0] + (dg0[b] if fixed_row else 0.0)
        nrm = SQRT5 * np.sqrt(0.5 * nn)
        bp = np.exp(-nrm / sig) * 5.0 / (3.0 * sig**4)
        beta, cn = 5.0 * bp, -(sig * sig + sig * nrm) * bp
        ctot += cn
        acc += beta * np.einsum('ac,be->abce', V, U)   # the MFMA part
        if post:
            cns.append(cn)
            # per permutation only the blocks of two MOVED atoms: single terms and their diagonal terms
            if not es:
                for a in range(nF_eff, N):
                    for b in range(nF_eff, N):
                        acc[a, b] += cn * np.outer(GI[a, pin[b]], GJ[b, permI[p][a]])
            if ed:   # summed per (row atom, column atom) pair over the group of 8 permutations, added one group late
                for b in range(nF_eff, N):
                    ED[(pin[b], b)] = ED.get((pin[b], b), 0.0) + cn * DG[b]
                if p % 8 == 7 or p == P - 1:
                    for (a, b), val in ED.items():
                        acc[a, b] += val
                    ED.clear()
            else:
                for b in range(nF_eff, N):
                    acc[pin[b], b] += cn * DG[b]
            continue
        for s in range(NG):
            for t in range(NG):
                if split and 16 * s + 15 < nF and 16 * t + 15 < nF:
                    continue  # tile group of fixed atoms only: once, below
                for a in range(16 * s, min(N, 16 * s + 16)):
                    for b in range(16 * t, min(N, 16 * t + 16)):
                        acc[a, b] += cn * np.outer(GI[a, pin[b]], GJ[b, permI[p][a]])
        for b in range(N):
            acc[pin[b], b] += cn * DG[b]
    for s in range(NG):
        for t in range(NG):
            if not post and split and 16 * s + 15 < nF and 16 * t + 15 < nF:
                for a in range(16 * s, 16 * s + 16):
                    for b in range(16 * t, 16 * t + 16):
                        acc[a, b] += ctot * np.outer(GI[a, b], GJ[b, a])
    if post:
        # once per block: W[x][y] = sum of cn_p over the permutations with pi_p x = y (x, y moved atoms)
        E = list(range(nF_eff, N))
        W = np.zeros((N, N))
        for p in range(P):
            for x in E:
                W[x, permI[p][x]] += cns[p]
        A1 = np.zeros((N, N, 3))   # A1(a, e) = sum_p cn_p G_i(a, pi_p^-1 e) = sum_e' G_i(a, e') W[e'][e]     (a fixed)
        B1 = np.zeros((N, N, 3))   # B1(b, e) = sum_p cn_p G_j(b, pi_p e)    = sum_e' G_j(b, e') W[e][e']     (b fixed)
        for a in range(nF_eff):
            for e in E:
                for e2 in E:
                    A1[a, e] += GI[a, e2] * W[e2, e]
                    B1[a, e] += GJ[a, e2] * W[e, e2]
        if es:   # single terms of moved x moved blocks: four partial sums over p = q, q + 4, ... added inside the quad
            for a in range(nF_eff, N):
                for b in range(nF_eff, N):
                    part = np.zeros((4, 3, 3))
                    for p in range(P):
                        part[p % 4] += cns[p] * np.outer(GI[a, pinvI[p][b]], GJ[b, permI[p][a]])
                    acc[a, b] += (part[0] + part[1]) + (part[2] + part[3])
        for a in range(N):
            for b in range(N):
                fa, fb = a < nF_eff, b < nF_eff
                if fa and fb:
                    if a != b:
                        acc[a, b] += ctot * np.outer(GI[a, b], GJ[b, a])
                    else:   # diagonal term of a fixed atom: base part for every permutation + the moved partners through A1
                        t = ctot * dg0[a]
                        for e in E:
                            t = t + np.outer(A1[a, e], GJ[a, e])
                        acc[a, a] += t
                elif fa and not fb:
                    acc[a, b] += np.outer(A1[a, b], GJ[b, a])
                elif fb and not fa:
                    acc[a, b] += np.outer(GI[a, b], B1[b, a])
    out = np.zeros((3 * N, 3 * N))
    for a in range(N):
        for b in range(N):
            out[3 * sigma[a]:3 * sigma[a] + 3, 3 * sigma[b]:3 * sigma[b] + 3] = acc[a, b]
    return out


def check(N, perms, sig=20.0, seed=11, **kw):
    ds = orc.synth_dataset(N, 2, seed=seed, jitter=0.25)
    xd, gd = orc.desc_from_R(ds['R'].reshape(2, -1))
    perms = np.asarray(perms)
    tp = orc.tril_perms_from_atom_perms(perms)
    Ko = orc.assemble_K(xd, gd, orc.tril_perms_lin_from_tril_perms(tp), sig)
    N3 = 3 * N
    dev = 0.0
    for i in range(2):
        for j in range(2):
            blk = block_perm2(xd[i], gd[i], xd[j], gd[j], perms, sig, **kw)
            dev = max(dev, np.abs(blk - Ko[i * N3:(i + 1) * N3, j * N3:(j + 1) * N3]).max())
    return dev / np.abs(Ko).max()


if __name__ == '__main__':
    from asm_perm_check import group_perms
    for N, kind in [(9, 'c3xc2'), (20, 'c3^3'), (26, 'c3xc2'), (36, 'c3^3')]:
        perms = group_perms(N, kind)
        # move the rotors into the middle of the molecule so that the renumbering is not the identity
        rng = np.random.default_rng(N)
        rel = rng.permutation(N)
        perms = np.argsort(rel)[perms[:, rel]]
        for kw in [{}, {'split': False}, {'nchk_e': 8}, {'post': True}, {'post': True, 'es': True, 'ed': True, 'n_extra': 1}]:
            print('N=%-3d P=%-2d %-6s %-18s max |K_emulated - K_oracle| / max|K| = %.1e' % (N, len(perms), kind, kw, check(N, perms, **kw)),
                  flush=True)
