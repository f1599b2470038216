"""
CPU oracle for the sGDML kernel linear-algebra hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy *restatement* (written from the math in SURVEY.md App. A, not
copied) of what the reference computes on this path.  It exists so that tests,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg can check / time
the HIP path against an independent implementation.  Nothing under ``sgdml_amd/``
imports it; the product path fails loudly when the HIP library is missing.

Parity pinning: the reference ships no golden vectors or tests, so this
oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (which imports the reference from a scratch
copy) and committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every function here against those fixtures.

Reference citations (relative to /root/reference/):
  desc_from_R           sgdml/utils/desc.py:44-77,80-110,166-239
  desc_perm             sgdml/utils/desc.py:509-539
  tril_perms_lin        sgdml/train.py:897-904
  d_desc_from_comp      sgdml/utils/desc.py:422-471
  d_desc_dot_vec        sgdml/utils/desc.py:368-385
  vec_dot_d_desc        sgdml/utils/desc.py:388-408
  assemble_K            sgdml/train.py:97-302 (worker), :1260-1535 (driver)
  analytic_solve        sgdml/solvers/analytic.py:49-151
  predict               sgdml/predict.py:84-245, :426-447, :1146-1294
  set_alphas            sgdml/predict.py:551-601
  nystroem_factor       sgdml/solvers/iterative.py:208-351, :414-471
  pcg                   sgdml/solvers/iterative.py:83-206, :740-752 (scipy cg semantics)
"""

import numpy as np
import scipy.linalg as sla

SQRT5 = np.sqrt(5.0)


# --------------------------------------------------------------------------- layout


def n_atoms_from_dim_d(dim_d):
    return int((1 + np.sqrt(8 * dim_d + 1)) / 2)


def tril_pairs(n_atoms):
    """(i_k, j_k) with i_k > j_k in np.tril_indices(N,-1) order (desc.py:264)."""
    return np.tril_indices(n_atoms, k=-1)


def desc_from_R(R, lat_and_inv=None):
    """R (M,3N) -> R_desc (M,D) = 1/|r_i-r_j|, R_d_desc (M,D,3) = (r_i-r_j)/d^3.

    With a lattice the pair differences are first wrapped to the minimum image:
    diff -= lat @ round(lat_inv @ diff)   (desc.py:44-77).
    """
    R = np.asarray(R, dtype=np.float64)
    if R.ndim == 1:
        R = R[None, :]
    M = R.shape[0]
    r = R.reshape(M, -1, 3)
    i, j = tril_pairs(r.shape[1])
    diff = r[:, i, :] - r[:, j, :]  # (M,D,3)
    if lat_and_inv is not None:
        lat, lat_inv = lat_and_inv
        c = np.einsum('ab,mdb->mda', lat_inv, diff)
        diff = diff - np.einsum('ab,mdb->mda', lat, np.around(c))
    dist = np.sqrt(np.sum(diff * diff, axis=-1))
    return 1.0 / dist, diff / (dist**3)[..., None]


def desc_perm(perm):
    """Atom permutation (N,) -> descriptor permutation (D,) (desc.py:509-539).

    tp[k] = index of the pair {perm[i_k], perm[j_k]}.
    """
    perm = np.asarray(perm)
    n = len(perm)
    i, j = tril_pairs(n)
    idx = np.zeros((n, n), dtype=np.int64)
    idx[i, j] = np.arange(len(i))
    idx = idx + idx.T
    return idx[perm[i], perm[j]]


def tril_perms_from_atom_perms(perms):
    return np.array([desc_perm(p) for p in np.atleast_2d(perms)], dtype=np.int64)


def tril_perms_lin_from_tril_perms(tril_perms):
    """(P,D) -> flat (P*D,), element [k*P+p] = tril_perms[p,k] + p*D (train.py:903-904)."""
    P, D = tril_perms.shape
    return (tril_perms + np.arange(P)[:, None] * D).flatten('F')


def tril_perms_from_lin(tril_perms_lin, dim_d):
    """Inverse of the above: recover (P,D)."""
    lin = np.asarray(tril_perms_lin, dtype=np.int64)
    P = lin.size // dim_d
    return lin.reshape(dim_d, P).T - np.arange(P)[:, None] * dim_d


def d_desc_from_comp(R_d_desc):
    """(M,D,3) -> full Jacobians (M,D,3N): atom j_k gets +v, atom i_k gets -v."""
    R_d_desc = np.asarray(R_d_desc)
    if R_d_desc.ndim == 2:
        R_d_desc = R_d_desc[None]
    M, D, _ = R_d_desc.shape
    N = n_atoms_from_dim_d(D)
    i, j = tril_pairs(N)
    out = np.zeros((M, D, N, 3))
    k = np.arange(D)
    out[:, k, j, :] = R_d_desc
    out[:, k, i, :] = -R_d_desc
    return out.reshape(M, D, 3 * N)


def d_desc_dot_vec(R_d_desc, vecs):
    """J v :  (M,D,3),(M,3N) -> (M,D);  (Jv)_k = g_k . (v[j_k] - v[i_k])."""
    R_d_desc = np.asarray(R_d_desc)
    if R_d_desc.ndim == 2:
        R_d_desc = R_d_desc[None]
    vecs = np.asarray(vecs)
    if vecs.ndim == 1:
        vecs = vecs[None]
    D = R_d_desc.shape[1]
    N = n_atoms_from_dim_d(D)
    i, j = tril_pairs(N)
    v = vecs.reshape(vecs.shape[0], N, 3)
    return np.sum(R_d_desc * (v[:, j, :] - v[:, i, :]), axis=-1)


def vec_dot_d_desc(R_d_desc, vecs):
    """J^T f : (M|1,D,3),(M|1,D) -> (M,3N)."""
    R_d_desc = np.asarray(R_d_desc)
    if R_d_desc.ndim == 2:
        R_d_desc = R_d_desc[None]
    vecs = np.asarray(vecs)
    if vecs.ndim == 1:
        vecs = vecs[None]
    D = R_d_desc.shape[1]
    N = n_atoms_from_dim_d(D)
    i, j = tril_pairs(N)
    w = R_d_desc * vecs[..., None]  # (M,D,3)
    M = w.shape[0]
    out = np.zeros((M, N, 3))
    for a in range(M):
        np.add.at(out[a], j, w[a])
        np.subtract.at(out[a], i, w[a])
    return out.reshape(M, 3 * N)


# --------------------------------------------------------------------------- kernel matrix


def _full_K(R_desc, R_d_desc, tril_perms, sig, use_E_cstr):
    """Un-negated full kernel matrix, (3NM [+M]) square."""
    M, D = R_desc.shape
    N = n_atoms_from_dim_d(D)
    dim_i = 3 * N
    P = tril_perms.shape[0]
    sig = float(sig)

    J = d_desc_from_comp(R_d_desc)  # (M,D,3N)
    Jt = np.ascontiguousarray(J.transpose(0, 2, 1))  # (M,3N,D)
    n_rows = M * dim_i + (M if use_E_cstr else 0)
    K = np.zeros((n_rows, n_rows))
    E_off = M * dim_i

    for j in range(M):
        xj_p = R_desc[j][tril_perms]  # (P,D)
        Jj_p = J[j][tril_perms]  # (P,D,3N)
        d = R_desc[:, None, :] - xj_p[None]  # (M,P,D)
        nrm = SQRT5 * np.sqrt(np.sum(d * d, axis=-1))  # (M,P)
        e = np.exp(-nrm / sig)
        b = 5.0 * e / (3.0 * sig**4)
        # (all contractions as batched matmul so that BLAS does the work, like the reference's np.dot)
        u = np.matmul(d.transpose(1, 0, 2), Jj_p).transpose(1, 0, 2)  # (M,P,3N): d_p^T J_j^p
        A = np.matmul((5.0 * b[..., None] * d).transpose(0, 2, 1), u)  # (M,D,3N)
        A -= np.tensordot((sig**2 + sig * nrm) * b, Jj_p, axes=([1], [0]))
        blk = np.matmul(Jt, A)  # (M,3N,3N) = J_i^T A
        K[:E_off, j * dim_i : (j + 1) * dim_i] = blk.reshape(M * dim_i, dim_i)

        if use_E_cstr:
            # row block: K[E_off+i, blk_j]  (train.py:235-248)
            w = 5.0 / (3.0 * sig**3) * (nrm + sig) * e  # (M,P)
            K[E_off:, j * dim_i : (j + 1) * dim_i] = -np.einsum(
                'mpd,pdc->mc', w[..., None] * d, Jj_p
            )
            # E column of point j  (train.py:250-300): roles swapped
            d2 = R_desc[j][None, None, :] - R_desc[:, tril_perms]  # (M,P,D): x_j - P_p x_i
            nrm2 = SQRT5 * np.sqrt(np.sum(d2 * d2, axis=-1))
            e2 = np.exp(-nrm2 / sig)
            w2 = 5.0 / (3.0 * sig**3) * (nrm2 + sig) * e2
            Ji_p = J[:, tril_perms, :]  # (M,P,D,3N)
            col = -np.einsum('mpd,mpdc->mc', w2[..., None] * d2, Ji_p)
            K[:E_off, E_off + j] = col.reshape(-1)
            K[E_off:, E_off + j] = -np.sum(
                (1.0 + (nrm2 / sig) * (1.0 + nrm2 / (3.0 * sig))) * e2, axis=-1
            )
    return K


def assemble_K(
    R_desc,
    R_d_desc,
    tril_perms_lin,
    sig,
    use_E_cstr=False,
    col_idxs=np.s_[:],
    alloc_extra_rows=0,
):
    """Same contract as GDMLTrain._assemble_kernel_mat (train.py:1260-1535).

    Returns the UN-negated (rows+alloc_extra_rows, n_cols) matrix; extra rows are zero
    here (uninitialised in the reference).
    """
    R_desc = np.asarray(R_desc, dtype=np.float64)
    R_d_desc = np.asarray(R_d_desc, dtype=np.float64)
    D = R_desc.shape[1]
    tril_perms = tril_perms_from_lin(tril_perms_lin, D)
    K = _full_K(R_desc, R_d_desc, tril_perms, sig, use_E_cstr)
    n_rows = K.shape[0]
    if isinstance(col_idxs, slice) and col_idxs == slice(None) and alloc_extra_rows == 0:
        return K  # every column, nothing appended: no gather copy (it was a sixth of the slowest CPU test)
    if isinstance(col_idxs, slice):
        cols = np.arange(n_rows)[col_idxs]
    else:
        cols = np.asarray(col_idxs)
        assert len(cols) == len(set(cols.tolist()))
        assert np.array_equal(cols, np.sort(cols))
    if len(cols) > n_rows:
        raise ValueError('Columns indexed beyond range.')
    out = np.zeros((n_rows + alloc_extra_rows, len(cols)))
    out[:n_rows] = K[:, cols]
    return out


# --------------------------------------------------------------------------- analytic solve


def analytic_solve(K_unneg, y, lam):
    """(-K + lam I) x = y ; alphas = -x  (analytic.py:65-99). Returns (alphas, used_lu)."""
    A = -np.array(K_unneg, dtype=np.float64)
    A[np.diag_indices_from(A)] += lam
    try:
        c = sla.cho_factor(A, overwrite_a=False, check_finite=False)
        return -sla.cho_solve(c, y, check_finite=False), False
    except np.linalg.LinAlgError:
        return -sla.solve(A, y, check_finite=False), True


# --------------------------------------------------------------------------- predict


def perm_tables(R_desc, R_d_desc_alpha, tril_perms):
    """Permuted training tables, row m*P+p, [.,k] = v[m, tril_perms[p,k]] (predict.py:426-441)."""
    M, D = R_desc.shape
    P = tril_perms.shape[0]
    Xp = R_desc[:, tril_perms].reshape(M * P, D)
    JAp = R_d_desc_alpha[:, tril_perms].reshape(M * P, D)
    return Xp, JAp


def predict_from_desc(
    r_desc, r_d_desc, R_desc, R_d_desc_alpha, tril_perms, sig, alphas_E=None, chunk=256
):
    """Unscaled [E', F] for queries given as descriptors (B,D),(B,D,3) (predict.py:168-245)."""
    sig = float(sig)
    Xp, JAp = perm_tables(R_desc, R_d_desc_alpha, tril_perms)
    P = tril_perms.shape[0]
    aE = None if alphas_E is None else np.repeat(np.asarray(alphas_E, dtype=np.float64), P)
    B, D = r_desc.shape
    N = n_atoms_from_dim_d(D)
    E = np.zeros(B)
    F = np.zeros((B, 3 * N))
    fact = 5.0 / (3.0 * sig**3)
    for s in range(0, B, chunk):
        x = r_desc[s : s + chunk]
        d = x[:, None, :] - Xp[None]  # (b,MP,D)
        nrm = SQRT5 * np.sqrt(np.sum(d * d, axis=-1))  # (b,MP)
        e = np.exp(-nrm / sig)
        b = fact * e
        a = np.einsum('bjd,jd->bj', d, JAp)
        Fx = (5.0 / sig) * np.einsum('bj,bjd->bd', a * b, d)
        b2 = b * (nrm + sig)
        Fx -= b2 @ JAp
        Ex = np.sum(a * b2, axis=1)
        if aE is not None:
            Fx += np.einsum('bj,bjd->bd', aE[None] * b2, d)
            Ex += ((1.0 + (nrm / sig) * (1.0 + nrm / (3.0 * sig))) * e) @ aE
        E[s : s + chunk] = Ex
        F[s : s + chunk] = vec_dot_d_desc(r_d_desc[s : s + chunk], Fx)
    return E, F


def predict(model, R=None, R_desc_cache=None, R_d_desc_cache=None):
    """GDMLPredict.predict(R) on a model dict (predict.py:1146-1294): returns (E, F)."""
    sig = model['sig']
    R_desc_train = np.ascontiguousarray(np.asarray(model['R_desc']).T)
    D = R_desc_train.shape[1]
    tril_perms = tril_perms_from_lin(model['tril_perms_lin'], D)
    lat_and_inv = None
    if 'lattice' in model:
        lat = np.asarray(model['lattice'])
        lat_and_inv = (lat, np.linalg.inv(lat))
    if R is None:
        r_desc, r_d_desc = R_desc_cache, R_d_desc_cache
    else:
        R = np.asarray(R, dtype=np.float64)
        if R.ndim == 1:
            R = R[None]
        r_desc, r_d_desc = desc_from_R(R.reshape(R.shape[0], -1), lat_and_inv)
    aE = model['alphas_E'] if 'alphas_E' in model else None
    E, F = predict_from_desc(
        r_desc, r_d_desc, R_desc_train, np.asarray(model['R_d_desc_alpha']), tril_perms, sig, aE
    )
    std = model['std'] if 'std' in model else 1.0
    return E * std + model['c'], F * std


def kernel_matvec(R_desc, R_d_desc, tril_perms, sig, lam, v, use_E_cstr=False):
    """K v - lam v via the prediction contraction (iterative.py:183-204)."""
    M, D = R_desc.shape
    v = np.asarray(v, dtype=np.float64)
    vF, vE = (v[:-M], v[-M:]) if use_E_cstr else (v, None)
    N = n_atoms_from_dim_d(D)
    JA = d_desc_dot_vec(R_d_desc, vF.reshape(M, 3 * N))
    E, F = predict_from_desc(R_desc, R_d_desc, R_desc, JA, tril_perms, sig, vE)
    out = np.hstack((F.ravel(), -E)) if use_E_cstr else F.ravel()
    return out - lam * v


# --------------------------------------------------------------------------- Nystroem / PCG


def cho_factor_stable(Mat, pre_reg=False, eps_mag_max=1):
    """Cholesky with escalating jitter (iterative.py:414-471). Returns (c, lower) or None."""
    eps = np.finfo(float).eps
    eps_mag = int(np.floor(np.log10(eps)))
    if pre_reg:
        Mat[np.diag_indices_from(Mat)] += eps
        eps_mag += 1
    for reg in 10.0 ** np.arange(eps_mag, eps_mag_max + 1):
        try:
            return sla.cho_factor(Mat, overwrite_a=False, check_finite=False)
        except np.linalg.LinAlgError:
            Mat[np.diag_indices_from(Mat)] += reg
    return None


def nystroem_factor(R_desc, R_d_desc, tril_perms_lin, sig, lam, col_idxs, use_E_cstr=False):
    """L^-1 K_mn (m x n) as in iterative.py:208-351 (no QR fallback here)."""
    K_nm = assemble_K(R_desc, R_d_desc, tril_perms_lin, sig, use_E_cstr, col_idxs=col_idxs)
    cols = np.asarray(col_idxs)
    K_mm = -K_nm[cols, :].copy()
    L_mm, lower = cho_factor_stable(K_mm, pre_reg=True)
    K_nm = sla.solve_triangular(L_mm, K_nm.T, lower=lower, trans='T', check_finite=False).T
    inner = K_nm.T @ K_nm
    inner[np.diag_indices_from(inner)] += lam
    L, lower = cho_factor_stable(inner, eps_mag_max=-14)
    K_nm = sla.solve_triangular(L, K_nm.T, lower=lower, trans='T', check_finite=False).T
    return np.ascontiguousarray(K_nm.T)


def precon_apply(L_inv_K_mn, lam, v):
    """P v = (L^T L v - v)/lam (iterative.py:120-140)."""
    return (L_inv_K_mn.T @ (L_inv_K_mn @ v) - v) / lam


def pcg(A_mv, b, x0=None, M_mv=None, rtol=1e-4, maxiter=1000, callback=None):
    """Preconditioned CG with scipy.sparse.linalg.cg semantics (rtol*||b||, atol=0).

    Returns (x, info, iters, resid_norm). info=0 converged, >0 = maxiter reached.
    """
    b = np.asarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
    bnrm = np.linalg.norm(b)
    if bnrm == 0:
        return np.zeros_like(b), 0, 0, 0.0
    atol = rtol * bnrm
    r = b - A_mv(x) if x.any() else b.copy()
    rho_prev, p = None, None
    for it in range(maxiter):
        rn = np.linalg.norm(r)
        if rn < atol:
            return x, 0, it, rn
        z = M_mv(r) if M_mv is not None else r
        rho = r @ z
        if it > 0:
            p = z + (rho / rho_prev) * p
        else:
            p = z.copy()
        q = A_mv(p)
        alpha = rho / (p @ q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho
        if callback is not None:
            callback(x)
    return x, maxiter, maxiter, np.linalg.norm(r)


# --------------------------------------------------------------------------- synthetic data


def synth_dataset(n_atoms, n_frames, seed=0, jitter=0.3, n_conformers=4, spacing=1.4):
    """Seeded synthetic geometries + analytic pair-potential labels.

    Base conformers are distinct subsets of a cubic grid (spacing 1.4), frames are
    conformer + N(0, jitter). E = sum_{i<j} 1/d_ij, F = -grad E.
    """
    rng = np.random.RandomState(seed)
    g = int(np.ceil(n_atoms ** (1.0 / 3.0))) + 1
    grid = np.array([[a, b, c] for a in range(g) for b in range(g) for c in range(g)], float) * spacing
    sel = rng.choice(len(grid), n_atoms, replace=False)
    base0 = grid[sel]
    bases = [base0]
    for _ in range(n_conformers - 1):
        bases.append(base0 + rng.normal(0, 0.25, base0.shape))
    bases = np.array(bases)
    which = rng.randint(0, len(bases), n_frames)
    R = bases[which] + rng.normal(0, jitter, (n_frames, n_atoms, 3))
    i, j = tril_pairs(n_atoms)
    diff = R[:, i, :] - R[:, j, :]
    dist = np.sqrt((diff**2).sum(-1))
    E = (1.0 / dist).sum(-1)
    g_pair = diff / (dist**3)[..., None]  # -dE/dr_i contribution = +diff/d^3 for atom i
    F = np.zeros_like(R)
    for m in range(n_frames):
        np.add.at(F[m], i, g_pair[m])
        np.subtract.at(F[m], j, g_pair[m])
    z = np.ones(n_atoms, dtype=np.int64) * 6
    return {'R': R, 'E': E, 'F': F, 'z': z}
