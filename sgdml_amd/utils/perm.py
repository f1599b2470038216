"""
Permutational-symmetry discovery, own implementation of the procedure of
sgdml/utils/perm.py:53-404.  The O(M^2) part -- one linear assignment problem per pair of geometries, 499 500 of them at the
1000 geometries the search is capped at: 18 s on the host at 21 atoms, 23 s in the reference with one process -- runs on the GPU
when a context is passed (csrc/perm_match.hip, a wavefront per pair; GDMLTrain.create_task always passes its own), and so do the
eigendecompositions in front of it (batched Jacobi, one workgroup per geometry); the distance matrices, the spanning tree and
the closure are host work.  The NumPy / SciPy form of the matching below is what the
CPU tests pin against the reference's output, and what the GPU tests pin the kernel against:

  1. ``bipartite_match``  (perm.py:53-255): for every pair (i, j) of geometries, the atom assignment that
     minimises  -|V_i| |V_j|^T  (V = eigenvectors of the distance matrix, ordered by decreasing
     eigenvalue) under a same-species constraint; an assignment is kept when permuting geometry i
     brings its distance matrix strictly closer to geometry j's.
  2. ``sync_perm_mat``    (perm.py:258-279): only the assignments on the edges of the minimum spanning
     tree of the pair costs are trusted (permutation synchronisation).
  3. ``complete_sym_group`` (perm.py:352-392): closure under composition, giving up beyond ``n_perms_max``
     elements; ``salvage_subgroup`` (perm.py:306-349) then drops candidates whose cycles overlap a larger
     cycle of another candidate and the closure is retried.

Differences in form (not in result): the N x N cost matrices of one row i are formed for all j > i in
one batched product, the symmetric eigensolver is used (the reference's general ``eig`` returns the same
vectors up to sign, and only |V| enters), and the closure works on a set of tuples breadth-first.  The
reference's hard-coded ``find_extra_perms``/``find_frag_perms`` experiments (perm.py:407-...) are not
reproduced.
"""
from functools import partial

import numpy as np
import scipy.optimize
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import minimum_spanning_tree

from .. import DONE, NOT_DONE


def _dist_matrices(R, lat_and_inv=None, chunk=32):
    """(M,N,N) interatomic distance matrices, minimum-image wrapped with a lattice (desc.py:44-110).  In chunks of geometries:
    the (M,N,N,3) temporaries of the one-shot form leave the caches at 100 atoms (2.4 s for 1000 geometries against 0.36 s,
    same bits)."""
    M, N = R.shape[:2]
    out = np.empty((M, N, N))
    for c in range(0, M, chunk):
        diff = R[c:c + chunk, :, None, :] - R[c:c + chunk, None, :, :]
        if lat_and_inv is not None:
            lat, lat_inv = lat_and_inv
            frac = np.einsum('ij,mabj->mabi', lat_inv, diff)
            diff = diff - np.einsum('ij,mabj->mabi', lat, np.rint(frac))
        np.sqrt((diff**2).sum(-1), out=out[c:c + chunk])
    return out


def bipartite_match(R, z, lat_and_inv=None, max_processes=None, callback=None, ctx=None):
    """Pairwise assignments.  Returns ({(i, j): perm}, sparse symmetric cost matrix).  ctx: a device context
    (sgdml_amd._lib.Context) -- the pairs are then matched by gdml_perm_match."""
    R = np.asarray(R, dtype=float)
    M, N = R.shape[:2]
    z = np.asarray(z)
    species_penalty = (z[:, None] != z[None, :]).astype(float)

    adj = _dist_matrices(R, lat_and_inv)
    if callback is not None:
        callback = partial(callback, disp_str='Bi-partite matching')

    if ctx is not None:  # eigenvectors (batched Jacobi) and the M (M - 1) / 2 assignment problems on the device
        _, species = np.unique(z, return_inverse=True)
        cost_ij, ij, pm = ctx.perm_match(None, adj, species)
        found = {(int(i), int(j)): p.astype(np.int64) for (i, j), p in zip(ij, pm)}
        if callback is not None:
            callback(M, M)
        sym = cost_ij + cost_ij.T
        np.fill_diagonal(sym, np.inf)
        return found, csr_matrix(sym)

    w, v = np.linalg.eigh(adj)                      # ascending eigenvalues
    absv = np.abs(v[:, :, ::-1])                     # columns by decreasing eigenvalue

    cost_ij = np.zeros((M, M))
    found = {}
    for i in range(M):
        if i + 1 < M:
            # cost[j, a, b] = -sum_k |V_i[a,k]| |V_j[b,k]|, species mixing pushed above every other entry
            c = -np.einsum('ak,jbk->jab', absv[i], absv[i + 1:])
            c += species_penalty[None] * np.abs(c).max(axis=(1, 2))[:, None, None]
            before = np.linalg.norm((adj[i][None] - adj[i + 1:]).reshape(M - i - 1, -1), axis=1)
            for jj, j in enumerate(range(i + 1, M)):
                _, perm = scipy.optimize.linear_sum_assignment(c[jj])
                after = np.linalg.norm(adj[i][np.ix_(perm, perm)] - adj[j])
                if after >= before[jj]:
                    cost_ij[i, j] = before[jj]
                else:
                    cost_ij[i, j] = after
                    if not np.isclose(before[jj], after):
                        found[i, j] = perm
        if callback is not None:
            callback(i, M)
    if callback is not None:
        callback(M, M)
    sym = cost_ij + cost_ij.T
    np.fill_diagonal(sym, np.inf)
    return found, csr_matrix(sym)


def sync_perm_mat(match_perms_all, match_cost, n_atoms, callback=None):
    """Candidates = identity + the assignments on the minimum spanning tree of the pair costs."""
    if callback is not None:
        callback = partial(callback, disp_str='Multi-partite matching (permutation synchronization)')
        callback(NOT_DONE)
    tree = minimum_spanning_tree(match_cost, overwrite=True)
    cands = {tuple(range(n_atoms))}
    for edge in zip(*tree.nonzero()):
        p = match_perms_all.get(edge)
        if p is not None:
            cands.add(tuple(int(a) for a in p))
    if callback is not None:
        callback(DONE)
    return np.array(sorted(cands), dtype=int)


def to_cycles(perm):
    """Disjoint cycles of a permutation (fixed points are 1-cycles)."""
    perm = list(perm)
    seen = [False] * len(perm)
    cycles = []
    for start in range(len(perm)):
        if seen[start]:
            continue
        cyc, a = [], start
        while not seen[a]:
            seen[a] = True
            a = perm[a]
            cyc.append(a)
        cycles.append(cyc)
    return cycles


def salvage_subgroup(perms):
    """Keep the candidates none of whose non-trivial cycles shares an atom with a LONGER cycle of any
    candidate (perm.py:306-349)."""
    perms = np.asarray(perms)
    long_cycles = [[set(c) for c in to_cycles(p) if len(c) > 1] for p in perms]
    everything = [c for cs in long_cycles for c in cs]
    keep = [
        k for k, cs in enumerate(long_cycles)
        if not any(len(c) < len(o) and not c.isdisjoint(o) for c in cs for o in everything)
    ]
    return perms[keep, :]


def complete_sym_group(perms, n_perms_max=None, disp_str='Permutation group completion', callback=None):
    """Closure of the candidate set under composition; None once n_perms_max elements are reached."""
    if callback is not None:
        callback = partial(callback, disp_str=disp_str)
        callback(NOT_DONE)
    perms = np.asarray(perms, dtype=int)
    order = [tuple(p) for p in perms]
    have = set(order)
    grew = True
    while grew:
        grew = False
        snapshot = list(order)
        for a in snapshot:
            pa = np.array(a)
            for b in snapshot:
                new = tuple(int(v) for v in pa[list(b)])
                if new not in have:
                    have.add(new)
                    order.append(new)
                    grew = True
                    if n_perms_max is not None and len(order) == n_perms_max:
                        if callback is not None:
                            callback(DONE, sec_disp_str='transitive closure has failed', done_with_warning=True)
                        return None
    if callback is not None:
        callback(DONE, sec_disp_str='found {:d} symmetries'.format(len(order)))
    return np.array(order, dtype=int)


def find_perms(R, z, lat_and_inv=None, callback=None, max_processes=None, ctx=None):
    """Permutation group (P,N) of the molecule sampled by the geometries R (M,N,3); identity first."""
    n_atoms = R.shape[1]
    pair_perms, cost = bipartite_match(R, z, lat_and_inv, max_processes, callback=callback, ctx=ctx)
    cands = sync_perm_mat(pair_perms, cost, n_atoms, callback=callback)
    group = complete_sym_group(cands, n_perms_max=100, callback=callback)
    if group is None:
        group = complete_sym_group(salvage_subgroup(cands), n_perms_max=100,
                                   disp_str='Closure disaster recovery', callback=callback)
    return group
