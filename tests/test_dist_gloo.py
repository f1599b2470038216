"""Multi-process (world_size 2 and 3, gloo, CPU) tests of the sharded iterative solver's structure.

The GPU library issues RCCL collectives itself; what can be validated without GPUs is the
ALGORITHM it implements (csrc/cg.hip, csrc/comm.hip): contiguous point shards, row-sharded
Nystroem factor with an all-reduced K_mm and K_nm^T K_nm, row-sharded preconditioner
(all-reduce of an m-vector + all-gather), query-sharded mat-vec (all-gather) and replicated CG
scalars.  Here that algorithm runs with real multi-process collectives, the NumPy oracle standing
in for the per-rank kernels, and must reproduce the single-process result.  The unique-id exchange
used on the GPU path (sgdml_amd.dist.init_comm_from_torch_distributed) is exercised with a stub
context."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from oracle import gdml_oracle as orc  # noqa: E402
from sgdml_amd.dist import init_comm_from_torch_distributed, shard_range, sharded_vector_positions  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a))
    dist.all_reduce(t)
    return t.numpy()


def _allgather_chunks(local, chunk, world):
    """In-place all-gather of equal chunks (padded), like comm_allgather_inplace."""
    buf = np.zeros(chunk)
    buf[: local.size] = local
    outs = [torch.zeros(chunk, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(outs, torch.from_numpy(buf))
    return np.concatenate([o.numpy() for o in outs])


def _sharded_solve(rank, world, case, rtol):
    g = dict(np.load(os.path.join(GOLDEN, case + '.npz')))
    lam, sig = float(g['lam']), float(g['sig'])
    M, N = g['R_train'].shape[:2]
    N3 = 3 * N
    n = M * N3
    idx = g['col_idxs']
    m = len(idx)
    p0, p1, per = shard_range(rank, world, M)
    row0, n_loc, chunk = p0 * N3, (p1 - p0) * N3, per * N3
    tp = orc.tril_perms_from_lin(g['tril_perms_lin'], g['R_desc'].shape[1])

    # local rows of K_nm (the oracle builds the block and the shard is sliced: stands in for the
    # row-range assembly kernel)
    K_nm = orc.assemble_K(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], sig, False, col_idxs=idx)
    X = K_nm[row0:row0 + n_loc].copy()
    # K_mm = -K[idx]: owned rows contributed, others zero, all-reduce
    S = np.zeros((m, m))
    for q, gi in enumerate(idx):
        if row0 <= gi < row0 + n_loc:
            S[q] = -X[gi - row0]
    S = _allreduce(S)
    import scipy.linalg as sla

    L, lower = orc.cho_factor_stable(S, pre_reg=True)
    X = sla.solve_triangular(L, X.T, lower=lower, trans='T', check_finite=False).T
    inner = _allreduce(X.T @ X)
    inner[np.diag_indices_from(inner)] += lam
    L2, lower2 = orc.cho_factor_stable(inner, eps_mag_max=-14)
    X = sla.solve_triangular(L2, X.T, lower=lower2, trans='T', check_finite=False).T
    lev = _allgather_chunks(np.einsum('ij,ij->i', X, X), chunk, world)[:n]

    def precon(v):
        t = _allreduce(X.T @ v[row0:row0 + n_loc])
        loc = (X @ t - v[row0:row0 + n_loc]) / lam
        return _allgather_chunks(loc, chunk, world)[:n]

    K_rows = g['K'][row0:row0 + n_loc]

    def A_mv(v):  # A v = -(K v - lam v), rows of this rank then all-gather
        loc = -(K_rows @ v - lam * v[row0:row0 + n_loc])
        return _allgather_chunks(loc, chunk, world)[:n]

    x, info, iters, resid = orc.pcg(A_mv, g['y'], M_mv=precon, rtol=rtol, maxiter=5000)
    return x, info, iters, lev


def _worker(rank, world, port, case, rtol, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x, info, iters, lev = _sharded_solve(rank, world, case, rtol)

        class StubCtx:  # records what the id exchange hands to gdml_comm_init
            def comm_init(self, uid, r, w):
                self.args = (uid, r, w)

        from sgdml_amd import _lib

        _lib.Context.comm_unique_id = staticmethod(lambda: bytes([rank + 7]) * 128)
        stub = StubCtx()
        r, w = init_comm_from_torch_distributed(stub)
        assert (r, w) == (rank, world) and stub.args == (bytes([7]) * 128, rank, world)
        # host-staged backend: the callbacks handed to gdml_comm_init_host complete the collectives in place
        class HostStub:
            def comm_init_host(self, r_, w_, allreduce, allgather):
                self.r, self.w, self.allreduce, self.allgather = r_, w_, allreduce, allgather

        hs = HostStub()
        r, w = init_comm_from_torch_distributed(hs, backend='host')
        assert (r, w, hs.r, hs.w) == (rank, world, rank, world)
        buf = np.arange(6, dtype=np.float64) + 10.0 * rank
        hs.allreduce(buf)
        np.testing.assert_array_equal(buf, world * np.arange(6) + 10.0 * sum(range(world)))
        chunk = 4
        g_buf = np.full(chunk * world, -1.0)
        g_buf[rank * chunk:(rank + 1) * chunk] = rank + 0.25 * np.arange(chunk)
        hs.allgather(g_buf, chunk)
        np.testing.assert_array_equal(g_buf, np.concatenate([q + 0.25 * np.arange(chunk) for q in range(world)]))
        np.testing.assert_array_equal(hs._bcast(np.arange(5) + 100 * rank), np.arange(5))  # rank 0's draw everywhere
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), x=x, info=info, iters=iters, lev=lev)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case,world', [('n6_p1', 2), ('n5_p4', 2), ('n9_p1', 3)])
def test_sharded_pcg_matches_single_process(tmp_path, case, world):
    rtol = 1e-6
    mp.spawn(_worker, args=(world, _free_port(), case, rtol, str(tmp_path)), nprocs=world, join=True)
    g = dict(np.load(os.path.join(GOLDEN, case + '.npz')))
    lam = float(g['lam'])
    n = g['K'].shape[0]
    A = -g['K'] + lam * np.eye(n)
    # single-process reference of the same algorithm
    Lf = orc.nystroem_factor(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], float(g['sig']), lam, g['col_idxs'])
    x_ref, info_ref, iters_ref, _ = orc.pcg(lambda v: A @ v, g['y'], M_mv=lambda v: orc.precon_apply(Lf, lam, v),
                                            rtol=rtol, maxiter=5000)
    lev_ref = np.einsum('ij,ij->j', Lf, Lf)
    xs = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    for r in range(world):
        assert int(xs[r]['info']) == 0
        # every rank holds the same replicated solution
        np.testing.assert_array_equal(xs[r]['x'], xs[0]['x'])
        assert np.linalg.norm(A @ xs[r]['x'] - g['y']) <= 2e-6 * np.linalg.norm(g['y'])
        np.testing.assert_allclose(xs[r]['lev'], lev_ref, rtol=1e-5, atol=1e-8 * lev_ref.max())
    assert abs(int(xs[0]['iters']) - iters_ref) <= max(3, iters_ref // 5)


def _sharded_solve_E(rank, world, case, rtol, out_dir):
    """The row-sharded Nystroem / PCG algorithm WITH energy constraints (csrc/cg.hip since round 6): a rank holds the force rows
    of its points followed by their energy rows, the replicated vectors are rank-major (sharded_vector_positions = VecLayout::pos),
    every result is one all-gather of `chunk` entries, the CG runs over the padded vectors."""
    import scipy.linalg as sla

    g = dict(np.load(os.path.join(GOLDEN, case + '.npz')))
    lam, sig = float(g['lam']), float(g['sig'])
    M, N = g['R_train'].shape[:2]
    N3, n_ff = 3 * N, M * 3 * N
    n = n_ff + M
    idx = g['col_idxs']
    m = len(idx)
    p0, p1, per = shard_range(rank, world, M)
    pos, chunk, n_pad = sharded_vector_positions(world, M, N3, True)
    loc_ref = np.concatenate([np.arange(p0 * N3, p1 * N3), n_ff + np.arange(p0, p1)])  # my rows, reference indices
    row0, n_loc = rank * chunk, len(loc_ref)
    assert np.array_equal(pos[loc_ref], row0 + np.arange(n_loc))  # ... are one contiguous run of the device order
    to_dev = lambda v: np.bincount(pos, weights=v, minlength=n_pad).astype(np.float64)  # scatter; padding stays zero
    K_nm = orc.assemble_K(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], sig, True, col_idxs=idx)
    X = K_nm[loc_ref].copy()
    S = np.zeros((m, m))
    for q, gi in enumerate(idx):  # K_mm = -K[idx]: the owner of a row contributes it
        r = pos[gi] - row0
        if 0 <= r < n_loc:
            S[q] = -X[r]
    S = _allreduce(S)
    L, lower = orc.cho_factor_stable(S, pre_reg=True)
    X = sla.solve_triangular(L, X.T, lower=lower, trans='T', check_finite=False).T
    inner = _allreduce(X.T @ X)
    inner[np.diag_indices_from(inner)] += lam
    L2, lower2 = orc.cho_factor_stable(inner, eps_mag_max=-14)
    X = sla.solve_triangular(L2, X.T, lower=lower2, trans='T', check_finite=False).T
    lev = _allgather_chunks(np.einsum('ij,ij->i', X, X), chunk, world)[pos]  # back to the reference order

    def precon(v):  # device order in, device order out
        t = _allreduce(X.T @ v[row0:row0 + n_loc])
        return _allgather_chunks((X @ t - v[row0:row0 + n_loc]) / lam, chunk, world)

    K_rows = g['K'][loc_ref]

    def A_mv(v):  # the mat-vec brings the coefficients back to the reference order; its output is the rank's chunk
        loc = -(K_rows @ v[pos] - lam * v[row0:row0 + n_loc])
        return _allgather_chunks(loc, chunk, world)

    x, info, iters, resid = orc.pcg(A_mv, to_dev(g['y']), M_mv=precon, rtol=rtol, maxiter=5000)
    np.savez(os.path.join(out_dir, 'e%d.npz' % rank), x=x[pos], info=info, iters=iters, lev=lev, pad=np.abs(np.delete(x, pos)).sum())


def _worker_E(rank, world, port, case, rtol, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _sharded_solve_E(rank, world, case, rtol, out_dir)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_pcg_with_energy_constraints_matches_single_process(tmp_path, world):
    """Fixture n5_p2_ecstr (8 points: shards 4 + 4 and 3 + 3 + 2; the inducing columns include an energy column)."""
    case, rtol = 'n5_p2_ecstr', 1e-6
    mp.spawn(_worker_E, args=(world, _free_port(), case, rtol, str(tmp_path)), nprocs=world, join=True)
    g = dict(np.load(os.path.join(GOLDEN, case + '.npz')))
    lam = float(g['lam'])
    n = g['K'].shape[0]
    assert g['col_idxs'].max() >= n - g['R_train'].shape[0]
    A = -g['K'] + lam * np.eye(n)
    Lf = orc.nystroem_factor(g['R_desc'], g['R_d_desc'], g['tril_perms_lin'], float(g['sig']), lam, g['col_idxs'], use_E_cstr=True)
    x_ref, info_ref, iters_ref, _ = orc.pcg(lambda v: A @ v, g['y'], M_mv=lambda v: orc.precon_apply(Lf, lam, v),
                                            rtol=rtol, maxiter=5000)
    lev_ref = np.einsum('ij,ij->j', Lf, Lf)
    xs = [np.load(os.path.join(str(tmp_path), 'e%d.npz' % r)) for r in range(world)]
    for r in range(world):
        assert int(xs[r]['info']) == 0 and float(xs[r]['pad']) == 0.0  # the padding of the device order stays zero
        np.testing.assert_array_equal(xs[r]['x'], xs[0]['x'])
        assert np.linalg.norm(A @ xs[r]['x'] - g['y']) <= 2e-6 * np.linalg.norm(g['y'])
        np.testing.assert_allclose(xs[r]['lev'], lev_ref, rtol=1e-5, atol=1e-8 * lev_ref.max())
    assert abs(int(xs[0]['iters']) - iters_ref) <= max(3, iters_ref // 5)


def test_sharded_vector_positions_are_a_padded_permutation():
    for M, W, d in ((8, 2, 15), (8, 3, 15), (40, 3, 27), (5, 8, 6), (1, 2, 3)):
        for use_E in (False, True):
            pos, chunk, n_pad = sharded_vector_positions(W, M, d, use_E)
            assert len(pos) == M * d + (M if use_E else 0) and len(set(pos.tolist())) == len(pos)
            assert pos.min() >= 0 and pos.max() < n_pad == chunk * W
            for r in range(W):  # a rank's entries: one run at the start of its chunk, forces before energies
                a, b, per = shard_range(r, W, M)
                mine = np.concatenate([pos[a * d:b * d], pos[M * d + a:M * d + b] if use_E else pos[:0]])
                first = r * chunk if use_E else a * d
                assert np.array_equal(mine, first + np.arange(len(mine)))


def test_shard_range_covers_points():
    for M in (1, 7, 64, 1000):
        for W in (1, 2, 3, 8):
            seen = []
            for r in range(W):
                a, b, per = shard_range(r, W, M)
                assert 0 <= a <= b <= M and b - a <= per
                seen += list(range(a, b))
            assert seen == list(range(M))


def _pick_backend_worker(rank, world, port, ids, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sgdml_amd import _lib
        from sgdml_amd.dist import pick_backend

        _lib.device_pci_bus_id = lambda device: ids[rank]  # no GPU here: the bus ids a launcher layout would produce
        with open(os.path.join(out_dir, 'r%d' % rank), 'w') as f:
            f.write(pick_backend(0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ids,expect', [(['0000:05:00.0', '0000:15:00.0'], 'rccl'),  # one GPU each (also with private views)
                                        (['0000:05:00.0', '0000:05:00.0'], 'host'),  # ranks share a GPU: RCCL would refuse
                                        (['0000:05:00.0', None], 'host')])           # a rank cannot tell: stay functional
def test_pick_backend_by_physical_gpu(tmp_path, ids, expect):
    mp.spawn(_pick_backend_worker, args=(2, _free_port(), ids, str(tmp_path)), nprocs=2, join=True)
    assert [open(os.path.join(str(tmp_path), 'r%d' % r)).read() for r in range(2)] == [expect, expect]


def _probe_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import json

        from sgdml_amd.dist import probe_rccl

        ok, detail = probe_rccl(0, timeout=120)
        with open(os.path.join(out_dir, 'r%d' % rank), 'w') as f:
            json.dump({'ok': ok, 'detail': detail}, f)
    finally:
        dist.destroy_process_group()


def test_rccl_probe_reports_failure_instead_of_hanging(tmp_path):
    """Without a GPU the probe's child processes cannot bring RCCL up: every rank must get the same (False, reasons)
    answer -- the branch bench.py takes to fall back to the host-staged collectives."""
    import json

    from sgdml_amd import _lib

    try:
        if _lib.device_count() > 0:
            pytest.skip('a GPU is visible: the failing branch is not reachable (the working one is a gpu test)')
    except OSError:
        pytest.skip('libgdml_hip.so not built')
    mp.spawn(_probe_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = [json.load(open(os.path.join(str(tmp_path), 'r%d' % r))) for r in range(2)]
    assert got[0] == got[1] and got[0]['ok'] is False
    assert set(got[0]['detail']['errors']) == {'0', '1'}


# ---------------------------------------------------------------------------------------------------------------
# Distributed Cholesky (csrc/dist_chol.hip): the block-row-cyclic algorithm with real multi-process collectives,
# NumPy standing in for the per-rank kernels.


def _dist_chol_worker(rank, world, port, case, nb, out_dir):
    import scipy.linalg as sla

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        keep_E = case.endswith('+E')  # energy-constraint rows stay in the system (they fill the last row blocks)
        g = _load_system(case)
        lam = float(g['lam'])
        n = g['K'].shape[0] - (g['R_train'].shape[0] if bool(g['use_E_cstr']) and not keep_E else 0)
        A_full = -g['K'][:n, :n] + lam * np.eye(n)
        y = g['y'][:n]
        nblk = -(-n // nb)
        rows_of = lambda b: min(nb, n - b * nb)
        mine = [b for b in range(nblk) if b % world == rank]
        # local share: my row blocks (lower part only is ever read) + the replicated right-hand-side row
        Aloc = {b: np.tril(A_full)[b * nb:b * nb + rows_of(b)].copy() for b in mine}
        rhs = y.copy()
        for k in range(nblk):
            k0, w = k * nb, rows_of(k)
            Lkk = np.zeros((nb, nb))
            if k % world == rank:
                Lkk[:w, :w] = np.linalg.cholesky(np.tril(Aloc[k][:, k0:k0 + w]) + np.tril(Aloc[k][:, k0:k0 + w], -1).T)
                Aloc[k][:, k0:k0 + w] = Lkk[:w, :w]
            Lkk = _allreduce(Lkk)[:w, :w]  # broadcast of the factored block
            below = [b for b in mine if b > k]
            for b in below:  # row-local solve
                Aloc[b][:, k0:k0 + w] = sla.solve_triangular(Lkk, Aloc[b][:, k0:k0 + w].T, lower=True).T
            rhs[k0:k0 + w] = sla.solve_triangular(Lkk, rhs[k0:k0 + w], lower=True)
            t0 = k0 + w
            if t0 >= n:
                break
            # gather the panel rows of all ranks in global order
            P = np.zeros((n - t0, w))
            for b in below:
                P[b * nb - t0:b * nb - t0 + rows_of(b)] = Aloc[b][:, k0:k0 + w]
            P = _allreduce(P)  # (the library all-gathers equal chunks and unpacks; the sum of disjoint rows is the same)
            for b in below:  # trailing update of my rows: columns t0 .. end of the block's diagonal
                hi = b * nb + rows_of(b)
                Aloc[b][:, t0:hi] -= Aloc[b][:, k0:k0 + w] @ P[:hi - t0].T
            rhs[t0:] -= P @ rhs[k0:k0 + w]
        # backward substitution
        acc, x = np.zeros(n), np.zeros(n)
        for k in range(nblk - 1, -1, -1):
            k0, w = k * nb, rows_of(k)
            s_blk = _allreduce(acc[k0:k0 + w].copy())
            if k % world == rank:
                Lkk = np.tril(Aloc[k][:, k0:k0 + w])
                x[k0:k0 + w] = sla.solve_triangular(Lkk, rhs[k0:k0 + w] - s_blk, lower=True, trans='T')
                acc[:k0] += Aloc[k][:, :k0].T @ x[k0:k0 + w]
        x = _allreduce(x)
        np.savez(os.path.join(out_dir, 'c%d.npz' % rank), x=x)
    finally:
        dist.destroy_process_group()


def _load_system(case):
    """Fixture by name; 'name+E' = the same fixture with its energy-constraint rows kept.  A fixture without the full matrix
    (ecstr_n9_p6_m40 stores the energy rows only) gets it from the oracle, which the test below first pins on those rows."""
    g = dict(np.load(os.path.join(GOLDEN, case.replace('+E', '') + '.npz')))
    if 'K' not in g:
        M = g['R_train'].shape[0]
        xd, gd = orc.desc_from_R(g['R_train'].reshape(M, -1))
        tpl = orc.tril_perms_lin_from_tril_perms(orc.tril_perms_from_atom_perms(g['perms']))
        g['K'] = orc.assemble_K(xd, gd, tpl, float(g['sig']), use_E_cstr=True)
        g['use_E_cstr'] = np.bool_(True)
    return g


@pytest.mark.parametrize('case,world,nb', [('n6_p1', 2, 32), ('n9_p1', 3, 64), ('n5_p4', 2, 48), ('n5_p2_ecstr+E', 2, 32),
                                           ('ecstr_n9_p6_m40+E', 3, 128)])
def test_block_row_cyclic_cholesky_matches_direct_solve(tmp_path, case, world, nb):
    """'+E' cases (round 6): the M energy-constraint rows / columns of train.py:235-300 stay in the system -- the distributed
    Cholesky carries them in its last row blocks (csrc/dist_chol.hip, assemble_erows_cyclic_launch)."""
    mp.spawn(_dist_chol_worker, args=(world, _free_port(), case, nb, str(tmp_path)), nprocs=world, join=True)
    g = _load_system(case)
    lam = float(g['lam'])
    n = g['K'].shape[0]
    if case.endswith('+E'):
        assert n == g['R_train'].shape[0] * (3 * g['R_train'].shape[1] + 1)
        if 'K_E_rows' in g:  # the oracle's energy rows against the reference's (train.py:235-300)
            assert np.abs(g['K'][-g['K_E_rows'].shape[0]:] - g['K_E_rows']).max() <= 1e-13 * np.abs(g['K_E_rows']).max()
    A = -g['K'] + lam * np.eye(n)
    for r in range(world):
        x = np.load(os.path.join(str(tmp_path), 'c%d.npz' % r))['x']
        assert np.linalg.norm(A @ x - g['y']) <= 1e-9 * np.linalg.norm(g['y'])
