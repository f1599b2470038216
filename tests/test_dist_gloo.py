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
from sgdml_amd.dist import init_comm_from_torch_distributed, shard_range  # noqa: E402

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


def test_shard_range_covers_points():
    for M in (1, 7, 64, 1000):
        for W in (1, 2, 3, 8):
            seen = []
            for r in range(W):
                a, b, per = shard_range(r, W, M)
                assert 0 <= a <= b <= M and b - a <= per
                seen += list(range(a, b))
            assert seen == list(range(M))
