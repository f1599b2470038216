"""Worker of tests/test_hip_scale.py::test_sharded_nystroem_pcg_with_energy_constraints (torch.distributed.run, gloo; `world`
processes share GPU 0, host-staged collectives): the row-sharded Nystroem factor, its leverage scores, one application of the
preconditioner, the query-sharded mat-vec and a PCG solve of the energy-constraint system of fixture ecstr_n9_p6_m40
(n = 3N M + M).  Every rank holds the force rows of its points followed by their energy rows.  Rank 0 writes the results."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ctx, g, idx, v):
    """The same calls on any context (sharded or not)."""
    M, N = g['R_train'].shape[:2]
    sig, lam = float(g['sig']), float(g['lam'])
    from oracle import gdml_oracle as orc  # index tables only

    xd, gd = ctx.desc_from_R(g['R_train'].reshape(M, -1), N)
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    ctx.train_upload(xd, gd, tp)
    ctx.assemble_K(sig, True, idx=idx, alloc_extra_rows=len(idx))
    rows_held = ctx.K_shape()[0]
    lev, _, info = ctx.nystroem_factor(lam, idx, want_factor=False, want_lev=True)
    z = ctx.precon_apply(lam, v)
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, sig, np.zeros(M))
    Kv = ctx.kernel_matvec(lam, True, v)
    seen = []

    def cb(it, res, fetch_x):
        if it == 5:
            seen.append(fetch_x())
        return False

    x, pinfo, iters, resid = ctx.pcg(lam, True, g['y'], rtol=1e-5, maxiter=3000, use_precon=True, callback=cb)
    x_w, _, iters_w, _ = ctx.pcg(lam, True, g['y'], x0=x, rtol=1e-5, maxiter=10, use_precon=True)  # warm start: done at once
    return dict(lev=lev, z=z, Kv=Kv, x=x, x5=seen[0], iters=iters, pinfo=pinfo, resid=resid, iters_w=iters_w,
                rows_held=rows_held, info=info)


def inputs(g):
    M, N = g['R_train'].shape[:2]
    n = M * (3 * N + 1)
    rs = np.random.RandomState(7)
    idx = np.sort(rs.choice(n, 300, replace=False))
    assert (idx >= M * 3 * N).sum() >= 2  # energy columns among the inducing columns
    return idx, rs.normal(size=n)


def main():
    out_path = sys.argv[1]
    import torch.distributed as dist

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from sgdml_amd import _lib
    from sgdml_amd.dist import init_comm_from_torch_distributed

    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'ecstr_n9_p6_m40.npz')))
    idx, v = inputs(g)
    ctx = _lib.Context(0)
    init_comm_from_torch_distributed(ctx, backend='host')
    r = run(ctx, g, idx, v)
    chk = [None] * world
    dist.all_gather_object(chk, (float(np.abs(r['x']).sum()), int(r['rows_held'])))
    assert len(set(c[0] for c in chk)) == 1, chk  # every rank holds the same solution
    if rank == 0:
        np.savez(out_path, rows=np.array([c[1] for c in chk]), **{k: np.asarray(val) for k, val in r.items()})
    ctx.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
