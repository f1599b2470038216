"""Worker of tests/test_hip_scale.py::test_distributed_cholesky_processes_share_one_gpu (torch.distributed.run, gloo):
`world` processes share GPU 0; the system matrix is assembled block-row-cyclic over them and factored by
gdml_dist_chol_solve with host-staged collectives.  Rank 0 writes the coefficients."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, n_atoms, n_train, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    backend = sys.argv[5] if len(sys.argv) > 5 else 'host'  # 'rccl': one GPU per rank, collectives inside the library
    lookahead = int(sys.argv[6]) if len(sys.argv) > 6 else 0  # dist.lookahead: the three-stream schedule + second communicator
    import torch.distributed as dist

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle import gdml_oracle as orc  # data generator only
    from sgdml_amd import _lib
    from sgdml_amd.dist import init_comm_from_torch_distributed

    ds = orc.synth_dataset(n_atoms, n_train, seed=9, jitter=0.3)
    y = ds['F'].ravel() / np.std(ds['F'])
    ctx = _lib.Context(0 if backend == 'host' else int(os.environ.get('LOCAL_RANK', rank)))
    init_comm_from_torch_distributed(ctx, backend=backend)
    ctx.set_option('dist.nb', nb)
    ctx.set_option('dist.lookahead', lookahead)
    xd, gd = ctx.desc_from_R(ds['R'].reshape(n_train, -1), n_atoms)
    ctx.train_upload(xd, gd, np.arange(n_atoms * (n_atoms - 1) // 2, dtype=np.int64)[None])
    alphas = ctx.dist_chol_solve(20.0, 1e-10, y)
    calls, nbytes = ctx.comm_stats()
    held = ctx.mem_info()[0]
    chk = [None] * world
    dist.all_gather_object(chk, float(np.abs(alphas).sum()))
    assert len(set(chk)) == 1, chk  # every rank received the same solution
    if rank == 0:
        np.savez(out_path, alphas=alphas, y=y, R=ds['R'], coll_calls=calls, coll_bytes=nbytes, held=held)
    ctx.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
