"""Worker of tests/test_hip_scale.py::test_sharded_iterative_two_processes_one_gpu (launched with
torch.distributed.run, world_size ranks sharing GPU 0): trains the pcg_n9_m400 fixture through
GDMLTrain with the iterative solver SHARDED over the ranks (or, solver 'analytic', the distributed Cholesky) -- csrc/cg.hip, predict.hip and comm.hip
with host-staged (gloo) collectives -- and writes rank 0's result next to the output path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, backend = sys.argv[1], sys.argv[2]
    solver = sys.argv[3] if len(sys.argv) > 3 else 'cg'
    import torch.distributed as dist

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    os.environ['LOCAL_RANK'] = '0' if backend == 'host' else os.environ.get('LOCAL_RANK', '0')

    from sgdml_amd.train import GDMLTrain

    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'pcg_n9_m400.npz')))
    M, N = g['R_train'].shape[:2]
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': g['z'], 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x',
        'sig': int(g['sig']), 'lam': float(g['lam']), 'use_E': True, 'use_E_cstr': False, 'use_sym': False,
        'perms': g['perms'], 'inducing_pts_idxs': g['inducing_pts_idxs'],
    }
    # the memory model must pick the fixture's k so that the given inducing columns are used as they are
    k = len(g['inducing_pts_idxs']) // (3 * N)
    np.random.seed(100 + rank)  # deliberately different per rank: every draw has to come from rank 0
    tr = GDMLTrain()
    tr._force_solver = solver
    tr._force_n_inducing_pts = k
    r, w = tr.init_distributed(backend=backend)
    assert (r, w) == (rank, world)
    model = tr.train(task)
    calls, nbytes = tr._context().comm_stats()
    if rank == 0 and solver == 'cg':
        np.savez(out_path, alphas=model['alphas_F'], iters=model['solver_iters'], resid=model['solver_resid'],
                 c=model['c'], inducing=model['inducing_pts_idxs'], coll_calls=calls, coll_bytes=nbytes,
                 norm_y=model['norm_y_train'])
    elif rank == 0:
        np.savez(out_path, alphas=model['alphas_F'], c=model['c'], coll_calls=calls, solver=model['solver_name'])
    # a second, restart-free property: all ranks hold the same coefficients
    chk = [None] * world
    dist.all_gather_object(chk, float(np.abs(model['alphas_F']).sum()))
    assert len(set(chk)) == 1, chk
    tr.__del__()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
