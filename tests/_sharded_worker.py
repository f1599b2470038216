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
    if backend == 'chan':
        # the PyTorch-free host group (sgdml_amd.hostchannel, rendezvous from RANK / WORLD_SIZE / MASTER_*): host-staged
        # collectives over it, no torch import in this process
        from sgdml_amd.dist import host_group

        hg = host_group()
        rank, world = hg.rank, hg.world
        all_gather, finish, backend = hg.allgather_obj, hg.close, 'host'
    else:
        import torch.distributed as dist

        dist.init_process_group('gloo')
        rank, world = dist.get_rank(), dist.get_world_size()

        def all_gather(obj):
            every = [None] * world
            dist.all_gather_object(every, obj)
            return every

        finish = dist.destroy_process_group
    os.environ['LOCAL_RANK'] = '0' if backend == 'host' else os.environ.get('LOCAL_RANK', '0')

    from sgdml_amd.train import GDMLTrain

    # solver 'ecstr' / 'ecstr_dist': energy constraints through the distributed Cholesky (round 6: the energy rows ride in
    # its last row blocks); 'ecstr_cg': through the sharded iterative solver; 'lu': what the sharded solvers do not carry, run
    # by every rank redundantly (parked communicator)
    fixture = {'ecstr': 'n5_p2_ecstr', 'ecstr_dist': 'ecstr_n9_p6_m40', 'ecstr_cg': 'n5_p2_ecstr',
               'lu': 'lu_branch'}.get(solver, 'pcg_n9_m400')
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', fixture + '.npz')))
    M, N = g['R_train'].shape[:2]
    task = {
        'type': 't', 'code_version': '1.0.3', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
        'z': g['z'] if 'z' in g else np.full(N, 6), 'R_train': g['R_train'], 'F_train': g['F_train'], 'E_train': g['E_train'],
        'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x',
        'sig': int(g['sig']), 'lam': float(g['lam']), 'use_E': True, 'use_E_cstr': solver in ('ecstr', 'ecstr_dist', 'ecstr_cg'),
        'use_sym': g['perms'].shape[0] > 1, 'perms': g['perms'],
    }
    if solver == 'ecstr_cg':
        # energy constraints through the ITERATIVE solver (row-sharded since round 6) with a checkpoint writer: the writer must be
        # the group's rank 0 only
        from sgdml_amd.solvers import iterative as it_mod

        class FakeClock(object):  # 60 s per timer call: a checkpoint is due every tenth iteration (iterative.py:675-680)
            t = 0.0

            def default_timer(self):
                self.t += 60.0
                return self.t

        it_mod.timeit = FakeClock()
        tr = GDMLTrain()
        tr._force_solver = 'cg'
        tr._force_n_inducing_pts = 1
        tr.init_distributed(backend=backend)
        np.random.seed(5 + rank)

        def writer(m):
            with open(out_path + '.ckpt.rank%d' % rank, 'a') as f:
                f.write('%d\n' % int(m['solver_iters']))

        model = tr.train(task, save_progr_callback=writer)
        assert tr._context().comm_info() == (rank, world)
        chk = all_gather(float(np.abs(model['alphas_F']).sum()))
        assert len(set(chk)) == 1, chk
        calls, _ = tr._context().comm_stats()
        if rank == 0:
            np.savez(out_path, iters=model['solver_iters'], alphas=model['alphas_F'], coll_calls=calls)
        tr.__del__()
        finish()
        return
    if solver in ('ecstr', 'ecstr_dist', 'lu'):
        from sgdml_amd.solvers.analytic import Analytic
        took, orig = [], Analytic.solve

        def spy(self, *a, **kw):
            out = orig(self, *a, **kw)
            took.append(bool(self.used_lu))
            return out

        Analytic.solve = spy
        tr = GDMLTrain()
        tr._force_solver = 'analytic'
        tr.init_distributed(backend=backend)
        if solver == 'ecstr_dist':
            tr._context().set_option('dist.nb', 128)  # n = 1120: nine row blocks, the energy rows in the ninth
        model = tr.train(task)
        assert tr._context().comm_info() == (rank, world)  # the communicator is back after the redundant solve
        chk = all_gather(float(np.abs(model['alphas_F']).sum()))
        assert len(set(chk)) == 1, chk
        calls, _ = tr._context().comm_stats()
        if rank == 0:  # what GDMLPredict reads from a model (predict.py:326-354)
            keep = {k: model[k] for k in ('z', 'R_desc', 'R_d_desc_alpha', 'sig', 'c', 'std', 'perms', 'tril_perms_lin')}
            if 'alphas_E' in model:
                keep['alphas_E'] = model['alphas_E']
            np.savez(out_path, used_lu=np.array(took), coll_calls=calls, alphas_F=model['alphas_F'], **keep)
        tr.__del__()
        finish()
        return
    task['inducing_pts_idxs'] = g['inducing_pts_idxs']
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
    chk = all_gather(float(np.abs(model['alphas_F']).sum()))
    assert len(set(chk)) == 1, chk
    if sys.argv[2] == 'chan':
        assert 'torch' not in sys.modules  # the whole sharded solve ran without PyTorch in the process
    tr.__del__()
    finish()


if __name__ == '__main__':
    main()
