"""
``GDMLTrain`` with the reference's public interface (sgdml/train.py:305-1646).  Orchestration
(task/model dictionaries, label normalisation, integration constant) is host glue; the numerics
(descriptors, kernel matrix, Cholesky / PCG, predictions) run in the HIP library.
"""
import hashlib
import logging
import timeit
from functools import partial

import numpy as np

from . import __version__, DONE, NOT_DONE
from . import _lib
from .predict import GDMLPredict
from .solvers.analytic import Analytic
from .utils import io
from .utils.desc import Desc

_instance_alive = False  # the reference allows one GDMLTrain per process (train.py:336-342)


class GDMLTrain(object):
    def __init__(self, max_memory=None, max_processes=None, use_torch=False):
        global _instance_alive
        if _instance_alive:
            raise Exception(
                'You can not create multiple instances of this class. Please reuse your first one.'
            )
        _instance_alive = True
        self._owns_singleton = True
        self.log = logging.getLogger(__name__)
        self._max_memory = max_memory  # [GB] soft limit on DEVICE memory for this backend
        self._max_processes = max_processes
        self._use_torch = use_torch  # accepted for API compatibility; the HIP backend always runs
        self._ctx = None
        self._force_solver = None  # testing hook: 'analytic' or 'cg' overrides the memory-based choice
        self._force_n_inducing_pts = None  # testing hook: inducing points of the iterative solver (else memory model)
        # Public switch: how the iterative solver sizes its preconditioner.  'cost' (default): the number of inducing points that
        # minimises predicted build + iteration time, never more than memory allows (solvers/iterative.py::cost_n_inducing_pts);
        # 'memory': the reference's rule -- as many as fit (iterative.py:498-503).
        self.inducing_pts_policy = 'cost'
        # Public switch: spend the np.random draws the reference's CPU path spends on its worker benchmark before the CG
        # loop (solvers/iterative.py::_spend_reference_benchmark_draw), so that a run under the same np.random.seed draws the
        # inducing columns of a freshly installed reference's NumPy path.  Off by default: the caller's global stream is
        # left alone, like the reference's own torch path does (README, "Things a user of the reference should know").
        self.emulate_reference_rng = False
        self._emulate_ref_rng = False  # rounds 3-4 name of the same switch (tests)

    def __del__(self):
        global _instance_alive
        if getattr(self, '_owns_singleton', False):  # an instance whose __init__ raised never owned it
            _instance_alive = False
        ctx = getattr(self, '_ctx', None)
        if ctx is not None:
            ctx.close()

    def _context(self):
        if self._ctx is None:
            self._ctx = _lib.Context()
        return self._ctx

    def init_distributed(self, group=None, backend='rccl'):
        """Shard the solvers over the ranks of a one-process-per-GPU job: row-sharded Nystroem factor, query-sharded kernel
        mat-vec, block-row-cyclic Cholesky, RCCL all-reduce / all-gather inside the library (csrc/comm.hip).  `group`: the
        host-side group that ships the RCCL id and rank 0's decisions -- None = the PyTorch-free sgdml_amd.hostchannel built
        from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (or the default torch.distributed group when one is initialised),
        a HostChannel, or a torch.distributed process group.  backend='host' stages the collectives through the group
        instead (ranks sharing a GPU).  The reference's only multi-GPU path is nn.DataParallel (train.py:1463-1469).
        Returns (rank, world)."""
        from . import dist as _dist

        return _dist.init_comm(self._context(), group=group, backend=backend)

    def reserve_device_memory(self, gb=None):
        """Reserve the process-level device arena once and keep it (gdml_mem_reserve): the kernel matrices of every later
        train() call on this GPU are carved from it instead of being hipMalloc'ed and freed, which costs seconds per call
        beyond ~128 GB on this driver.  gb=None: 85 % of the HBM that is free now.  Returns the bytes held.  No counterpart
        in the reference (its matrices live in host RAM)."""
        return self._context().mem_reserve(None if gb is None else int(gb * 2**30))

    def _device_budget_bytes(self):
        ctx = self._context()
        _, free_b, total_b = ctx.mem_info()
        # the context keeps the previous kernel matrix resident for reuse: that memory is available to the next one
        budget = free_b + ctx.resident_K_bytes()
        if self._max_memory is not None:
            budget = min(budget, int(self._max_memory) * 1024**3)
        return budget

    # ------------------------------------------------------------------ task / model dictionaries

    def create_task(
        self,
        train_dataset,
        n_train,
        valid_dataset,
        n_valid,
        sig,
        lam=1e-10,
        perms=None,
        use_sym=True,
        use_E=True,
        use_E_cstr=False,
        callback=None,
    ):
        """Build a task dictionary (schema of train.py:507-524)."""
        if use_E and 'E' not in train_dataset:
            raise ValueError(
                'No energy labels found in dataset!\n'
                + 'By default, force fields are always reconstructed including the\n'
                + 'corresponding potential energy surface (this can be turned off).\n'
                + 'However, the energy labels are missing in the provided dataset.\n'
            )
        use_E_cstr = use_E and use_E_cstr

        if callback is not None:
            cb = partial(callback, disp_str='Hashing dataset(s)')
            cb(NOT_DONE)
        md5_train = io.dataset_md5(train_dataset)
        md5_valid = io.dataset_md5(valid_dataset)
        if callback is not None:
            cb(DONE)
            cb = partial(callback, disp_str='Sampling training and validation subsets')
            cb(NOT_DONE)

        if 'E' in train_dataset:
            idxs_train = self.draw_strat_sample(train_dataset['E'], n_train)
        else:
            idxs_train = np.random.choice(np.arange(train_dataset['F'].shape[0]), n_train, replace=False)
        excl_idxs = idxs_train if md5_train == md5_valid else np.array([], dtype=np.uint)
        if 'E' in valid_dataset:
            idxs_valid = self.draw_strat_sample(valid_dataset['E'], n_valid, excl_idxs=excl_idxs)
        else:
            cands = np.setdiff1d(np.arange(valid_dataset['F'].shape[0]), excl_idxs, assume_unique=True)
            idxs_valid = np.random.choice(cands, n_valid, replace=False)
        if callback is not None:
            cb(DONE)

        R_train = train_dataset['R'][idxs_train, :, :]
        task = {
            'type': 't',
            'code_version': __version__,
            'dataset_name': train_dataset['name'].astype(str),
            'dataset_theory': train_dataset['theory'].astype(str),
            'z': train_dataset['z'],
            'R_train': R_train,
            'F_train': train_dataset['F'][idxs_train, :, :],
            'idxs_train': idxs_train,
            'md5_train': md5_train,
            'idxs_valid': idxs_valid,
            'md5_valid': md5_valid,
            'sig': sig,
            'lam': lam,
            'use_E': use_E,
            'use_E_cstr': use_E_cstr,
            'use_sym': use_sym,
        }
        if use_E:
            task['E_train'] = train_dataset['E'][idxs_train]
        if 'lattice' in train_dataset:
            task['lattice'] = train_dataset['lattice']
            try:
                np.linalg.inv(task['lattice'])
            except np.linalg.LinAlgError:
                raise ValueError(
                    'Provided dataset contains invalid lattice vectors (not invertible). Note: Only rank 3 lattice vector matrices are supported.'
                )
        if 'r_unit' in train_dataset and 'e_unit' in train_dataset:
            task['r_unit'] = train_dataset['r_unit']
            task['e_unit'] = train_dataset['e_unit']

        n_atoms = train_dataset['R'].shape[1]
        if use_sym:
            if perms is None:
                if 'perms' in train_dataset:
                    self.log.info(
                        'Using {:d} permutations included in dataset.'.format(train_dataset['perms'].shape[0])
                    )
                    task['perms'] = train_dataset['perms']
                else:
                    # Symmetry discovery (train.py:560-584): utils/perm.py, the pairwise matching on the GPU (perm_match.hip)
                    from .utils import perm as perm_mod

                    lat_and_inv = None
                    if 'lattice' in task:
                        lat_and_inv = (task['lattice'], np.linalg.inv(task['lattice']))
                    n_train_pts = R_train.shape[0]
                    R_sync = R_train
                    if n_train_pts > 1000:  # train.py:565-573: at most 1000 geometries enter the matching
                        R_sync = R_train[np.random.choice(n_train_pts, 1000, replace=False), :, :]
                        self.log.info(
                            'Symmetry search has been restricted to a random subset of 1000/{:d} training points '
                            'for faster convergence.'.format(n_train_pts)
                        )
                    task['perms'] = perm_mod.find_perms(
                        R_sync, train_dataset['z'], lat_and_inv=lat_and_inv, callback=callback,
                        max_processes=self._max_processes, ctx=self._context(),
                    )
            else:
                n_perms, perms_len = perms.shape
                if perms_len != n_atoms:
                    raise ValueError('Provided permutations do not match the number of atoms in dataset.')
                self.log.info('Using {:d} externally provided permutations.'.format(n_perms))
                task['perms'] = perms
        else:
            task['perms'] = np.arange(n_atoms)[None, :]
        return task

    def create_task_from_model(self, model, dataset):
        """Task for resuming an unconverged model (schema of train.py:683-725)."""
        idxs_train = model['idxs_train']
        use_E = 'e_err' in model
        use_E_cstr = 'alphas_E' in model
        task = {
            'type': 't',
            'code_version': __version__,
            'dataset_name': model['dataset_name'],
            'dataset_theory': model['dataset_theory'],
            'z': model['z'],
            'R_train': dataset['R'][idxs_train, :, :],
            'F_train': dataset['F'][idxs_train, :, :],
            'idxs_train': idxs_train,
            'md5_train': model['md5_train'],
            'idxs_valid': model['idxs_valid'],
            'md5_valid': model['md5_valid'],
            'sig': model['sig'],
            'lam': model['lam'],
            'use_E': model['use_E'],
            'use_E_cstr': use_E_cstr,
            'use_sym': model['perms'].shape[0] > 1,
            'perms': model['perms'],
        }
        if use_E:
            task['E_train'] = dataset['E'][idxs_train]
        for src, dst in (('lattice', 'lattice'), ('alphas_F', 'alphas0_F'), ('alphas_E', 'alphas0_E'),
                         ('solver_iters', 'solver_iters'), ('inducing_pts_idxs', 'inducing_pts_idxs')):
            if src in model:
                task[dst] = model[src]
        if 'r_unit' in model and 'e_unit' in model:
            task['r_unit'] = model['r_unit']
            task['e_unit'] = model['e_unit']
        return task

    def create_model(self, task, solver, R_desc, R_d_desc, tril_perms_lin, std, alphas_F, alphas_E=None):
        """Model dictionary (schema of train.py:793-832)."""
        n_train, dim_d = R_d_desc.shape[:2]
        n_atoms = int((1 + np.sqrt(8 * dim_d + 1)) / 2)
        desc = Desc(n_atoms, max_processes=self._max_processes)
        R_d_desc_alpha = desc.d_desc_dot_vec(R_d_desc, np.asarray(alphas_F).reshape(-1, 3 * n_atoms))
        model = {
            'type': 'm',
            'code_version': __version__,
            'dataset_name': task['dataset_name'],
            'dataset_theory': task['dataset_theory'],
            'solver_name': solver,
            'z': task['z'],
            'idxs_train': task['idxs_train'],
            'md5_train': task['md5_train'],
            'idxs_valid': task['idxs_valid'],
            'md5_valid': task['md5_valid'],
            'n_test': 0,
            'md5_test': None,
            'f_err': {'mae': np.nan, 'rmse': np.nan},
            'R_desc': R_desc.T if R_desc.flags.writeable else R_desc.T.copy(),  # (cached descriptors are read-only)
            'R_d_desc_alpha': R_d_desc_alpha,
            'c': 0.0,
            'std': std,
            'sig': task['sig'],
            'lam': task['lam'],
            'alphas_F': alphas_F,
            'perms': task['perms'],
            'tril_perms_lin': tril_perms_lin,
            'use_E': task['use_E'],
        }
        if task['use_E']:
            model['e_err'] = {'mae': np.nan, 'rmse': np.nan}
            if task['use_E_cstr']:
                model['alphas_E'] = alphas_E
        if 'lattice' in task:
            model['lattice'] = task['lattice']
        if 'r_unit' in task and 'e_unit' in task:
            model['r_unit'] = task['r_unit']
            model['e_unit'] = task['e_unit']
        return model

    # ------------------------------------------------------------------ training

    def _train_descriptors(self, desc, R, lat_and_inv, callback):
        """Descriptors and Jacobians of the training geometries, reused across train() calls on the same geometries.

        sigma-grid reuse (the reference's `sgdml all` rebuilds them for every sigma, cli.py:993-1098): they do
        not depend on sigma, lam or the labels, so a hyper-parameter sweep computes them once.  Keyed by the CONTENT of the
        geometries (+ lattice): the same array modified in place is a different key.  One entry; the cached arrays are
        read-only because models keep views of them (`R_desc.T`).  What a sweep shares beyond this -- device-side tables,
        validation set, matrix buffer -- is content-hashed inside the library; the pairwise norms themselves are NOT cached:
        measured at < 3 % of a sweep (profiles/r03_sweep_kernel_stats.txt)."""
        key = None
        if R.nbytes <= (16 << 20):
            h = hashlib.blake2b(np.ascontiguousarray(R).tobytes(), digest_size=16)
            if lat_and_inv is not None:
                h.update(np.ascontiguousarray(lat_and_inv[0], dtype=np.float64).tobytes())
            key = (desc.n_atoms, R.shape, h.digest())
            hit = getattr(self, '_desc_cache', None)
            if hit is not None and hit[0] == key:
                if callback is not None:
                    callback(R.shape[0], R.shape[0], sec_disp_str='reused')
                return hit[1], hit[2]
        R_desc, R_d_desc = desc.from_R(R, lat_and_inv=lat_and_inv, callback=callback)
        if key is not None and R_d_desc.nbytes <= (1 << 30):
            R_desc.setflags(write=False)
            R_d_desc.setflags(write=False)
            self._desc_cache = (key, R_desc, R_d_desc)
        return R_desc, R_d_desc

    def train(self, task, save_progr_callback=None, callback=None):
        """Train a model from a task (train.py:836-1088)."""
        task = dict(task)
        n_train, n_atoms = task['R_train'].shape[:2]
        ctx = self._context()
        desc = Desc(n_atoms, max_processes=self._max_processes)
        desc._ctx = ctx

        tril_perms = np.array([Desc.perm(p) for p in task['perms']])
        n_perms = tril_perms.shape[0]
        tril_perms_lin = (tril_perms + np.arange(n_perms)[:, None] * desc.dim).flatten('F')  # train.py:903-904

        lat_and_inv = None
        if 'lattice' in task:
            try:
                lat_and_inv = (task['lattice'], np.linalg.inv(task['lattice']))
            except np.linalg.LinAlgError:
                raise ValueError(
                    'Provided dataset contains invalid lattice vectors (not invertible). Note: Only rank 3 lattice vector matrices are supported.'
                )

        R = task['R_train'].reshape(n_train, -1)
        R_desc, R_d_desc = self._train_descriptors(
            desc,
            R,
            lat_and_inv,
            partial(callback, disp_str='Generating descriptors and their Jacobians') if callback is not None else None,
        )

        # label vector (train.py:937-947)
        E_train_mean = None
        y = task['F_train'].ravel().copy()
        if task['use_E'] and task['use_E_cstr']:
            E_train = task['E_train'].ravel().copy()
            E_train_mean = np.mean(E_train)
            y = np.hstack((y, -E_train + E_train_mean))
        y_std = np.std(y)
        y /= y_std

        # solver choice: the reference compares 3 n^2 8 bytes with host RAM (train.py:949-964);
        # here the matrix lives in HBM and is factored in place, so the test is n^2 8 bytes vs HBM.
        budget = self._device_budget_bytes()
        est_analytic = Analytic.est_device_memory(n_train, n_atoms, task['use_E_cstr'])
        world = self._context().comm_info()[1]
        if world > 1:
            # distributed Cholesky: every rank holds 1/world of the matrix plus panel buffers
            n_sys = n_train * 3 * n_atoms + (n_train if task['use_E_cstr'] else 0)
            est_analytic = est_analytic / world + (2 * n_sys + 600000) * 512 * 8
        use_analytic_solver = est_analytic < 0.95 * budget
        if self._force_solver is not None:
            use_analytic_solver = self._force_solver == 'analytic'
        # free HBM differs between ranks (rank 0 usually holds more): every rank must take rank 0's branch, or one enters
        # the distributed Cholesky's collectives while another enters the sharded iterative solver's
        bcast = getattr(self._context(), '_bcast', None)
        if bcast is not None:
            use_analytic_solver = bool(np.asarray(bcast(np.array([int(use_analytic_solver)], dtype=np.int64)))[0])
        solver_keys = {}

        if use_analytic_solver:
            self.log.info('Using analytic solver (expected device memory use: ~{:.1f} GB)'.format(est_analytic / 2**30))
            analytic = Analytic(self, desc, callback=callback)
            alphas = analytic.solve(task, R_desc, R_d_desc, tril_perms_lin, y)
        else:
            from .solvers.iterative import Iterative

            self.log.info('Using iterative solver')
            iterative = Iterative(self, desc, self._max_memory, self._max_processes, self._use_torch,
                                  callback=callback)
            (
                alphas,
                solver_keys['solver_tol'],
                solver_keys['solver_iters'],
                solver_keys['solver_resid'],
                train_rmse,
                solver_keys['inducing_pts_idxs'],
                is_conv,
            ) = iterative.solve(task, R_desc, R_d_desc, tril_perms_lin, y, y_std,
                                save_progr_callback=save_progr_callback)
            solver_keys['norm_y_train'] = np.linalg.norm(y)
            self._last_precon_form = getattr(iterative, 'precon_form', None)  # 'stored' / 'matrix-free' (diagnostics)
            if not is_conv:
                self.log.warning('Iterative solver did not converge!')

        alphas_E = None
        alphas_F = alphas
        if task['use_E_cstr']:
            alphas_E = alphas[-n_train:]
            alphas_F = alphas[:-n_train]

        model = self.create_model(
            task, 'analytic' if use_analytic_solver else 'cg', R_desc, R_d_desc, tril_perms_lin, y_std,
            alphas_F, alphas_E=alphas_E,
        )
        model.update(solver_keys)

        if model['use_E']:
            c = (
                self._recov_int_const(model, task, R_desc=R_desc, R_d_desc=R_d_desc)
                if E_train_mean is None
                else E_train_mean
            )
            model['c'] = c
        return model

    def _recov_int_const(self, model, task, R_desc=None, R_d_desc=None):
        """Integration constant + label sanity diagnostics (train.py:1090-1258)."""
        # the predictor borrows the trainer's context: the training set is resident there already (content-hashed upload),
        # and a context of its own would cost more than this whole prediction for a small system
        gdml_predict = GDMLPredict(model, max_memory=self._max_memory, max_processes=self._max_processes,
                                   log_level=logging.CRITICAL, _borrow_ctx=self._context())
        gdml_predict.set_R_desc(R_desc)
        gdml_predict.set_R_d_desc(R_d_desc)
        E_pred, _ = gdml_predict.predict()
        E_ref = np.squeeze(task['E_train'])

        e_fact = np.linalg.lstsq(np.column_stack((E_pred, np.ones(E_ref.shape))), E_ref, rcond=-1)[0][0]
        corrcoef = np.corrcoef(E_ref, E_pred)[0, 1]
        if np.sign(e_fact) == -1:
            self.log.warning(
                'It looks like the provided dataset may contain gradients instead of force labels (flipped sign).'
            )
        if corrcoef < 0.95:
            self.log.warning(
                'Potentially inconsistent energy labels detected! (correlation coefficient {:.2f})'.format(corrcoef)
            )
        if np.abs(e_fact - 1) > 1e-1:
            self.log.warning(
                'Potentially inconsistent scales in energy vs. force labels detected! (factor ~{:.2f})'.format(e_fact)
            )
        return np.sum(E_ref - E_pred) / E_ref.shape[0]

    # ------------------------------------------------------------------ kernel matrix (semi-public)

    def _assemble_kernel_mat(
        self,
        R_desc,
        R_d_desc,
        tril_perms_lin,
        sig,
        desc,
        use_E_cstr=False,
        col_idxs=np.s_[:],
        alloc_extra_rows=0,
        callback=None,
    ):
        """Host copy of the un-negated kernel matrix, contract of train.py:1260-1535."""
        n_train, dim_d = R_d_desc.shape[:2]
        dim_i = 3 * int((1 + np.sqrt(8 * dim_d + 1)) / 2)
        K_n_rows = n_train * dim_i + (n_train if use_E_cstr else 0)

        points = idx = None
        if isinstance(col_idxs, slice):
            K_n_cols = len(range(*col_idxs.indices(K_n_rows)))
            is_M_subset = (
                col_idxs.start is None
                and (col_idxs.stop is None or col_idxs.stop % dim_i == 0)
                and col_idxs.step is None
            )
            if is_M_subset:
                if col_idxs.stop is not None:
                    points = (0, int(col_idxs.stop // dim_i))
            else:
                idx = np.arange(K_n_rows)[col_idxs]
        else:
            idx = np.asarray(col_idxs)
            assert len(idx) == len(set(idx.tolist()))  # train.py:1341
            assert np.array_equal(idx, np.sort(idx))  # train.py:1345
            K_n_cols = len(idx)
        if K_n_cols > K_n_rows:
            raise ValueError('Columns indexed beyond range.')

        if callback is not None:
            callback(0, 100)
        start = timeit.default_timer()
        ctx = self._context()
        ctx.train_upload(R_desc, R_d_desc, _lib.tril_perms_from_lin(tril_perms_lin, dim_d))
        K = ctx.assemble_K(sig, use_E_cstr, points=points, idx=idx, alloc_extra_rows=alloc_extra_rows,
                           to_host=True)
        if callback is not None:
            dur_s = timeit.default_timer() - start
            callback(DONE, sec_disp_str='took {:.1f} s'.format(dur_s) if dur_s >= 0.1 else '')
        return K

    # ------------------------------------------------------------------ sampling

    def draw_strat_sample(self, T, n, excl_idxs=None):
        """Stratified sample of n indices preserving the distribution of T (train.py:1537-1646):
        Freedman-Diaconis histogram, proportional per-bin counts, uniform draws inside bins."""
        if excl_idxs is None or len(excl_idxs) == 0:
            excl_idxs = None
        if n == 0:
            return np.array([], dtype=np.uint)
        if T.size == n:
            assert excl_idxs is None
            return np.arange(n)
        if n == 1:
            cands = np.setdiff1d(np.arange(T.size), excl_idxs, assume_unique=True)
            return np.array([np.random.choice(cands)])

        q75, q25 = np.percentile(T, [75, 25])
        h = 2 * (q75 - q25) / np.cbrt(n)
        n_bins = int(np.ceil((np.max(T) - np.min(T)) / h)) if h > 0 else 1
        n_bins = min(n_bins, int(n / 2))
        bins = np.linspace(np.min(T), np.max(T), n_bins, endpoint=False)
        bin_of = np.digitize(T, bins)
        if excl_idxs is not None:
            bin_of[excl_idxs] = n_bins + 1  # parked in an impossible bin

        uniq_all, cnts_all = np.unique(bin_of, return_counts=True)
        if excl_idxs is not None:
            keep = uniq_all != n_bins + 1
            uniq_all, cnts_all = uniq_all[keep], cnts_all[keep]

        reduced = np.ceil(cnts_all / np.sum(cnts_all, dtype=float) * n).astype(int)
        reduced = np.minimum(reduced, cnts_all)
        delta = n - np.sum(reduced)
        while np.abs(delta) > 0:
            max_red = np.min(reduced[np.where(reduced > 1)]) - 1
            picked = np.random.choice(
                uniq_all,
                min(max_red, np.abs(delta)),
                p=(reduced - 1) / np.sum(reduced - 1, dtype=float),
                replace=True,
            )
            u, cnt = np.unique(picked, return_counts=True)
            where = np.where(np.isin(uniq_all, u, assume_unique=True))[0]
            reduced[where] += np.sign(delta) * cnt
            delta = n - np.sum(reduced)

        out = np.empty((0,), dtype=int)
        for b, cnt in zip(uniq_all, reduced):
            members = np.where(bin_of.ravel() == b)[0]
            out = np.append(out, np.random.choice(members, cnt, replace=False))
        return out
