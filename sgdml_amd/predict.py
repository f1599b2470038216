"""
``GDMLPredict`` with the reference's public interface (sgdml/predict.py:248-1294), evaluated by
the HIP prediction kernels (csrc/predict.hip).  Outputs, scaling (std, c) and the training-set
mode (``predict()`` with cached descriptors, predict.py:1221-1233) follow the reference.
"""
import logging
import sys

import numpy as np

from . import _lib
from .utils.desc import Desc


class GDMLPredict(object):
    def __init__(
        self,
        model,
        batch_size=None,
        num_workers=None,
        max_memory=None,
        max_processes=None,
        use_torch=False,
        log_level=None,
        _borrow_ctx=None,
        devices=None,
    ):
        """devices (no counterpart in the reference's signature; its torch path wraps the model in nn.DataParallel over every
        visible GPU, predict.py:375-378): list of GPU indices to replicate the model on -- query batches are then split over
        them (replicas + query sharding).  None: every visible GPU when `use_torch` is set (what the reference
        does), else GPU 0 only."""
        self.log = logging.getLogger(__name__)
        if log_level is not None:
            self.log.setLevel(log_level)

        if 'type' not in model or not (model['type'] == 'm' or model['type'] == b'm'):
            self.log.critical('The provided data structure is not a valid model.')
            sys.exit()

        self.n_atoms = model['z'].shape[0]
        self.desc = Desc(self.n_atoms, max_processes=max_processes)

        self.R_desc = None
        self.R_d_desc = None

        self.lat_and_inv = (
            (np.asarray(model['lattice'], dtype=np.float64), np.linalg.inv(model['lattice']))
            if 'lattice' in model
            else None
        )

        R_desc_train = np.ascontiguousarray(np.asarray(model['R_desc'], dtype=np.float64).T)  # stored D x M
        self.n_train = R_desc_train.shape[0]
        self.sig = float(model['sig'])
        self.std = float(model['std']) if 'std' in model else 1.0
        self.c = float(model['c'])
        self.n_perms = np.asarray(model['perms']).shape[0]
        self.tril_perms_lin = np.asarray(model['tril_perms_lin'])
        self._tril_perms = _lib.tril_perms_from_lin(self.tril_perms_lin, self.desc.dim)

        # parameters of the reference's CPU pool; accepted and ignored (no pools on the GPU path)
        self.max_memory, self.max_processes = max_memory, max_processes
        self.use_torch = use_torch
        self.bulk_mp, self.num_workers, self.chunk_size = False, 0, self.n_train
        self.pool = None

        # _borrow_ctx (internal): evaluate on a context somebody else owns -- the trainer's, for the short-lived predictors of
        # GDMLTrain._recov_int_const and sweep.sigma_sweep.  The model tables of a context belong to whoever uploaded last.
        self._owns_ctx = _borrow_ctx is None
        if devices is None:
            devices = list(range(_lib.device_count())) if (use_torch and _borrow_ctx is None) else [0]
        devices = [int(d) for d in devices] or [0]
        self._ctx = _lib.Context(devices[0]) if _borrow_ctx is None else _borrow_ctx
        # replicas on the other devices (one context per entry: the same index twice gives two contexts on one GPU, which is
        # how the sharded path is tested on a one-GPU box); the training-set mode and set_alphas stay on the first one
        self._replicas = [] if _borrow_ctx is not None else [_lib.Context(d) for d in devices[1:]]
        for ctx in [self._ctx] + self._replicas:
            ctx.predict_upload_model(
                R_desc_train,
                model['R_d_desc_alpha'],
                self._tril_perms,
                self.sig,
                model['alphas_E'] if 'alphas_E' in model else None,
            )
        self._replicas_stale = False
        self._min_shard = 64  # geometries per device below which a batch is not worth splitting
        self._train_resident = False

    def __del__(self):
        for ctx in getattr(self, '_replicas', []):
            ctx.close()
        ctx = getattr(self, '_ctx', None)
        if ctx is not None and getattr(self, '_owns_ctx', True):
            ctx.close()

    def _sharded(self, n_items, fn):
        """fn(ctx, lo, hi) for contiguous query shards, one per device, concurrently (ctypes releases the GIL inside the
        library calls); results in shard order.  One shard when the batch is small or the replicas do not hold the current
        coefficients (set_alphas re-targets only the first context)."""
        ctxs = [self._ctx] + ([] if self._replicas_stale else self._replicas)
        n_sh = min(len(ctxs), max(1, n_items // self._min_shard))
        if n_sh <= 1:
            return [fn(self._ctx, 0, n_items)]
        import threading

        cuts = [n_items * k // n_sh for k in range(n_sh + 1)]
        out, err = [None] * n_sh, []

        def run(k):
            try:
                out[k] = fn(ctxs[k], cuts[k], cuts[k + 1])
            except BaseException as e:  # re-raised in the caller's thread
                err.append(e)

        ths = [threading.Thread(target=run, args=(k,)) for k in range(1, n_sh)]
        for t in ths:
            t.start()
        run(0)
        for t in ths:
            t.join()
        if err:
            raise err[0]
        return out

    # ---- training-mode hooks (predict.py:510-601)

    def set_R_desc(self, R_desc):
        self.R_desc = R_desc
        self._train_resident = False

    def set_R_d_desc(self, R_d_desc):
        self.R_d_desc = R_d_desc
        self._train_resident = False

    def _ensure_train_resident(self):
        if self._train_resident:
            return
        if self.R_d_desc is None:
            raise AssertionError('set_R_d_desc() must be called first')
        R_desc = self.R_desc
        if R_desc is None:
            raise AssertionError('set_R_desc() must be called first')
        self._ctx.train_upload(R_desc, self.R_d_desc, self._tril_perms)
        self._train_resident = True

    def set_alphas(self, alphas_F, alphas_E=None):
        """Re-target the model with new coefficients (predict.py:551-601); J alpha runs on the GPU."""
        self._ensure_train_resident()
        self._ctx.set_alphas(alphas_F, alphas_E)
        self._replicas_stale = True  # the replicas still hold the constructor's coefficients

    # ---- CPU-tuning API of the reference: no-ops here, like its torch path (predict.py:826-830)

    def prepare_parallel(self, n_bulk=1, n_reps=1, return_is_from_cache=False):
        return None

    def set_opt_num_workers_and_batch_size_fast(self, n_bulk=1, n_reps=1):
        return None

    def _set_num_workers(self, num_workers=None, force_reset=False):
        self.num_workers = 0

    def _set_chunk_size(self, chunk_size=None):
        self.chunk_size = self.n_train

    def _set_batch_size(self, batch_size=None):
        self._set_chunk_size(batch_size)

    def _set_bulk_mp(self, bulk_mp=False):
        self.bulk_mp = False

    def get_GPU_batch(self):
        return self.n_train

    # ---- test / validation errors on the device (the loop body of sgdml/cli.py:1564-1605)

    def test_errors(self, R, F, E=None):
        """MAE / RMSE of energies, force components, force magnitudes and normalised force angles
        for labelled geometries, with the reference's definitions (cli.py:_online_err :1170).
        Predictions never leave the GPU.  Returns a dict of (mae, rmse) pairs."""
        R = np.asarray(R, dtype=np.float64)
        if R.ndim == 1:
            R = R[None, :]
        B = R.shape[0]
        R = R.reshape(B, -1)
        F = np.asarray(F, dtype=np.float64).reshape(B, -1)
        E = None if E is None else np.asarray(E, dtype=np.float64).reshape(B)
        parts = self._sharded(B, lambda ctx, lo, hi: ctx.predict_errors(
            R[lo:hi], F[lo:hi], None if E is None else E[lo:hi], std=self.std, c=self.c, lat_and_inv=self.lat_and_inv))
        s = np.sum(np.asarray(parts, dtype=np.float64), axis=0)  # the eight sums are additive over query shards
        n_f, n_a = B * 3 * self.n_atoms, B * self.n_atoms
        out = {
            'force': (s[2] / n_f, np.sqrt(s[3] / n_f)),
            'magnitude': (s[4] / n_a, np.sqrt(s[5] / n_a)),
            'angle': (s[6] / n_a, np.sqrt(s[7] / n_a)),
        }
        if E is not None:
            out['energy'] = (s[0] / B, np.sqrt(s[1] / B))
        return out

    # ---- prediction

    def predict(self, R=None, return_E=True):
        """Energies (B,) and forces (B,3N) for geometries R (B,3N); R=None -> training-set mode."""
        if R is not None:
            R = np.asarray(R, dtype=np.float64)
            if R.ndim == 1:
                R = R[None, :]
            R = R.reshape(R.shape[0], -1)
            parts = self._sharded(R.shape[0], lambda ctx, lo, hi: ctx.predict(R[lo:hi], self.lat_and_inv, return_E=return_E))
            F = parts[0][1] if len(parts) == 1 else np.concatenate([p_[1] for p_ in parts])
            E = None if not return_E else (parts[0][0] if len(parts) == 1 else np.concatenate([p_[0] for p_ in parts]))
        else:
            if self.R_desc is None or self.R_d_desc is None:
                self.log.critical(
                    'A reference to the training geometry descriptors and Jacobians needs to be set '
                    'for this function to work without arguments.'
                )
                raise AssertionError('training descriptors not set')
            self._ensure_train_resident()
            E, F = self._ctx.predict(None, None, return_E=return_E)

        F *= self.std  # predict.py:1286-1288
        ret = (F,)
        if return_E:
            E *= self.std
            E += self.c
            ret = (E,) + ret
        return ret
