"""
Closed-form solver with the reference's ``Analytic`` interface (sgdml/solvers/analytic.py:37-159).

The kernel matrix is assembled on the GPU, stays in HBM, is factored in place by the blocked fp64
MFMA Cholesky (csrc/chol.hip) and never crosses PCIe; only y goes in and alphas come out.
"""
import logging
import timeit
from functools import partial

import numpy as np

from .. import DONE, NOT_DONE
from .. import _lib


class Analytic(object):
    def __init__(self, gdml_train, desc, callback=None):
        self.log = logging.getLogger(__name__)
        self.gdml_train = gdml_train
        self.desc = desc
        self.callback = callback
        self.n_refine = 0  # iterative-refinement steps (0 = LAPACK cho_solve semantics)
        self.used_lu = False  # set when the last solve went through the LU branch (analytic.py:101-114)

    def solve(self, task, R_desc, R_d_desc, tril_perms_lin, y):
        sig, lam, use_E_cstr = task['sig'], task['lam'], task['use_E_cstr']
        n_train, dim_d = R_d_desc.shape[:2]

        ctx = self.gdml_train._context()
        ctx.train_upload(R_desc, R_d_desc, _lib.tril_perms_from_lin(tril_perms_lin, dim_d))

        if ctx.comm_info()[1] > 1:
            # GDMLTrain.init_distributed: the system matrix is partitioned block-row-cyclic over the ranks and
            # factored by the distributed Cholesky (csrc/dist_chol.hip); every rank gets the coefficients.  With energy
            # constraints y carries the M energy labels and the library appends the M energy rows (train.py:235-300)
            if self.callback is not None:
                cb = partial(self.callback, disp_str='Solving linear system (distributed Cholesky factorization)')
                cb(NOT_DONE)
            start = timeit.default_timer()
            try:
                alphas = ctx.dist_chol_solve(sig, lam, y)
            except np.linalg.LinAlgError:
                # analytic.py:101-114: "try a solver that makes less assumptions".  There is no distributed LU: when the
                # whole matrix fits one GPU every rank runs the single-GPU LU branch on its own device, redundantly (the
                # failing pivot is a property of the replicated inputs, so every rank arrives here together).
                n = len(y)
                need = Analytic.est_device_memory(n_train, (1 + int(np.sqrt(8 * dim_d + 1))) // 2, use_E_cstr)
                _, free_b, _ = ctx.mem_info()
                fits = need <= free_b + ctx.resident_K_bytes()
                # free HBM differs between ranks: the decision is collective (minimum over the ranks), or one rank re-raises
                # while the others finish the LU and the next collective hangs
                all_min = getattr(ctx, '_all_min', None)
                if all_min is not None:
                    fits = bool(all_min(1.0 if fits else 0.0))
                if not fits:
                    raise
                if self.callback is not None:
                    cb = partial(self.callback, disp_str='Solving linear system (LU factorization)      ')
                    cb(NOT_DONE)
                self.used_lu = True
                with ctx.comm_suspended():
                    ctx.assemble_K(sig, use_E_cstr)
                    alphas = ctx.lu_solve(lam, y)
            if self.callback is not None:
                dur_s = timeit.default_timer() - start
                self.callback(DONE, disp_str='Training on {:,} points'.format(n_train),
                              sec_disp_str='took {:.1f} s'.format(dur_s) if dur_s >= 0.1 else '')
            return alphas

        cb = self.callback
        if cb is not None:
            cb = partial(cb, disp_str='Assembling kernel matrix')
            cb(0, 100)
        start = timeit.default_timer()
        # A = -K + lam I, device resident (analytic.py:65,82 fused into the assembly where the kernel allows it);
        # one spare row: the right-hand side rides along through the factorisation, whose panel solves and
        # trailing updates then perform the forward substitution of cho_solve (analytic.py:97)
        ctx.assemble_K(sig, use_E_cstr, alloc_extra_rows=1, for_cholesky=lam)
        ctx.chol_set_rhs(y)
        if cb is not None:
            dur_s = timeit.default_timer() - start
            cb(DONE, sec_disp_str='took {:.1f} s'.format(dur_s) if dur_s >= 0.1 else '')
            cb = partial(self.callback, disp_str='Solving linear system (Cholesky factorization)')
            cb(NOT_DONE)

        start = timeit.default_timer()
        try:
            # A = -K + lam I, A = L L^T (analytic.py:65,82,94)
            ctx.chol_factor(lam)
            alphas = ctx.chol_solve(None, n_refine=self.n_refine)  # backward substitution; = -A^-1 y (analytic.py:97-99)
        except np.linalg.LinAlgError:  # "Try a solver that makes less assumptions" (analytic.py:101-114)
            if self.callback is not None:
                cb = partial(self.callback, disp_str='Solving linear system (LU factorization)      ')  # Keep whitespaces!
                cb(NOT_DONE)
            self.used_lu = True
            # the failed Cholesky consumed the matrix (the reference keeps a second copy, overwrite_a=False): the
            # LU needs both triangles, so the full K is assembled again (milliseconds) and factored with partial
            # pivoting on the device
            ctx.assemble_K(sig, use_E_cstr)
            alphas = ctx.lu_solve(lam, y)

        if self.callback is not None:
            dur_s = timeit.default_timer() - start
            self.callback(
                DONE,
                disp_str='Training on {:,} points'.format(n_train),
                sec_disp_str='took {:.1f} s'.format(dur_s) if dur_s >= 0.1 else '',
            )
        return alphas

    @staticmethod
    def est_memory_requirement(n_train, n_atoms):
        """Host-RAM estimate of the reference (analytic.py:153-159), kept for API parity."""
        est_bytes = 3 * (n_train * 3 * n_atoms) ** 2 * 8
        est_bytes += (n_train * 3 * n_atoms) * 8
        return est_bytes

    @staticmethod
    def est_device_memory(n_train, n_atoms, use_E_cstr=False):
        """HBM needed by this backend: K is factored in place -> one n x n fp64 matrix."""
        n = n_train * 3 * n_atoms + (n_train if use_E_cstr else 0)
        ld = (n + 15) // 16 * 16  # rows start on 128-byte boundaries (gdml_assemble_K)
        return (n + 1) * ld * 8 + 8 * n * 8  # + the right-hand-side row, + solve vectors
