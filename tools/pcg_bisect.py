"""Bisect of the PCG mid-trajectory spread on the reference fixture pcg_n12_p6_m200 (round-4 review, weak #1a: the
reference first passed 3e-3 ||y|| at step 120, the oracle at 128, the GPU loop at 272 -- all converging after 523 +- 10 %).

One host CG loop (oracle.pcg: the recurrence of scipy's cg) is run with every combination of
    mat-vec:          dense K @ v on the CPU  |  gdml_kernel_matvec (GPU)
    preconditioner:   oracle factor on the CPU  |  gdml_precon_apply (GPU, stored factor)  |  GPU, matrix-free form  |  GPU, fp32 form
and the device loop gdml_pcg at pcg.depth 0 and 2 in every preconditioner form (maxiter 1500 = "does not converge").  For each run: iterations to rtol 1e-4 and the
first iteration at which ||r|| passes 0.3 ... 1e-3 of ||y||.  Plus the operator differences themselves.  CPU-only companions
(tools/pcg_cpu_spread.py, run in the build container) show what two CPU runs of the SAME algorithm do to each other.

    python tools/pcg_bisect.py > gpurun_out/pcg_bisect.txt
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from _pcg_compare import crossings  # noqa: E402
from oracle import gdml_oracle as orc  # noqa: E402
from sgdml_amd import _lib  # noqa: E402

LEVELS = (0.3, 0.1, 0.03, 0.01, 3e-3, 1e-3)


def main():
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'pcg_n12_p6_m200.npz'), allow_pickle=True))
    M, N = g['R_train'].shape[:2]
    sig, lam, y, idx = float(g['sig']), float(g['lam']), g['y'], g['inducing_pts_idxs']
    ny = np.linalg.norm(y)
    tp = orc.tril_perms_from_atom_perms(g['perms'])
    c = _lib.Context()
    xd, gd = c.desc_from_R(g['R_train'].reshape(M, -1), N)
    c.train_upload(xd, gd, tp)
    K = c.assemble_K(sig, False, to_host=True)  # the reference's matrix to 1e-12 (test_iterative_solver_with_permutation_group...)
    K_nm = np.ascontiguousarray(K[:, idx])
    # oracle factor (iterative.py:208-351) from that matrix
    K_mm = -K_nm[idx, :].copy()
    # how many jitter escalations does LAPACK need on K_mm (iterative.py:442-463)?  (the GPU reports its own count in info >> 8)
    eps = np.finfo(float).eps
    trial = K_mm.copy()
    trial[np.diag_indices_from(trial)] += eps
    n_jit_cpu = 0
    for reg in 10.0 ** np.arange(-15, 2):
        try:
            sla.cho_factor(trial, overwrite_a=False, check_finite=False)
            break
        except np.linalg.LinAlgError:
            trial[np.diag_indices_from(trial)] += reg
            n_jit_cpu += 1
    ev = np.linalg.eigvalsh(K_mm)
    print('K_mm: eigenvalues %.3e ... %.3e, %d below 1e-14 |max|; CPU Cholesky needs %d jitter escalations' %
          (ev[0], ev[-1], int((ev < 1e-14 * ev[-1]).sum()), n_jit_cpu))
    L_mm, lower = orc.cho_factor_stable(K_mm, pre_reg=True)
    B = sla.solve_triangular(L_mm, K_nm.T, lower=lower, trans='T', check_finite=False).T
    inner = B.T @ B
    inner[np.diag_indices_from(inner)] += lam
    L, lower = orc.cho_factor_stable(inner, eps_mag_max=-14)
    fac = np.ascontiguousarray(sla.solve_triangular(L, B.T, lower=lower, trans='T', check_finite=False))

    def gpu_setup(form):
        c.set_option('pcg.precon_form', form)
        c.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
        _, _, info = c.nystroem_factor(lam, idx, want_lev=False)
        c.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
        print('   [GPU factor: form %d, info = %d -> jitter escalations %d, QR branch %d]' % (form, info, info >> 8, info & 1))
        return info

    mv_cpu = lambda v: -(K @ v - lam * v)
    mv_gpu = lambda v: -c.kernel_matvec(lam, False, v)
    P_cpu = lambda r: orc.precon_apply(fac, lam, r)
    P_gpu = lambda r: c.precon_apply(lam, r)

    print('fixture pcg_n12_p6_m200: n = %d, m = %d, ||y|| = %.4f' % (len(y), len(idx), ny))
    print('reference (scipy cg inside Iterative.solve)          : iters %4d  crossings %s' %
          (int(g['n_iters']), crossings(g['resid_hist'], ny, LEVELS).tolist()))

    # ---- operator differences
    rng = np.random.default_rng(3)
    v = rng.standard_normal(len(y))
    gpu_setup(0)
    a, b = mv_cpu(v), mv_gpu(v)
    print('mat-vec     |GPU - CPU| / |CPU|                       : %.2e' % (np.abs(a - b).max() / np.abs(a).max()))
    pc, pg = P_cpu(v), P_gpu(v)
    print('precon      |GPU stored - CPU| / |CPU|                : %.2e' % (np.abs(pc - pg).max() / np.abs(pc).max()))
    gpu_setup(1)
    pm = P_gpu(v)
    print('precon      |GPU matrix-free - CPU| / |CPU|           : %.2e' % (np.abs(pc - pm).max() / np.abs(pc).max()))
    print('precon      |GPU matrix-free - GPU stored| / |stored| : %.2e' % (np.abs(pg - pm).max() / np.abs(pg).max()))

    def host_loop(A, P, name):
        h = []
        x, info, it, res = orc.pcg(A, y, M_mv=lambda r: (h.append(np.linalg.norm(r)), P(r))[1], rtol=1e-4, maxiter=1500)
        hist = np.array(h[1:] + [res])
        print('%-52s : iters %4d  crossings %s' % (name, it, crossings(hist, ny, LEVELS).tolist()), flush=True)

    def dev_loop(depth, name):
        c.set_option('pcg.depth', depth)
        h = []
        x, info, it, res = c.pcg(lam, False, y, rtol=1e-4, maxiter=1500, callback=lambda i, r, f: h.append(r) or False)
        print('%-52s : iters %4d  crossings %s' % (name, it, crossings(np.array(h), ny, LEVELS).tolist()), flush=True)

    print('levels (fraction of ||y||): %s' % (LEVELS,))
    gpu_setup(0)
    host_loop(mv_cpu, P_cpu, 'host loop: CPU mat-vec, CPU precon')
    host_loop(mv_cpu, P_gpu, 'host loop: CPU mat-vec, GPU precon (stored)')
    host_loop(mv_gpu, P_cpu, 'host loop: GPU mat-vec, CPU precon')
    host_loop(mv_gpu, P_gpu, 'host loop: GPU mat-vec, GPU precon (stored)')
    dev_loop(0, 'device loop gdml_pcg, depth 0, stored')
    dev_loop(2, 'device loop gdml_pcg, depth 2, stored')
    gpu_setup(1)
    host_loop(mv_cpu, P_gpu, 'host loop: CPU mat-vec, GPU precon (matrix-free)')
    host_loop(mv_gpu, P_gpu, 'host loop: GPU mat-vec, GPU precon (matrix-free)')
    dev_loop(0, 'device loop gdml_pcg, depth 0, matrix-free')
    dev_loop(2, 'device loop gdml_pcg, depth 2, matrix-free')
    gpu_setup(3)
    pf = P_gpu(v)
    print('precon      |GPU fp32 form - CPU| / |CPU|              : %.2e' % (np.abs(pc - pf).max() / np.abs(pc).max()))
    host_loop(mv_cpu, P_gpu, 'host loop: CPU mat-vec, GPU precon (fp32 + Gram corr.)')
    host_loop(mv_gpu, P_gpu, 'host loop: GPU mat-vec, GPU precon (fp32 + Gram corr.)')
    dev_loop(0, 'device loop gdml_pcg, depth 0, fp32 + Gram corr.')
    dev_loop(2, 'device loop gdml_pcg, depth 2, fp32 + Gram corr.')
    c.close()


if __name__ == '__main__':
    main()
