#!/usr/bin/env python
"""The REFERENCE's CPU path timed beside the oracle (the NumPy port bench.py's cpu_baseline leg times on the GPU box) on
identical inputs, in the two thread layouts of BASELINE.md section 3.2.  Build container only (needs /root/reference):

    python tools/cpu_ref_vs_port.py [M]          -> profiles/r06_cpu_reference_vs_port_m<M>.json

  layout "blas":  one Python process, BLAS/LAPACK threads = all cores   (best for the Cholesky)
  layout "procs": reference worker pool = all cores, BLAS threads = 1    (best for the reference's assembly: it forks
                  one worker per core over block columns, train.py:1491-1523; the port has no pool -- it vectorises over
                  row points instead -- so its number under this layout is its single-thread number)
Phases are timed where the reference times itself: _assemble_kernel_mat (train.py:1260-1535) and Analytic.solve minus the
assembly (analytic.py:75-140).  The ratio reference / port per phase is what turns the port's full-size measurement on the
GPU box (profiles/r03_cpu_baseline_full.json) into a statement about the reference."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(layout, M):
    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import bench
    import make_golden_r2 as g2
    from oracle import gdml_oracle as orc

    N, sig, lam = 21, 20, 1e-10
    cores = os.cpu_count() or 1
    R, E, F = bench.synth_geometries(N, M, seed=0)
    r = g2.ref()
    Desc = r['Desc']
    procs = cores if layout == 'procs' else 1
    import sgdml.train as ref_train_mod

    gt = r['train']
    gt._max_processes = procs
    desc = Desc(N, max_processes=procs)
    perms = np.arange(N)[None, :]
    tril_perms = np.array([Desc.perm(p) for p in perms])
    lin = (tril_perms + np.arange(1)[:, None] * desc.dim).flatten('F')
    R_desc, R_d_desc = desc.from_R(R.reshape(M, -1))
    y = F.ravel() / np.std(F)
    ds = {'R': R, 'E': E, 'F': F, 'z': np.full(N, 6)}
    task = g2.make_task(ds, M, perms, sig, lam)
    # ---- reference
    t0 = time.perf_counter()
    K = gt._assemble_kernel_mat(R_desc, R_d_desc, lin, sig, desc)
    t_ref_asm = time.perf_counter() - t0
    del K
    from sgdml.solvers.analytic import Analytic

    an = Analytic(gt, desc)
    t0 = time.perf_counter()
    # a COPY of y: the reference's LU branch calls scipy.linalg.solve(..., overwrite_b=True) (analytic.py:112-114), which
    # leaves the solution in the caller's label vector -- the round-5 record computed the reference's residual against that
    # overwritten vector and reported 0.99999 for a solve that was in fact fine
    y_ref = y.copy()
    alphas_ref = an.solve(task, R_desc, R_d_desc, lin, y_ref)
    t_ref_total = time.perf_counter() - t0  # assembles again inside (analytic.py:57-63)
    ref_took_lu = not np.array_equal(y_ref, y)
    # ---- port
    xo, go = orc.desc_from_R(R.reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(perms)
    lin_o = orc.tril_perms_lin_from_tril_perms(tp)
    t0 = time.perf_counter()
    Ko = orc.assemble_K(xo, go, lin_o, sig)
    t_port_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    alphas_port, used_lu = orc.analytic_solve(Ko, y, lam)
    t_port_solve = time.perf_counter() - t0
    n = 3 * N * M
    A = -Ko + lam * np.eye(n)
    res = {'layout': layout, 'M': M, 'n': n, 'cores': cores, 'reference_processes': procs,
           'blas_threads': os.environ.get('OPENBLAS_NUM_THREADS', 'all'),
           'reference': {'assemble_s': t_ref_asm, 'solve_s': max(0.0, t_ref_total - t_ref_asm), 'analytic_solve_total_s': t_ref_total},
           'port': {'assemble_s': t_port_asm, 'solve_s': t_port_solve, 'lu_fallback': bool(used_lu)},
           'reference_lu_fallback': bool(ref_took_lu),
           'resid_reference': float(np.linalg.norm(A @ (-alphas_ref) - y) / np.linalg.norm(y)),
           'resid_port': float(np.linalg.norm(A @ (-alphas_port) - y) / np.linalg.norm(y))}
    print('RESULT ' + json.dumps(res), flush=True)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    out = {}
    for layout in ('blas', 'procs'):
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        if layout == 'procs':
            env.update(OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', layout, str(M)], env=env,
                           capture_output=True, text=True)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('RESULT ')]
        if not lines:
            raise SystemExit('child %s failed: %s' % (layout, (p.stdout + p.stderr)[-2000:]))
        out[layout] = json.loads(lines[-1][7:])
    # per phase: the reference's best layout over the port's best layout
    best = {}
    for ph in ('assemble_s', 'solve_s'):
        ref_best = min(out[k]['reference'][ph] for k in out)
        port_best = min(out[k]['port'][ph] for k in out)
        best[ph] = {'reference_best_s': ref_best, 'port_best_s': port_best, 'reference_over_port': ref_best / port_best}
    import numpy
    import scipy

    rec = {'what': 'reference (sgdml, /root/reference) vs oracle/gdml_oracle.py on identical inputs, build container',
           'n_atoms': 21, 'M': M, 'layouts': out, 'best_of_layouts': best,
           'host': {'nproc': os.cpu_count(), 'numpy': numpy.__version__, 'scipy': scipy.__version__}}
    path = os.path.join(ROOT, "profiles", "r06_cpu_reference_vs_port_m%d.json" % M)
    with open(path, 'w') as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec['best_of_layouts']))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]))
    else:
        main()
