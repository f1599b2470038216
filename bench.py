#!/usr/bin/env python
"""
Benchmark of the sGDML hot path on MI355X (BASELINE.json metric: kernel-matrix build+solve
wall-clock [s] and predict forces/s at N_train=1000, aspirin-sized = 21 atoms).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N=1: plain python)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`value` measures the SAME workload for every --gpus N (--workload, default analytic), so that a driver's N = 1, 2, 4, 8 runs
form a curve; `metric` / `config.workload` come from line_skeleton() and are identical across N:

analytic (default; BASELINE.json configs[1], N = 21, N_train = 1000, n = 63 000): one "step" = one pass of the hot path over one
synthetic batch, all inputs resident in HBM before the timed region starts:
    assemble -K + lam I (fp64, lower blocks, stays in HBM)  ->  in-place fp64 MFMA Cholesky  ->  triangular solves (alphas)
    [N = 1 only: -> batched force/energy prediction of B query geometries]
  `value` = build+solve seconds.  N = 1: single-GPU factorisation; the line also carries `configs` (the configs[0]-shaped
  sigma sweep, configs[2] / [3] / [4] to solver_tol on this one GPU), `predict`, `first_call_in_process`, `symmetry_search`, `cpu_baseline`.
  N > 1: the same system block-row-cyclic over the ranks through gdml_dist_chol_solve (RCCL broadcasts + all-gathers),
  "scaling": "strong"; the configs[2] step and its run to solver_tol ride along (`cg`, `time_to_tol`).
cg (BASELINE.json configs[2]: aspirin N_train=5000, iterative solver sharded over the GPUs with RCCL): one "step" = this
  rank's rows of K_nm (n x m, m = 3N k inducing columns), Nystroem factor (RCCL all-reduce of the m x m blocks), then a
  FIXED number of PCG iterations (query-sharded mat-vec + all-gather, row-sharded preconditioner: all-reduce of an m-vector
  + all-gather).  `value` = seconds per step (max over ranks), any N including 1.
Both lines carry `scale_point` (analytic seconds) and / or `scale_point_cg` under the same keys.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TF = 78.6  # MI355X fp64 matrix peak (256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk)


def line_skeleton(workload, n_atoms, n_train, world, cg_iters=None, cg_inducing=None):
    """`metric` and `config.workload` of a bench line: ONE string per workload, the same for every number of GPUs, so that the
    values of `--gpus 1, 2, 4, 8` form a curve (tests/test_host_policy_cpu.py).  What differs between the lines lives in
    `n_gpus` and `config.parallelism`."""
    if workload == 'analytic':
        return {'metric': 'kernel-matrix build+solve wall-clock, N_train={} {}-atom (aspirin-sized)'.format(n_train, n_atoms),
                'workload': 'aspirin-sized N={} N_train={} analytic Cholesky (BASELINE.json configs[1])'.format(n_atoms, n_train),
                'parallelism': 'single GPU' if world == 1 else
                               'kernel matrix block-row-cyclic over {} GPUs, distributed Cholesky (gdml_dist_chol_solve)'.format(world)}
    if workload == 'cg':
        return {'metric': 'sharded Nystroem-PCG solve wall-clock per step (K_nm rows + preconditioner + {} PCG iterations), '
                          'N_train={} {}-atom (aspirin-sized)'.format(cg_iters, n_train, n_atoms),
                'workload': 'aspirin-sized N={} N_train={} iterative solver (Nystroem-preconditioned CG, k={} inducing points, {} '
                            'iterations per step) (BASELINE.json configs[2])'.format(n_atoms, n_train, cg_inducing, cg_iters),
                'parallelism': 'single GPU' if world == 1 else
                               'row-sharded Nystroem factor + query-sharded mat-vec over {} ranks'.format(world)}
    raise ValueError(workload)


def synth_geometries(n_atoms, n_frames, seed=0, jitter=0.3, n_conformers=4, spacing=1.4):
    """Seeded synthetic frames: conformers of a jittered cubic-grid molecule,
    labels from the analytic pair potential E = sum 1/d, F = -grad E."""
    rng = np.random.RandomState(seed)
    g = int(np.ceil(n_atoms ** (1.0 / 3.0))) + 1
    grid = np.array([[a, b, c] for a in range(g) for b in range(g) for c in range(g)], float) * spacing
    base0 = grid[rng.choice(len(grid), n_atoms, replace=False)]
    bases = np.array([base0] + [base0 + rng.normal(0, 0.25, base0.shape) for _ in range(n_conformers - 1)])
    R = bases[rng.randint(0, len(bases), n_frames)] + rng.normal(0, jitter, (n_frames, n_atoms, 3))
    return (R,) + pair_potential_labels(R)


def synth_trajectory(n_atoms, n_frames, seed=0, n_modes=8, amp=0.15, noise=0.01, spacing=1.4):
    """Seeded synthetic "MD trajectory" for the iterative-solver workload (configs[2]): the molecule of synth_geometries
    moving along n_modes collective displacement modes (random amplitudes) plus a small thermal jitter -- like a real
    trajectory the frames lie near a low-dimensional manifold, which is what makes the Nystroem preconditioner of the
    reference (iterative.py:208-351) effective; labels from the same pair potential."""
    rng = np.random.RandomState(seed)
    g = int(np.ceil(n_atoms ** (1.0 / 3.0))) + 1
    grid = np.array([[a, b, c] for a in range(g) for b in range(g) for c in range(g)], float) * spacing
    base = grid[rng.choice(len(grid), n_atoms, replace=False)]
    modes = np.linalg.qr(rng.normal(size=(3 * n_atoms, n_modes)))[0].T.reshape(n_modes, n_atoms, 3)
    coef = rng.normal(0, amp, (n_frames, n_modes))
    R = base[None] + np.einsum('fk,kad->fad', coef, modes) * np.sqrt(n_atoms) + rng.normal(0, noise, (n_frames, n_atoms, 3))
    return (R,) + pair_potential_labels(R)


def pair_potential_labels(R):
    n_frames, n_atoms = R.shape[:2]
    i, j = np.tril_indices(n_atoms, -1)
    diff = R[:, i, :] - R[:, j, :]
    dist = np.sqrt((diff**2).sum(-1))
    E = (1.0 / dist).sum(-1)
    gp = diff / (dist**3)[..., None]
    F = np.zeros_like(R)
    for m in range(n_frames):
        np.add.at(F[m], i, gp[m])
        np.subtract.at(F[m], j, gp[m])
    return E, F


def start_cpu_sample(M):
    """The CPU leg's largest sample runs WHILE the GPU works on the analytic entries of the configs[] section (long kernels,
    the host cores are idle then):
    tools/cpu_baseline_full.py M in a child process -- the oracle at M training points with the BLAS / LAPACK pool on all
    host cores.  Returns (process, output path) or None."""
    import subprocess
    import tempfile

    try:
        fd, path = tempfile.mkstemp(prefix='gdml_cpu_sample_', suffix='.json')
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'cpu_baseline_full.py'), str(int(M))],
                                stdout=fd, stderr=subprocess.DEVNULL, cwd=ROOT)
        os.close(fd)
        return proc, path
    except Exception:
        return None


def finish_cpu_sample(handle, timeout):
    if handle is None:
        return None
    proc, path = handle
    try:
        proc.wait(timeout=timeout)
        with open(path) as f:
            rec = json.loads(f.read().strip().splitlines()[-1])
        return rec
    except Exception:
        try:
            proc.kill()
        except Exception:
            pass
        return None
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def cpu_baseline(n_atoms, sig, lam, full_M, M_single=100, M_threads=300, overlapped=None):
    """The oracle (NumPy port of the reference's algorithm, kind = "port") timed on the host cores on bounded
    samples of the same workload: M_single training points on ONE thread and M_threads points with the BLAS /
    LAPACK pool unrestricted (all host cores).  MEASURED numbers are reported as they are; the extrapolation to
    the benchmark size (assembly ~ M^2, factorisation ~ M^3) is reported separately and labelled."""
    from oracle import gdml_oracle as orc

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    R, E, F = synth_geometries(n_atoms, max(M_single, M_threads) + 64, seed=0)
    Rf = R.reshape(len(R), -1)
    tp = orc.tril_perms_from_atom_perms(np.arange(n_atoms)[None])
    lin = orc.tril_perms_lin_from_tril_perms(tp)

    def run(m):
        xo, go = orc.desc_from_R(Rf[:m])
        t0 = time.perf_counter()
        K = orc.assemble_K(xo, go, lin, sig)
        t1 = time.perf_counter()
        y = F[:m].ravel() / np.std(F[:m])
        alphas, used_lu = orc.analytic_solve(K, y, lam)
        t2 = time.perf_counter()
        JA = orc.d_desc_dot_vec(go, alphas.reshape(m, -1))
        xq, gq = orc.desc_from_R(Rf[m:m + 64])
        orc.predict_from_desc(xq, gq, xo, JA, tp, sig)
        t3 = time.perf_counter()
        return {'M': m, 'n': m * 3 * n_atoms, 'assemble_s': t1 - t0, 'solve_s': t2 - t1,
                'predict_geoms_per_s': 64.0 / (t3 - t2), 'lu_fallback': bool(used_lu)}

    cores = os.cpu_count() or 1
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            m1 = run(M_single)
    else:
        m1 = run(M_single)
    mt = run(M_threads)  # BLAS/LAPACK threads unrestricted: all host cores

    def extrap(m):
        sc = full_M / float(m['M'])
        return m['assemble_s'] * sc**2 + m['solve_s'] * sc**3

    est1, estt = extrap(m1), extrap(mt)
    best_threads = estt <= est1
    extrapolated = min(est1, estt)
    # the sample that ran beside the GPU work of this very run (start_cpu_sample): the largest one, hence the extrapolation
    # this run stands on -- or, with --cpu-full, the measurement at the benchmark size itself
    in_run = None
    if overlapped is not None and overlapped.get('n_atoms') == n_atoms:
        in_run = dict(overlapped, extrapolated_to_full_s=extrap(overlapped),
                      how='tools/cpu_baseline_full.py {} in a child process while the GPU ran the analytic configs[] entries'.format(overlapped['M']))
        if overlapped['M'] != full_M:
            extrapolated = in_run['extrapolated_to_full_s']
    # the one run of the same code at the benchmark size (tools/cpu_baseline_full.py on the GPU box's host, committed record)
    measured_full = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r03_cpu_baseline_full.json')) as f:
            rec = json.load(f)
        if rec.get('M') == full_M and rec.get('n_atoms') == n_atoms:
            measured_full = dict(rec, provenance='profiles/r03_cpu_baseline_full.json (tools/cpu_baseline_full.py, one run, all cores)')
    except (OSError, ValueError):
        pass
    # the REFERENCE itself beside the port on identical inputs, both thread layouts of BASELINE.md 3.2 (build container:
    # /root/reference does not exist on the GPU box): tools/cpu_ref_vs_port.py -> per-phase ratio reference / port
    ref_vs_port = None
    try:
        points = []
        for fn in ('r04_cpu_reference_vs_port.json', 'r05_cpu_reference_vs_port_m500.json'):  # M = 250 and M = 500
            with open(os.path.join(ROOT, 'profiles', fn)) as f:
                points.append(json.load(f))
        rv = points[-1]  # the larger sample is the one applied (assembly ~M^2 and solve ~M^3 weigh the phases differently)
        b = rv['best_of_layouts']
        ref_vs_port = {'assemble': b['assemble_s']['reference_over_port'], 'solve': b['solve_s']['reference_over_port'],
                       'measured_at': 'N={} M={} on {} cores of the build container, best of the two thread layouts per phase '
                                      '(profiles/r05_cpu_reference_vs_port_m500.json)'.format(rv['n_atoms'], rv['M'], rv['host']['nproc']),
                       'all_points': [{'M': p_['M'], 'assemble': p_['best_of_layouts']['assemble_s']['reference_over_port'],
                                       'solve': p_['best_of_layouts']['solve_s']['reference_over_port']} for p_ in points]}
        src = measured_full if measured_full is not None else \
            {'assemble_s': (m1 if not best_threads else mt)['assemble_s'] * (full_M / float((m1 if not best_threads else mt)['M'])) ** 2,
             'solve_s': (m1 if not best_threads else mt)['solve_s'] * (full_M / float((m1 if not best_threads else mt)['M'])) ** 3}
        ref_vs_port['estimated_reference_build_solve_s'] = src['assemble_s'] * ref_vs_port['assemble'] + \
            src['solve_s'] * ref_vs_port['solve']
        ref_vs_port['note'] = ('estimate: the port\'s phase times at the benchmark size on this box times the measured ratio; the '
                               'reference\'s assembly forks one worker per core and is FASTER than the single-process port')
    except (OSError, ValueError, KeyError):
        pass
    full_in_run = in_run is not None and in_run['M'] == full_M
    return {
        # headline: the number MEASURED at the benchmark size -- in this run (--cpu-full) or the committed record of the same
        # code on the same host type; the samples taken in this run and their extrapolation are the side fields
        'value': in_run['build_solve_s'] if full_in_run else (measured_full['build_solve_s'] if measured_full is not None else extrapolated),
        'value_is': 'measured at the benchmark size in this run' if full_in_run else
                    ('measured at the benchmark size (one committed run); cross-checked by this run\'s samples, see in_run_sample'
                     if measured_full is not None else 'extrapolated from the samples of this run (assembly ~M^2, solve ~M^3)'),
        'in_run_sample': in_run,
        'in_run_sample_vs_committed': None if (in_run is None or measured_full is None) else
        (in_run['build_solve_s'] if full_in_run else in_run['extrapolated_to_full_s']) / measured_full['build_solve_s'] - 1.0,
        'unit': 's',
        'cores': (measured_full['cores'] if measured_full is not None else (cores if best_threads else 1)),
        'kind': 'port',
        'sample': ('oracle/gdml_oracle.py (NumPy restatement of train.py:97-302 + scipy cho_factor/cho_solve); this run: '
                   'M={} (n={}) on 1 thread and M={} (n={}) with the BLAS/LAPACK pool on all {} host cores'
                   ).format(m1['M'], m1['n'], mt['M'], mt['n'], cores),
        'measured_full': measured_full,
        'bounded_sample': {'one_thread': m1, 'all_cores': mt},
        'extrapolated_s': {'one_thread': est1, 'all_cores': estt,
                           'error_vs_measured_full': None if measured_full is None else extrapolated / measured_full['build_solve_s'] - 1.0},
        'reference_over_port': ref_vs_port,
        'predict_geoms_per_s': max(m1['predict_geoms_per_s'] * m1['M'], mt['predict_geoms_per_s'] * mt['M']) / full_M,
        'host': {'nproc': cores},
    }


# ----------------------------------------------------------------------------------------------------------
# BASELINE configs[2]: aspirin-sized N_train = 5000, Nystroem-preconditioned CG (sharded when the context has a
# communicator).  One step = K_nm rows + Nystroem factor + a fixed number of PCG iterations.
def make_cg_workload(ctx, n_atoms, n_train, k_inducing, sig, lam):
    R, E, F = synth_trajectory(n_atoms, n_train, seed=3)  # identical on every rank; the workload configs[2] converges on
    N3 = 3 * n_atoms
    y = F.ravel().copy()
    y /= np.std(y)
    tp = np.arange(n_atoms * (n_atoms - 1) // 2, dtype=np.int64)[None]
    xd, gd = ctx.desc_from_R(R.reshape(n_train, -1), n_atoms)
    ctx.train_upload(xd, gd, tp)
    ctx.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
    # inducing COLUMNS drawn uniformly (what the reference's own first stage does, iterative.py:372-379; rounds 1-4 took all 3N
    # columns of k random points, which gives a numerically rank-deficient K_mm -- min pivot^2 6e-9 -- that no leverage-sampled
    # run produces and on which the fp32 form of the preconditioner declines, profiles/r05_precon_forms.txt)
    idx = np.sort(np.random.RandomState(1).choice(n_train * N3, k_inducing * N3, replace=False)).astype(np.int64)
    return {'y': y, 'idx': idx, 'sig': sig, 'lam': lam, 'n': n_train * N3, 'm': idx.size}


def cg_step(ctx, wl, n_iters):
    ctx.assemble_K(wl['sig'], False, idx=wl['idx'], alloc_extra_rows=wl['m'])
    _, _, info = ctx.nystroem_factor(wl['lam'], wl['idx'], want_lev=False)
    wl['precon_form'] = {0: 'stored fp64 factor', 2: 'matrix-free', 4: 'fp32 factor + Gram correction'}[info & 6]
    x, info, iters, resid = ctx.pcg(wl['lam'], False, wl['y'], rtol=0.0, maxiter=n_iters)  # rtol 0: exactly n_iters
    assert iters == n_iters, (iters, n_iters)
    return resid, {k: ctx.phase_ms(k)[0] for k in ('assemble', 'precon', 'pcg')}


def time_cg(ctx, wl, n_iters, steps, warmup, barrier):
    for _ in range(warmup):
        cg_step(ctx, wl, n_iters)
    barrier()
    c0, b0 = ctx.comm_stats()
    t0 = time.perf_counter()
    ph = []
    for _ in range(steps):
        resid, p = cg_step(ctx, wl, n_iters)
        ph.append(p)
    barrier()
    t1 = time.perf_counter()
    c1, b1 = ctx.comm_stats()
    phases = {k: float(np.mean([p[k] for p in ph])) for k in ph[0]}
    # the preconditioner GEMVs' own time (HIP events around the kernels, outside the timed region): what `roofline` is quoted on
    ctx.profile(True)
    cg_step(ctx, wl, n_iters)
    g_ms, g_n, g_by = ctx.kernel_stat('precon_gemv')
    ctx.profile(False)
    barrier()
    return {'s_per_step': (t1 - t0) / max(1, steps), 'phases_ms': phases, 'ms_per_pcg_iteration': phases['pcg'] / n_iters,
            'gemv': {'ms_per_application': g_ms / max(1, g_n), 'bytes_per_application': g_by / max(1, g_n), 'applications': g_n},
            'resid_over_norm_y': float(resid / np.linalg.norm(wl['y'])),
            'collectives_per_step': (c1 - c0) / max(1, steps), 'collective_bytes_per_step_per_rank': (b1 - b0) / max(1, steps)}


def cg_algorithmic_bytes(wl, n_iters, world):
    """HBM bytes one rank's dominant kernels must move per step: the preconditioner factor X (n/W x m fp64) is read
    twice per PCG iteration (X^T v, then X t), K_nm is written once; the mat-vec tables are cache resident."""
    x_bytes = 8.0 * wl['n'] * wl['m'] / world
    return x_bytes * (1 + 2 * n_iters), 2.0 * x_bytes


def sigma_sweep_config0(n_train=200, n_valid=1000, n_test=5000, sigs=None):
    """BASELINE configs[0] shape (`sgdml all <9 atoms> 200 1000 5000`): N=9, P=6, the whole train -> validate ->
    select -> test loop of sgdml_amd.sweep on synthetic geometries."""
    from sgdml_amd.sweep import sigma_sweep
    from sgdml_amd.train import GDMLTrain

    N = 9
    n_all = n_train + n_valid + n_test + 64
    R, E, F = synth_geometries(N, n_all, seed=5)
    ds = {'type': 'd', 'name': np.array('synth9'), 'theory': np.array('pair'), 'z': np.array([6, 6, 8, 1, 1, 1, 1, 1, 1]),
          'R': R, 'E': E, 'F': F}
    perms = [tuple(range(N))]
    gens = [(1, 2, 0, 3, 4, 5, 6, 7, 8), (0, 1, 2, 4, 3, 5, 6, 7, 8)]
    frontier = list(perms)
    while frontier:
        nxt = []
        for a in frontier:
            for g in gens:
                c = tuple(a[i] for i in g)
                if c not in perms:
                    perms.append(c)
                    nxt.append(c)
        frontier = nxt
    # The sweep runs twice: HIP loads a kernel's code object at its first launch in a process (deferred loading), and this
    # sweep is the first user of a dozen kernels (small-n factorisation chain, permutation-group assembly, error sums):
    # 0.65 s of first-touch cost on a 0.3 s sweep (profiles/r04_sweep_hostprof.txt).  Both numbers are reported.
    walls = []
    for rep in range(2):
        np.random.seed(0)
        tr = GDMLTrain()
        try:
            t0 = time.perf_counter()
            best, table, tm = sigma_sweep(tr, ds, n_train, n_valid, n_test, sigs=sigs, perms=np.array(perms), early_stop=False)
            walls.append((time.perf_counter() - t0, tm['train_s']))
        finally:
            tr.__del__()
    wall = walls[-1][0]
    geoms = tm['n_models'] * tm['n_valid'] + tm['n_test']
    return {
        'config': 'configs[0] shape: N=9 P={} N_train={} sigma grid of {} models, {} validation + {} test geometries '
                  '(sgdml_amd.sweep.sigma_sweep = the loop of `sgdml all`)'.format(len(perms), n_train, tm['n_models'],
                                                                                 tm['n_valid'], tm['n_test']),
        'wall_s': wall, 'first_call_in_process': {'wall_s': walls[0][0], 'train_s': walls[0][1]},
        'create_task_s': tm['create_task_s'], 'train_s': tm['train_s'],
        'validate_s': tm['validate_s'], 'test_s': tm['test_s'],
        'validate_test_geoms_per_s': geoms / max(1e-9, tm['validate_s'] + tm['test_s']),
        'best_sig': float(best['sig']), 'best_f_rmse': best['f_err']['rmse'],
        'valid_f_rmse_by_sig': {str(r[0]): r[4] for r in table},
    }


def perm_group(n_atoms, kind):
    """Permutation groups of the BASELINE shapes: 'c3x3' = three independent 3-cycles (27 elements, configs[3]),
    'c2x2' = two independent swaps (4 elements); None = identity."""
    if not kind:
        return np.arange(n_atoms)[None, :]
    gens = []
    if kind == 'c3x3':
        for a in (3, n_atoms // 2 - 4, n_atoms - 12):
            g = list(range(n_atoms))
            g[a], g[a + 1], g[a + 2] = a + 1, a + 2, a
            gens.append(tuple(g))
    elif kind == 'c2x2':
        for a in (2, n_atoms - 5):
            g = list(range(n_atoms))
            g[a], g[a + 1] = a + 1, a
            gens.append(tuple(g))
    else:
        raise ValueError(kind)
    perms = [tuple(range(n_atoms))]
    frontier = list(perms)
    while frontier:
        nxt = []
        for a in frontier:
            for g in gens:
                c = tuple(a[i] for i in g)
                if c not in perms:
                    perms.append(c)
                    nxt.append(c)
        frontier = nxt
    return np.array(perms)


def assembly_kernel_name(n_atoms, n_perms):
    """Which assembly kernel the library dispatches a dense (all columns / lower form) build of this shape to -- a label for the
    per-config `roofline_assemble` entries; the rule itself lives in csrc/assemble.hip, assemble_perm.hip, assemble_perm2.hip
    (tests/test_host_policy_cpu.py checks that the two agree)."""
    if n_perms == 1 and n_atoms <= 21:
        return 'assemble_strip_kernel' if n_atoms >= 11 else 'assemble_wave_kernel'
    if n_perms == 1 and 22 <= n_atoms <= 256:
        return 'assemble_big1_kernel (P = 1: rank-one term + one 3 x 3 outer product per atom pair; one workgroup per block pair)'
    if n_perms > 1 and 8 <= n_atoms <= 24:
        return 'assemble_pts_kernel'
    if n_atoms <= 42 and ((n_perms >= 16 and n_atoms >= 36) or (n_perms >= 6 and n_atoms >= 40)):
        return 'assemble_perm2_kernel (fp64-MFMA outer products, fixed-atom split; lower blocks)'
    return 'assemble_perm_kernel'


def solve_config(label, n_atoms, n_train, perms_kind=None, solver='analytic', sig=20, lam=1e-10, max_memory=None, seed=3,
                 traj=None, n_inducing=None, dist_backend=None, options=None, policy=None):
    """One BASELINE configuration shape run to a SOLUTION through the drop-in GDMLTrain.train (sgdml/train.py:836-1088):
    analytic = assemble + Cholesky + solves; cg = the reference's iterative policy (leverage-score inducing points,
    Nystroem preconditioner, PCG to solver_tol = 1e-4, restarts) -- wall-clock to the converged model, phases,
    iterations, and the residual of the returned coefficients through the matrix-free operator.
    dist_backend ('rccl' / 'host'): every rank of the job calls this (host group: sgdml_amd.dist.host_group); the solve is
    sharded over them (GDMLTrain.init_distributed)."""
    from sgdml_amd.solvers.iterative import Iterative
    from sgdml_amd.train import GDMLTrain

    if traj is not None:
        R, E, F = synth_trajectory(n_atoms, n_train, seed=seed, **traj)
    else:
        R, E, F = synth_geometries(n_atoms, n_train, seed=seed)
    perms = perm_group(n_atoms, perms_kind)
    task = {'type': 't', 'code_version': 'bench', 'dataset_name': np.array('synth'), 'dataset_theory': np.array('pair'),
            'z': np.full(n_atoms, 6), 'R_train': R, 'F_train': F, 'E_train': E, 'idxs_train': np.arange(n_train),
            'md5_train': 'x', 'idxs_valid': np.arange(0), 'md5_valid': 'x', 'sig': sig, 'lam': lam, 'use_E': True,
            'use_E_cstr': False, 'use_sym': perms.shape[0] > 1, 'perms': perms}
    n = 3 * n_atoms * n_train
    draws = []
    orig = Iterative.inducing_pts_from_lev_scores

    def spy(self, lev, m):
        idx = orig(self, lev, m)
        draws.append(len(idx) // (3 * n_atoms))
        return idx

    Iterative.inducing_pts_from_lev_scores = spy
    tr = GDMLTrain(max_memory=max_memory)
    try:
        tr._force_solver = solver
        tr._force_n_inducing_pts = n_inducing
        if policy is not None:
            tr.inducing_pts_policy = policy
        ctx = tr._context()
        for k_, v_ in (options or {}).items():
            ctx.set_option(k_, v_)
        if dist_backend is not None:
            tr.init_distributed(backend=dist_backend)
        ctx.profile(True)
        np.random.seed(seed)
        ctx.sync()
        t0 = time.perf_counter()
        model = tr.train(task)
        ctx.sync()
        wall = time.perf_counter() - t0
        out = {'config': label, 'n_atoms': n_atoms, 'n_train': n_train, 'n_perms': int(perms.shape[0]), 'matrix_n': n,
               'solver': solver, 'sig': sig, 'lam': lam, 'train_wall_s': wall}
        ph = {}
        for k in ('assemble', 'factor', 'solve', 'precon', 'pcg'):
            try:
                ph[k] = ctx.phase_ms(k)[0]
            except Exception:  # phase never ran on this solver branch
                pass
        out['phases_ms_last'] = ph
        if solver == 'analytic':
            out['wall_minus_phases_s'] = wall - sum(ph.values()) / 1e3  # host work + allocation outside the kernels
        a_ms, a_n, a_by = ctx.kernel_stat('assemble')
        if a_ms > 0 and solver == 'analytic':
            a_kernel = assembly_kernel_name(n_atoms, int(perms.shape[0]))
            out['roofline_assemble'] = {'kernel': a_kernel, 'bound': 'hbm', 'achieved': a_by / (a_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                        'frac': a_by / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'ms': a_ms / a_n,
                                        'algorithmic_bytes_per_launch': a_by / a_n,
                                        'fp64_valu_model_TFLOPs': n_train * (n_train + 1) / 2.0 * perms.shape[0] *
                                        (2.0 * (3 * n_atoms) ** 2 + 100.0 * n_atoms * (n_atoms - 1) / 2) / (a_ms / a_n * 1e-3) / 1e12}
        if solver == 'analytic' and 'factor' in ph:
            out['cholesky_TFLOPs'] = n**3 / 3.0 / (ph['factor'] * 1e-3) / 1e12
        y = F.ravel() / np.std(F.ravel())
        from sgdml_amd.utils.desc import Desc
        tril = np.array([Desc.perm(p_) for p_ in perms])
        xd, gd = ctx.desc_from_R(R.reshape(n_train, -1), n_atoms)
        ctx.train_upload(xd, gd, tril)
        ctx.predict_upload_model(xd, np.zeros_like(xd), tril, sig, None)
        Kv = ctx.kernel_matvec(lam, False, -np.asarray(model['alphas_F']).ravel())
        out['resid_over_norm_y'] = float(np.linalg.norm(-Kv - y) / np.linalg.norm(y))
        if solver == 'cg':
            out.update({'time_to_tol_s': wall, 'solver_tol': float(model['solver_tol']), 'solver_iters': int(model['solver_iters']),
                        'converged': bool(model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']),
                        'inducing_pts_per_stage': draws, 'restarts': max(0, len(draws) - 1),
                        'inducing_pts_policy': 'forced' if n_inducing else tr.inducing_pts_policy,
                        'precon_form': getattr(tr, '_last_precon_form', None),
                        'f32_gram_min_pivot': ctx.get_option('pcg.f32_last_min_pivot'),
                        'ms_per_pcg_iteration': ph.get('pcg', 0.0) / max(1, int(model['solver_iters']))})
        return out
    finally:
        Iterative.inducing_pts_from_lev_scores = orig
        tr.__del__()


def load_pmc_traffic():
    """HBM traffic of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected
    as MI355X_MICROARCH.md prescribes), committed by tools/pmc_traffic.py as profiles/hbm_traffic.json."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--n-atoms', type=int, default=21)
    ap.add_argument('--n-train', type=int, default=1000)
    ap.add_argument('--n-query', type=int, default=1000)
    ap.add_argument('--sig', type=float, default=20.0)
    ap.add_argument('--lam', type=float, default=1e-10)
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU-baseline leg')
    ap.add_argument('--cpu-sample', type=int, default=500, help='training points of the CPU sample that runs beside the configs[] section')
    ap.add_argument('--cpu-full', action='store_true', help='time the CPU port at the benchmark size in this run (4-5 minutes of host time, overlapped with the GPU work)')
    ap.add_argument('--no-configs', action='store_true', help='N=1: skip the configs[0] sweep and the configs[2] point')
    ap.add_argument('--no-profile', action='store_true', help='do not bracket kernels with HIP events')
    ap.add_argument('--separate-solve', action='store_true',
                    help='A/B: forward substitution as a separate triangular solve instead of inside the factorisation')
    ap.add_argument('--cg-n-train', type=int, default=5000)
    ap.add_argument('--cg-inducing', type=int, default=200)
    ap.add_argument('--cg-iters', type=int, default=50, help='PCG iterations per step of the configs[2] workload')
    ap.add_argument('--no-to-tol', action='store_true', help='N>1: skip the sharded run to solver_tol')
    ap.add_argument('--dist-chol-timeout', type=float, default=900.0, help='N>1: seconds the distributed-Cholesky measurement may take')
    ap.add_argument('--to-tol-timeout', type=float, default=600.0, help='N>1: seconds the sharded run to solver_tol may take')
    ap.add_argument('--dist-chol', action='store_true', help='N>1: also time the configs[1] system through the distributed Cholesky')
    ap.add_argument('--workload', default='auto', choices=('auto', 'analytic', 'cg'),
                    help="what `value` measures, the SAME for every --gpus N: analytic (default) = configs[1] build+solve (N>1: "
                         "through the distributed Cholesky); cg = the configs[2] sharded Nystroem-PCG step")
    ap.add_argument('--no-cg-extras', action='store_true', help='N>1 analytic: skip the configs[2] step / time-to-tol blocks')
    ap.add_argument('--comm', default='auto', help="N>1: 'rccl', 'host' (collectives staged through the host channel), or auto (rccl if every rank has a GPU)")
    args = ap.parse_args()

    # ---- `--gpus N` with N > 1 and no launcher around us: become the launcher (one rank per GPU over RCCL)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) != args.gpus:
        sys.exit('bench.py: --gpus {} but the launcher started {} ranks'.format(args.gpus, os.environ['WORLD_SIZE']))

    if int(os.environ.get('LOCAL_RANK', '0')) == 0 and 'GDML_BENCH_CHILD' not in os.environ:
        try:  # a child process takes the first touch of the GPU (see sgdml_amd._lib.preflight)
            from sgdml_amd import _lib as _pre

            _pre.preflight()
        except Exception as e:
            sys.stderr.write('bench.py: GPU preflight: %r\n' % (e,))
    # stdout carries exactly one line (the JSON): libraries that print banners to fd 1 (gloo's rank
    # messages, RCCL's version block) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    workload = 'analytic' if args.workload == 'auto' else args.workload
    if world == 1 and workload == 'analytic':
        out = run_analytic(args)
    else:
        out = run_multi(args, rank, world, workload)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
        stuck = [out.get(k) for k in ('time_to_tol', 'dist_cholesky')] if isinstance(out, dict) else []
        if any(isinstance(x, dict) and str(x.get('error', '')).startswith('no result within') for x in stuck):
            os._exit(0)  # a worker thread is stuck inside a collective: do not wait for it at interpreter exit


def self_launch(args):
    """python bench.py --gpus N (N > 1) without a launcher: start N copies of this script, one rank per GPU on 127.0.0.1.  With fewer GPUs than ranks the run is refused unless `--comm host` asks for the functional mode
    (ranks share GPUs, collectives staged through the host)."""
    import socket
    import subprocess

    from sgdml_amd import _lib

    try:
        n_dev = _lib.device_count()
    except Exception as e:  # no library / no driver
        sys.stderr.write('bench.py: --gpus {}: cannot count GPUs ({})\n'.format(args.gpus, e))
        return 2
    if n_dev < args.gpus and args.comm != 'host':
        sys.stderr.write('bench.py: --gpus {} but {} GPU(s) visible; pass --comm host for a functional run with ranks sharing '
                         'GPUs\n'.format(args.gpus, n_dev))
        return 2
    try:  # a child process takes the first touch of the GPU (see sgdml_amd._lib.preflight); the ranks then skip theirs
        _lib.preflight()
    except Exception as e:
        sys.stderr.write('bench.py: GPU preflight: %r\n' % (e,))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    # one rank per GPU, the environment a launcher would set (what the driver's torch.distributed.run provides); the ranks
    # rendezvous through sgdml_amd.hostchannel, so no launcher package is needed
    from sgdml_amd.hostchannel import new_token

    procs = []
    token = os.environ.get('GDML_CHANNEL_TOKEN') or new_token()  # per-job secret of the host channel's handshake
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   GDML_BENCH_CHILD='1', GDML_CHANNEL_TOKEN=token)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = p_.wait() or rc
    return rc


def run_multi(args, rank, world, workload):
    """N > 1 (and N = 1 with --workload cg): one process per GPU, RCCL inside the library; the host side (rendezvous, barrier,
    max over ranks) is sgdml_amd.hostchannel: no PyTorch in these processes.
    workload 'analytic' (default): `value` = build+solve seconds of the configs[1] system through the distributed Cholesky --
    the SAME metric and workload string as the N = 1 line (line_skeleton), strong scaling; the configs[2] step and its run to
    solver_tol ride along as `cg` / `time_to_tol`.  workload 'cg': `value` = seconds per configs[2] step (the round 1-5 line)."""
    from sgdml_amd import _lib
    from sgdml_amd.dist import host_group, init_comm, pick_backend, probe_rccl

    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    hg = host_group()
    n_dev = _lib.device_count()
    comm = args.comm
    rccl_probe = None
    if world > 1:
        if comm == 'auto':  # RCCL when every rank has a physical GPU of its own; ranks sharing a GPU: functional host-staged run
            comm = pick_backend(local_rank % max(1, n_dev), hg)
        if comm == 'rccl' and args.comm == 'auto':
            # a toy sharded solve through RCCL in child processes, under a time limit, before the real ranks commit to it: a
            # node where RCCL does not come up still gets its (slower) strong-scaling line through the host-staged collectives
            ok, rccl_probe = probe_rccl(local_rank % max(1, n_dev), hg)
            if not ok:
                comm = 'host'
                sys.stderr.write('bench.py: RCCL probe failed ({}); using host-staged collectives\n'.format(rccl_probe))
    else:
        comm = 'none'
    ctx = _lib.Context(local_rank % max(1, n_dev))
    rccl_ranks_seen = None
    if world > 1:
        init_comm(ctx, group=hg, backend=comm)
        rccl_ranks_seen = ctx.get_option('comm.rccl_ranks')  # ncclCommCount of the communicator the library built (None: host-staged)
        rccl_ranks_seen = None if rccl_ranks_seen is None else int(rccl_ranks_seen)

    def barrier():
        ctx.sync()
        hg.barrier()

    def max_over_ranks(values):
        every = hg.allgather_obj([float(v) for v in values])
        return [max(e[i] for e in every) for i in range(len(values))]

    N, M, k = args.n_atoms, args.cg_n_train, args.cg_inducing
    with_cg = workload == 'cg' or not args.no_cg_extras
    with_chol = world > 1 and (workload == 'analytic' or args.dist_chol)

    # ---- configs[1] (n = 63 000) through the distributed Cholesky over the communicator: the `value` of the default N > 1
    # line.  Timed like the N = 1 headline: W untimed solves, then K solves between barriers, max over ranks; one solve =
    # assemble this rank's row blocks of A = -K + lam I + factor + both substitutions.  Default schedule (dist.lookahead
    # unset: one panel of look-ahead from two ranks on), then the other schedule for comparison (3 repetitions).
    dchol = None
    chol_hung = False
    if with_chol:
        # in a worker thread under a time limit (--dist-chol-timeout): the RCCL branch of the distributed Cholesky has never run
        # on more than one physical GPU; should a collective hang there, a line must still be printed (value null + the error)
        # instead of the driver's timeout killing the run
        import threading

        cbox = {}

        def _chol():
            try:
                Mc = args.n_train
                Rc, Ec, Fc = synth_geometries(N, Mc, seed=0)
                yc = Fc.ravel() / np.std(Fc)
                xdc, gdc = ctx.desc_from_R(Rc.reshape(Mc, -1), N)
                tpc = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
                ctx.train_upload(xdc, gdc, tpc)
                la_default = 1 if world > 1 else 0

                def residual(a_):
                    ctx.predict_upload_model(xdc, np.zeros_like(xdc), tpc, args.sig, None)
                    Kv = ctx.kernel_matvec(args.lam, False, -a_)
                    ctx.train_upload(xdc, gdc, tpc)
                    return float(np.linalg.norm(-Kv - yc) / np.linalg.norm(yc))

                # (1) the in-order schedule first (one stream, one communicator: what needs least from RCCL) -- if the look-ahead
                # schedule below should not come back on some node, this is the number the line falls back to
                ctx.set_option('dist.lookahead', 0)
                ts = []
                for rep in range(3):
                    barrier()
                    t0 = time.perf_counter()
                    a_c = ctx.dist_chol_solve(args.sig, args.lam, yc)
                    barrier()
                    ts.append(time.perf_counter() - t0)
                (t_in,) = max_over_ranks([min(ts[1:])])
                ph_in = {k_: max_over_ranks([ctx.phase_ms(k_)[0]])[0] for k_ in ('assemble', 'factor', 'solve')}
                cbox['inorder'] = {'dist.lookahead': 0, 's_per_solve': t_in, 'phases_ms': ph_in,
                                   'build_solve_s': (ph_in['assemble'] + ph_in['factor'] + ph_in['solve']) / 1e3,
                                   'solve_rel_residual': residual(a_c)}
                # (2) the default schedule (one panel of look-ahead from two ranks on): W untimed, K timed solves
                ctx.set_option('dist.lookahead', la_default)
                for _ in range(max(1, args.warmup) if workload == 'analytic' else 1):
                    ctx.dist_chol_solve(args.sig, args.lam, yc)
                ctx.profile(True)
                barrier()
                t0 = time.perf_counter()
                ph = []
                n_steps = args.steps if workload == 'analytic' else 2
                for _ in range(n_steps):
                    a_c = ctx.dist_chol_solve(args.sig, args.lam, yc)
                    ph.append({k_: ctx.phase_ms(k_)[0] for k_ in ('assemble', 'factor', 'solve')})
                barrier()
                wall = (time.perf_counter() - t0) / max(1, n_steps)
                g_ms, g_n, g_fl = ctx.kernel_stat('gemm_nt_sub')
                ctx.profile(False)
                (wall, g_ms_max) = max_over_ranks([wall, g_ms])
                phases = {k_: max_over_ranks([float(np.mean([p_[k_] for p_ in ph]))])[0] for k_ in ph[0]}
                resid = residual(a_c)
                ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
                dchol_ = {'s_per_solve': wall, 'build_solve_s': (phases['assemble'] + phases['factor'] + phases['solve']) / 1e3,
                          'phases_ms': phases, 'solve_rel_residual': resid, 'matrix_n': Mc * 3 * N,
                          'matrix_bytes_per_rank': ctx.mem_info()[0],
                          'schedule': 'dist.lookahead = %d (library default for %d rank%s)' % (la_default, world, '' if world == 1 else 's'),
                          'roofline': {'kernel': 'gemm_nt_sub_kernel (fp64 MFMA trailing updates of this rank\'s row blocks, K = 512, '
                                                 'block-cyclic lower tile predicate)', 'bound': 'mfma', 'achieved': ach,
                                       'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': ach / FP64_MFMA_PEAK_TF, 'traffic': None,
                                       'launches': g_n, 'avg_launch_ms': g_ms / max(1, g_n), 'rank': 0,
                                       'kernel_ms_per_solve_max_over_ranks': g_ms_max / max(1, n_steps)},
                          'other_schedule': cbox['inorder']}
                cbox['r'] = dchol_
            except Exception as e:  # a line must still be printed
                cbox['r'] = {'error': repr(e)}

        th_c = threading.Thread(target=_chol, daemon=True)
        th_c.start()
        th_c.join(timeout=args.dist_chol_timeout)
        chol_hung = th_c.is_alive()
        if chol_hung and 'inorder' in cbox:  # the look-ahead schedule did not come back: the in-order measurement stands in
            io = cbox['inorder']
            dchol = {'s_per_solve': io['s_per_solve'], 'build_solve_s': io['build_solve_s'], 'phases_ms': io['phases_ms'],
                     'solve_rel_residual': io['solve_rel_residual'], 'matrix_n': args.n_train * 3 * N,
                     'schedule': 'dist.lookahead = 0: the default (look-ahead) schedule did not return within {} s'.format(args.dist_chol_timeout),
                     'roofline': {'kernel': 'gemm_nt_sub_kernel', 'bound': 'mfma', 'achieved': None, 'peak': FP64_MFMA_PEAK_TF,
                                  'unit': 'TFLOP/s', 'frac': None, 'traffic': None},
                     'error': 'no result within {} s (look-ahead schedule)'.format(args.dist_chol_timeout)}
        else:
            dchol = {'error': 'no result within {} s'.format(args.dist_chol_timeout)} if chol_hung else cbox.get('r')
        if chol_hung:  # the context is stuck inside a collective: nothing more can run on it
            with_cg = False

    # ---- configs[2]: K_nm rows + preconditioner + a fixed number of PCG iterations
    res = wl = None
    s_per_step = asm_ms = pre_ms = pcg_ms = gemv_ms = None
    if with_cg:
        try:
            wl = make_cg_workload(ctx, N, M, k, args.sig, args.lam)
            res = time_cg(ctx, wl, args.cg_iters, args.steps if workload == 'cg' else 2, args.warmup if workload == 'cg' else 1, barrier)
            s_per_step, asm_ms, pre_ms, pcg_ms, gemv_ms = max_over_ranks(
                [res['s_per_step'], res['phases_ms']['assemble'], res['phases_ms']['precon'], res['phases_ms']['pcg'],
                 res['gemv']['ms_per_application']])
        except Exception as e:
            if workload == 'cg':
                raise
            res = None
            wl = {'error': repr(e)}
    if not chol_hung:
        ctx.close()

    # the same workload as the 1-GPU configs[2] entry run to solver_tol through GDMLTrain.train, sharded over the ranks
    # (leverage-sampled inducing points, the reference's restart policy): what a SCALE record means as a SOLVE
    to_tol = None
    hung = chol_hung
    if with_cg and not args.no_to_tol:
        # in a worker thread under a time limit: this leg builds a second communicator and runs several hundred sharded
        # iterations; should a collective ever hang on some node, the line above must still be printed (the main
        # thread gives up waiting, reports it, and the process leaves through os._exit)
        import threading

        box = {}

        def _run():
            try:
                box['r'] = solve_config('configs[2] to solver_tol 1e-4, sharded over {} ranks'.format(world), N, M, solver='cg',
                                        max_memory=32, traj={'n_modes': 8, 'amp': 0.15, 'noise': 0.01}, sig=args.sig,
                                        lam=args.lam, dist_backend=comm if world > 1 else None)
            except Exception as e:
                box['r'] = {'error': repr(e)}

        th = threading.Thread(target=_run, daemon=True)
        th.start()
        th.join(timeout=args.to_tol_timeout)
        hung = th.is_alive()
        to_tol = {'error': 'no result within {} s'.format(args.to_tol_timeout)} if hung else box.get('r')
        if not hung and to_tol is not None and 'train_wall_s' in to_tol:
            (tw,) = max_over_ranks([to_tol['train_wall_s']])
            to_tol['time_to_tol_s'] = to_tol['train_wall_s'] = tw

    one_gpu = None
    if rank == 0 and not hung and world > 1 and res is not None:  # the 1-GPU point of the configs[2] curve, same run, rank 0's GPU
        c1 = _lib.Context(local_rank % max(1, n_dev))
        wl1 = make_cg_workload(c1, N, M, k, args.sig, args.lam)
        one_gpu = time_cg(c1, wl1, args.cg_iters, 1, 1, c1.sync)
        c1.close()
    cpu = None
    if rank == 0 and not hung and not args.no_cpu:
        try:
            cpu = cpu_baseline(N, args.sig, args.lam, args.n_train)
        except Exception as e:
            cpu = {'error': repr(e)}
    if not hung:
        hg.barrier()
        hg.close()
    if rank != 0:
        if hung:
            os._exit(0)
        return None
    comm_name = {'rccl': 'RCCL', 'host': 'host-staged collectives over TCP', 'none': 'none (one rank)'}[comm]
    cg_block = None
    if res is not None:
        gemv_bytes = res['gemv']['bytes_per_application']
        ach = gemv_bytes / (gemv_ms * 1e-3) / 1e9 if gemv_ms and gemv_ms > 0 else 0.0
        cg_block = {
            'workload': line_skeleton('cg', N, M, world, args.cg_iters, k)['workload'],
            's_per_step': s_per_step, 'phases_ms': {'assemble': asm_ms, 'precon': pre_ms, 'pcg': pcg_ms},
            'ms_per_pcg_iteration': pcg_ms / args.cg_iters, 'n_inducing_points': k, 'matrix_n': wl['n'], 'precon_m': wl['m'],
            'pcg_iterations_per_step': args.cg_iters, 'preconditioner_form': wl.get('precon_form'),
            'collectives_per_step': res['collectives_per_step'],
            'collective_bytes_per_step_per_rank': res['collective_bytes_per_step_per_rank'],
            'resid_over_norm_y': res['resid_over_norm_y'],
            'one_gpu_s_per_step': None if one_gpu is None else one_gpu['s_per_step'], 'one_gpu': one_gpu,
            'roofline': {'kernel': 'gemv_t_part + gemv_n_precon kernels (the two passes over this rank\'s rows of the preconditioner '
                                   'factor in one application, incl. the m-vector all-reduce between them): ' + str(wl.get('precon_form')),
                         'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                         'traffic': None, 'avg_application_ms': gemv_ms, 'algorithmic_bytes_per_application': gemv_bytes,
                         'note': 'achieved = algorithmic bytes of one application (2 x n/W x m x 8 B, 4 B in the fp32 form, + the '
                                 'm x m correction) / the kernels\' own time from HIP events (gdml_kernel_stat), max over ranks: '
                                 'comparable with the N = 1 record'}}
    elif wl is not None:
        cg_block = wl  # {'error': ...}
    sk = line_skeleton(workload, N, args.n_train if workload == 'analytic' else M, world, args.cg_iters, k)
    common = {'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': False, 'scaling': 'strong',
              'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic'}
    cfg = {'workload': sk['workload'], 'n_atoms': N, 'sig': args.sig, 'lam': args.lam, 'parallelism': sk['parallelism'],
           'collectives': comm, 'collectives_name': comm_name, 'rccl_probe': rccl_probe, 'rccl_ranks_seen': rccl_ranks_seen,
           'host_side': 'sgdml_amd.hostchannel (no PyTorch in the ranks)'}
    if workload == 'analytic':
        ok = isinstance(dchol, dict) and 'build_solve_s' in dchol
        if ok and not dchol['solve_rel_residual'] < 1e-10:
            raise SystemExit('bench: residual of the timed distributed solve is %.3e (> 1e-10): refusing to report' % dchol['solve_rel_residual'])
        cfg.update({'n_train': args.n_train, 'n_perms': 1, 'matrix_n': args.n_train * 3 * N})
        out = dict(common, metric=sk['metric'], value=dchol['build_solve_s'] if ok else None, unit='s',
                   ms_per_step=dchol['s_per_solve'] * 1e3 if ok else None, config=cfg,
                   phases_ms=dchol.get('phases_ms') if ok else None, solve_rel_residual=dchol.get('solve_rel_residual') if ok else None,
                   roofline=dchol.get('roofline') if ok else None, dist_cholesky=dchol, cg=cg_block, time_to_tol=to_tol,
                   cpu_baseline=cpu)
        out['scale_point'] = {'workload': sk['workload'], 'seconds': out['value'], 'n_gpus': world,
                              'roofline_frac': None if not ok else dchol['roofline']['frac']}
        if cg_block is not None and 's_per_step' in cg_block:
            out['scale_point_cg'] = {'workload': cg_block['workload'], 'seconds': cg_block['s_per_step'], 'n_gpus': world,
                                     'time_to_tol_s': None if not isinstance(to_tol, dict) else to_tol.get('time_to_tol_s', to_tol.get('train_wall_s'))}
        return out
    cfg.update({'n_train': M, 'n_inducing_points': k, 'matrix_n': wl['n'], 'precon_m': wl['m'],
                'pcg_iterations_per_step': args.cg_iters, 'preconditioner_form': wl.get('precon_form')})
    out = dict(common, metric=sk['metric'], value=s_per_step, unit='s', ms_per_step=s_per_step * 1e3, config=cfg,
               phases_ms=cg_block['phases_ms'], ms_per_pcg_iteration=cg_block['ms_per_pcg_iteration'],
               collectives_per_step=cg_block['collectives_per_step'],
               collective_bytes_per_step_per_rank=cg_block['collective_bytes_per_step_per_rank'],
               resid_over_norm_y=cg_block['resid_over_norm_y'], one_gpu_s_per_step=cg_block['one_gpu_s_per_step'], one_gpu=one_gpu,
               time_to_tol=to_tol, dist_cholesky=dchol, roofline=cg_block['roofline'], cpu_baseline=cpu)
    out['scale_point_cg'] = {'workload': cg_block['workload'], 'seconds': s_per_step, 'n_gpus': world,
                             'time_to_tol_s': None if not isinstance(to_tol, dict) else to_tol.get('time_to_tol_s', to_tol.get('train_wall_s'))}
    if isinstance(dchol, dict) and 'build_solve_s' in dchol:
        out['scale_point'] = {'workload': line_skeleton('analytic', N, args.n_train, world)['workload'], 'seconds': dchol['build_solve_s'],
                              'n_gpus': world, 'roofline_frac': dchol['roofline']['frac']}
    return out


def predict_sweep(ctx, lib, Rq_all, N, M, P, n_dev_reps=20):
    """Prediction throughput of the resident model at B = 1, 64 (host arrays in, host E / F out: what the ASE calculator and a
    small validation loop pay, latency bound) and B = 1000, 16384 (device-resident queries and results: the contraction itself).
    Roofline per SURVEY.md 8(d): one query costs ~10 M P D flops and, streamed once, 16 M P D bytes of table; a batch shares
    the table, so large batches are bound by the fp64 pipes and a single query by HBM (in practice by launch latency)."""
    D = N * (N - 1) // 2
    N3 = 3 * N
    flops_q = 10.0 * M * P * D
    bytes_t = 16.0 * M * P * D
    out = []
    for B in (1, 64):
        q = np.ascontiguousarray(Rq_all[:B])
        for _ in range(30):
            ctx.predict(q)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            ctx.predict(q)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        out.append({'batch': B, 'path': 'host arrays in / out (gdml_predict' + (', single launch)' if B <= 8 else ')'),
                    'latency_us': t * 1e6, 'geoms_per_s': B / t, 'forces_per_s': B * N3 / t,
                    'algorithmic_TFLOPs': flops_q * B / t / 1e12, 'table_GBs': bytes_t / t / 1e9})
    for B in (1000, 16384):
        reps = -(-B // Rq_all.shape[0])
        Rq = np.ascontiguousarray(np.tile(Rq_all, (reps, 1))[:B])
        dR, dE, dF = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ctx._check(lib.gdml_dev_alloc(ctx._h, B * N3 * 8, C.byref(dR)))
        ctx._check(lib.gdml_dev_alloc(ctx._h, B * 8, C.byref(dE)))
        ctx._check(lib.gdml_dev_alloc(ctx._h, B * N3 * 8, C.byref(dF)))
        ctx._check(lib.gdml_memcpy_h2d(ctx._h, dR, Rq.ctypes.data_as(C.c_void_p), Rq.nbytes))
        ms = []
        for rep in range(n_dev_reps + 3):
            ctx._check(lib.gdml_predict_dev(ctx._h, dR, B, None, None, dE, dF))
            if rep >= 3:
                ms.append(ctx.phase_ms('predict')[0])
        for p_ in (dR, dE, dF):
            ctx._check(lib.gdml_dev_free(ctx._h, p_))
        t = float(np.median(ms)) * 1e-3
        out.append({'batch': B, 'path': 'device-resident queries and results (gdml_predict_dev: descriptors + contraction + epilogue)',
                    'ms': t * 1e3, 'geoms_per_s': B / t, 'forces_per_s': B * N3 / t,
                    'algorithmic_TFLOPs': flops_q * B / t / 1e12, 'table_GBs': bytes_t / t / 1e9})
    big, one = out[-1], out[0]
    roof = {'large_batch': {'batch': big['batch'], 'bound': 'fp64 pipes', 'achieved': big['algorithmic_TFLOPs'], 'peak': FP64_MFMA_PEAK_TF,
                            'unit': 'TFLOP/s', 'frac': big['algorithmic_TFLOPs'] / FP64_MFMA_PEAK_TF,
                            'algorithmic_flops_per_query': flops_q},
            'single_query': {'batch': 1, 'bound': 'hbm (streamed tables) -- in practice launch latency', 'achieved': one['table_GBs'],
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': one['table_GBs'] / HBM_PEAK_GBS,
                             'algorithmic_bytes_per_query': bytes_t}}
    return out, roof


def run_analytic(args):
    from sgdml_amd import _lib

    N, M, B = args.n_atoms, args.n_train, args.n_query
    N3 = 3 * N
    n = M * N3
    sig = args.sig
    R, E, F = synth_geometries(N, M + B, seed=0)
    Rf = R.reshape(M + B, -1)
    y = F[:M].ravel().copy()
    y_std = np.std(y)
    y /= y_std

    t_first = time.perf_counter()
    ctx = _lib.Context(0)
    first_call = {'context_s': time.perf_counter() - t_first}
    tp = np.zeros((1, N * (N - 1) // 2), dtype=np.int64)
    tp[0] = np.arange(tp.shape[1])
    xd, gd = ctx.desc_from_R(Rf[:M], N)
    ctx.train_upload(xd, gd, tp)  # resident before timing
    first_call['descriptors_and_upload_s'] = time.perf_counter() - t_first - first_call['context_s']
    # query geometries resident in HBM
    dR, dE, dF = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib = ctx._lib
    ctx._check(lib.gdml_dev_alloc(ctx._h, B * N3 * 8, C.byref(dR)))
    ctx._check(lib.gdml_dev_alloc(ctx._h, B * 8, C.byref(dE)))
    ctx._check(lib.gdml_dev_alloc(ctx._h, B * N3 * 8, C.byref(dF)))
    Rq = np.ascontiguousarray(Rf[M:])
    ctx._check(lib.gdml_memcpy_h2d(ctx._h, dR, Rq.ctypes.data_as(C.c_void_p), Rq.nbytes))

    phases = {'assemble': [], 'factor': [], 'solve': [], 'predict': []}
    info_last = [0]

    def step(record):
        # the calls of sgdml_amd.solvers.analytic.Analytic.solve
        if args.separate_solve:
            ctx.assemble_K(sig, False, for_cholesky=args.lam)
            info_last[0] = ctx.chol_factor(args.lam)
            alphas = ctx.chol_solve(y)
        else:
            ctx.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=args.lam)
            ctx.chol_set_rhs(y)
            info_last[0] = ctx.chol_factor(args.lam)
            alphas = ctx.chol_solve(None)
        # model for prediction: J alpha on the device (training-set Jacobians are resident)
        if step.first:
            ctx.predict_upload_model(xd, np.zeros_like(xd), tp, sig, None)
            step.first = False
        ctx.set_alphas(alphas)
        ctx._check(lib.gdml_predict_dev(ctx._h, dR, B, None, None, dE, dF))
        if record:
            for k in phases:
                phases[k].append(ctx.phase_ms(k)[0])
        return alphas

    step.first = True
    # what the FIRST build+solve of a process costs (`first_call_in_process`, beside the steady-state `value`): code-object
    # loading at first launch, the 31.8 GB hipMalloc of the matrix, the model upload
    t_fs = time.perf_counter()
    n_warm = args.warmup
    if n_warm > 0:
        step(False)
        ctx.sync()
        n_warm -= 1
        first_call['first_step_s'] = time.perf_counter() - t_fs
        first_call['first_build_solve_s'] = sum(ctx.phase_ms(k_)[0] for k_ in ('assemble', 'factor', 'solve')) / 1e3
        first_call['process_start_to_first_solution_s'] = time.perf_counter() - t_first
    for _ in range(n_warm):
        step(False)
    if not args.no_profile:
        ctx.profile(True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        alphas = step(True)
    ctx.sync()
    t1 = time.perf_counter()
    wall_ms = (t1 - t0) * 1e3 / max(1, args.steps)
    build_solve_ms = float(np.mean(phases['assemble']) + np.mean(phases['factor']) + np.mean(phases['solve']))
    pred_ms = float(np.mean(phases['predict']))

    # sanity of the result that was timed: residual of the solve through the matrix-free operator
    Kv = ctx.kernel_matvec(args.lam, False, -alphas)
    resid = float(np.linalg.norm(-Kv - y) / np.linalg.norm(y))  # (-K + lam I) x = -(Kx - lam x)
    if not resid < 1e-10:  # a fast wrong answer is not a result (the analytic contract)
        raise SystemExit('bench: residual of the timed solve is %.3e (> 1e-10): refusing to report' % resid)
    roof = None
    extra = {}
    traffic = load_pmc_traffic()
    if not args.no_profile:
        # dominant kernel: gemm_nt_sub_diag_kernel, the trailing-update launches of the blocked Cholesky (each also carries
        # the workgroup that factors the next diagonal block); the plain gemm_nt_sub_kernel launches (first / last panels,
        # K = 64 steps of the tail) are reported next to it
        g_ms, g_n, g_fl = ctx.kernel_stat('gemm_nt_sub_diag')
        g2_ms, g2_n, g2_fl = ctx.kernel_stat('gemm_nt_sub')
        gemm_kernel = 'gemm_nt_sub_diag_kernel (fp64 MFMA trailing update of the blocked Cholesky, K = 1024 / 512)'
        if g_ms <= 0:
            g_ms, g_n, g_fl, g2_ms = g2_ms, g2_n, g2_fl, 0.0
            gemm_kernel = 'gemm_nt_sub_kernel (fp64 MFMA SYRK/GEMM of the blocked Cholesky)'
        a_ms, a_n, a_by = ctx.kernel_stat('assemble')
        p_ms, p_n, p_fl = ctx.kernel_stat('predict')
        if g_ms > 0:
            ach = g_fl / (g_ms * 1e-3) / 1e12
            tr = traffic.get('gemm_nt_sub_diag', traffic.get('gemm_nt_sub', {}))
            roof = {'kernel': gemm_kernel,
                    'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                    'frac': ach / FP64_MFMA_PEAK_TF, 'traffic': tr.get('hbm_bytes_per_launch'),
                    'traffic_source': tr.get('source'),
                    'launches': g_n, 'avg_launch_ms': g_ms / max(1, g_n),
                    'algorithmic_flops_per_launch': g_fl / max(1, g_n)}
            if g2_ms > 0:
                roof['other_gemm_launches'] = {'kernel': 'gemm_nt_sub_kernel', 'launches': g2_n, 'total_ms': g2_ms,
                                               'achieved': g2_fl / (g2_ms * 1e-3) / 1e12}
        if a_ms > 0:
            ach = a_by / (a_ms * 1e-3) / 1e9
            tr = traffic.get('assemble', {})
            extra['roofline_assemble'] = {'kernel': 'assemble_strip_kernel (-K + lam I, blocks on/below the diagonal, 64-column strips)',
                                          'bound': 'hbm', 'achieved': ach,
                                          'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                                          'traffic': tr.get('hbm_bytes_per_launch'), 'traffic_source': tr.get('source'),
                                          'algorithmic_bytes_per_launch': a_by / max(1, a_n),
                                          'avg_launch_ms': a_ms / max(1, a_n)}
        if p_ms > 0:
            extra['predict_kernel'] = {'kernel': 'predict_kernel', 'avg_launch_ms': p_ms / max(1, p_n),
                                       'algorithmic_TFLOPs': p_fl / (p_ms * 1e-3) / 1e12}
    chol_tf = (n**3 / 3.0) / (np.mean(phases['factor']) * 1e-3) / 1e12
    sk = line_skeleton('analytic', N, M, 1)
    out = {
        'metric': sk['metric'],
        'value': build_solve_ms / 1e3,
        'unit': 's',
        'n_gpus': 1,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': wall_ms,
        'higher_is_better': False,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': sk['workload'], 'n_atoms': N, 'n_train': M, 'n_perms': 1, 'sig': sig, 'lam': args.lam,
                   'matrix_n': n, 'query_batch': B, 'parallelism': sk['parallelism']},
        'phases_ms': {k: float(np.mean(v)) for k, v in phases.items()},
        'cholesky_info': info_last[0],
        'solve_rel_residual': resid,
        'cholesky_TFLOPs_whole_factorization': chol_tf,
        'predict': {'geoms_per_s': B / (pred_ms * 1e-3), 'forces_per_s': B * N3 / (pred_ms * 1e-3),
                    'batch': B, 'ms': pred_ms},
        'roofline': roof,
    }
    out.update(extra)
    out['first_call_in_process'] = dict(first_call, note='context creation, descriptors + upload, then the first step (warm-up, untimed): '
                                        'first_build_solve_s are its device phases (first_step_s - that = hipMalloc of the matrix, code-object '
                                        'loading, model upload); `value` is the steady state')
    # the point this line contributes to a `--gpus 1, 2, 4, 8` curve (same keys in the N > 1 line)
    out['scale_point'] = {'workload': sk['workload'], 'seconds': out['value'], 'n_gpus': 1,
                          'roofline_frac': None if roof is None else roof['frac']}
    cpu_handle = None
    try:  # the metric names "predict forces/sec": the resident model at four batch sizes, with both rooflines
        ctx.profile(False)
        out['predict']['by_batch'], out['roofline_predict'] = predict_sweep(ctx, lib, Rq, N, M, 1)
    except Exception as e:
        out['predict']['by_batch'] = {'error': repr(e)}
    try:  # what precedes the solve in the user's flow: GDMLTrain.create_task's symmetry search over (at most) 1000 geometries
        from sgdml_amd.utils import perm as perm_mod

        Ms = min(M, 1000)
        zs = np.array([6] * (N // 3) + [1] * (N - N // 3))
        walls = []
        for _ in range(2):
            t0 = time.perf_counter()
            grp = perm_mod.find_perms(R[:Ms].reshape(Ms, N, 3), zs, ctx=ctx)
            walls.append(time.perf_counter() - t0)
        out['symmetry_search'] = {
            'geometries': Ms, 'pairs': Ms * (Ms - 1) // 2, 'wall_s': walls[-1], 'first_call_s': walls[0],
            'device_ms': ctx.phase_ms('perm_match')[0], 'group_order': int(grp.shape[0]),
            'note': 'sgdml_amd.utils.perm.find_perms with eigenvectors (batched Jacobi) and pairwise matching on the device '
                    '(gdml_perm_match: one assignment problem per wavefront); the same search costs 18.1 s in NumPy/SciPy on the host and 22.6 / 5.0 s in the '
                    'reference with 1 / 8 processes (profiles/r06_perm_match.txt) -- not part of `value`'}
    except Exception as e:
        out['symmetry_search'] = {'error': repr(e)}
    ctx.close()
    if not args.no_configs:
        cfgs = []
        try:
            cfgs.append(sigma_sweep_config0())
        except Exception as e:  # the headline must not die with an extra
            cfgs.append({'config': 'configs[0] sweep', 'error': repr(e)})
        # One device arena for everything that follows (GDMLTrain.reserve_device_memory): the 127 / 180 GB matrices of the
        # large shapes otherwise spend 3.5-6.5 s each in hipMalloc (profiles/r04_malloc_probe.txt); a long-lived process
        # pays that once.  Reported, not hidden: `arena`.
        arena = None
        try:
            c0 = _lib.Context(0)
            t0 = time.perf_counter()
            got = c0.mem_reserve()
            arena = {'reserved_GB': got / 2**30, 'reserve_s': time.perf_counter() - t0,
                     'note': 'one hipMalloc per process (gdml_mem_reserve); every large buffer below (kernel matrices, the fp32 '
                             'copy of the preconditioner factor) is carved from it first-fit'}
            c0.close()
        except Exception as e:
            arena = {'error': repr(e)}
        out['arena'] = arena
        # the other BASELINE configuration shapes, each run to a SOLUTION through GDMLTrain.train on this one GPU.  The
        # iterative entries are the STATED sizes of configs[2], [3], [4] through the solver the reference would pick for them
        # (sgdml/train.py:949-964: a matrix that does not fit -> Iterative), on the synthetic-trajectory workload; sigma is
        # scaled with the molecule (the descriptor has N (N - 1) / 2 entries; the reference's CLI searches 10:10:100):
        # 20 for 21 atoms, 60 for 42, 100 for 100 -- at sigma = 20 the 42-atom system needs 4306 iterations instead of 331
        # (profiles/r04_cfg3_m2000_first_probe.txt, r04_large_molecule_cg_probe.txt).
        from sgdml_amd.solvers.iterative import Iterative

        TRAJ = {'n_modes': 8, 'amp': 0.15, 'noise': 0.01}
        for label, kw in (
            ('configs[2]: aspirin-sized N=21, N_train={} iterative solver to solver_tol 1e-4 on a synthetic trajectory '
             '(bench.synth_trajectory), device-memory budget 32 GB; k inducing points by the COST rule (the default, '
             'Iterative.cost_n_inducing_pts: minimiser of predicted build + iteration time, never more than memory '
             'allows)'.format(args.cg_n_train),
             dict(n_atoms=N, n_train=args.cg_n_train, solver='cg', max_memory=32, traj=TRAJ, sig=args.sig)),
            ('configs[2] again with the REFERENCE\'s form of the preconditioner (pcg.precon_form = 0: the stored fp64 factor, '
             'iterative.py:120-140) instead of the fp32 factor + Gram correction the library picks at this size',
             dict(n_atoms=N, n_train=args.cg_n_train, solver='cg', max_memory=32, traj=TRAJ, sig=args.sig,
                  options={'pcg.precon_form': 0})),
            ('configs[2] with inducing_pts_policy = "memory": as many inducing points as the 32 GB hold, the REFERENCE\'s rule '
             '(iterative.py:498-503) applied to this backend\'s HBM model -- rounds 3-5\' default',
             dict(n_atoms=N, n_train=args.cg_n_train, solver='cg', max_memory=32, traj=TRAJ, sig=args.sig, policy='memory')),
            ('configs[2] at the k the REFERENCE\'s own memory rule gives for the same max_memory (iterative.py:827-844 counts four '
             'n x m host arrays: k = 53 at 32 GB; this backend\'s HBM model holds one and allows ~2.7 x as many)',
             dict(n_atoms=N, n_train=args.cg_n_train, solver='cg', max_memory=32, traj=TRAJ, sig=args.sig,
                  n_inducing=Iterative.max_n_inducing_pts(args.cg_n_train, N, 32 * 1024**3))),
            ('configs[3]: N=42 with a 27-element permutation group, N_train=2000 (n = 252 000: 508 GB as a matrix), iterative '
             'solver to solver_tol 1e-4, budget 64 GB, k by the cost rule (default)',
             dict(n_atoms=42, n_train=2000, perms_kind='c3x3', solver='cg', max_memory=64, traj=TRAJ, sig=60)),
            ('configs[3] with inducing_pts_policy = "memory" (rounds 3-5\' default)',
             dict(n_atoms=42, n_train=2000, perms_kind='c3x3', solver='cg', max_memory=64, traj=TRAJ, sig=60, policy='memory')),
            ('configs[3] at the k of the reference\'s memory rule for 64 GB (k = 64; measured optimum of the sweep ~70)',
             dict(n_atoms=42, n_train=2000, perms_kind='c3x3', solver='cg', max_memory=64, traj=TRAJ, sig=60,
                  n_inducing=Iterative.max_n_inducing_pts(2000, 42, 64 * 1024**3))),
            ('configs[4]: 100-atom molecule, N_train=3000 (n = 900 000: 6.5 TB as a matrix), iterative solver to solver_tol 1e-4, '
             'budget 64 GB (memory-limited: both rules give the same k)', dict(n_atoms=100, n_train=3000, solver='cg', max_memory=64, traj=TRAJ, sig=100)),
            ('configs[3] shape at the largest N_train one GPU factors directly: N=42, P=27, N_train=1000 (n = 126 000, 127 GB), analytic',
             dict(n_atoms=42, n_train=1000, perms_kind='c3x3', solver='analytic', sig=args.sig)),
            ('configs[4] shape at the largest N_train one GPU factors directly: 100 atoms, N_train=500 (n = 150 000, 180 GB), analytic',
             dict(n_atoms=100, n_train=500, solver='analytic', sig=args.sig)),
        ):
            if kw.get('solver') == 'analytic' and cpu_handle is None and not args.no_cpu:
                # the CPU leg's largest sample starts here: the two analytic shapes are minutes of long GPU kernels with no
                # host role, so the host cores are free -- beside the iterative entries above it cost their per-iteration host
                # callbacks 20 % (measured: configs[2] 7.2 -> 8.6 s), so it does not run there
                cpu_handle = start_cpu_sample(M if args.cpu_full else min(M, args.cpu_sample))
            try:
                cfgs.append(solve_config(label, lam=args.lam, **kw))
            except Exception as e:
                cfgs.append({'config': label, 'error': repr(e)})
        out['configs'] = cfgs
        for c_ in cfgs:  # the configs[2] point of this one-GPU run, under the key the N > 1 lines use
            if isinstance(c_, dict) and str(c_.get('config', '')).startswith('configs[2]: ') and 'train_wall_s' in c_:
                out['scale_point_cg'] = {'workload': 'configs[2] to solver_tol 1e-4 through GDMLTrain.train', 'seconds': None,
                                         'n_gpus': 1, 'time_to_tol_s': c_['train_wall_s']}
                break
    if not args.no_cpu:
        if cpu_handle is None:  # --no-configs: nothing to overlap with
            cpu_handle = start_cpu_sample(M if args.cpu_full else min(M, args.cpu_sample))
        over = finish_cpu_sample(cpu_handle, timeout=900 if args.cpu_full else 240)
        out['cpu_baseline'] = cpu_baseline(N, args.sig, args.lam, M, overlapped=over)
    else:
        out['cpu_baseline'] = None
    return out


if __name__ == '__main__':
    main()
