#!/usr/bin/env python
"""CPU baseline measured ONCE at the benchmark size (BASELINE.json configs[1]: N = 21, N_train = 1000, n = 63 000): the
oracle (NumPy restatement of train.py:97-302 + scipy cho_factor / cho_solve, BLAS/LAPACK pool on all host cores) builds
the 31.75 GB matrix and solves it.  bench.py keeps timing a bounded sample (M = 100 / 300) and reports its
extrapolation error against the record this script writes:
    python tools/cpu_baseline_full.py [M] > profiles/r03_cpu_baseline_full.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import gdml_oracle as orc  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N, sig, lam = 21, 20, 1e-10
R, E, F = bench.synth_geometries(N, M + 64, seed=0)
Rf = R.reshape(len(R), -1)
tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
lin = orc.tril_perms_lin_from_tril_perms(tp)
xo, go = orc.desc_from_R(Rf[:M])
t0 = time.perf_counter()
K = orc.assemble_K(xo, go, lin, sig)
t1 = time.perf_counter()
y = F[:M].ravel() / np.std(F[:M])
alphas, used_lu = orc.analytic_solve(K, y, lam)
t2 = time.perf_counter()
del K
out = {'M': M, 'n': M * 3 * N, 'n_atoms': N, 'assemble_s': t1 - t0, 'solve_s': t2 - t1, 'build_solve_s': t2 - t0,
       'lu_fallback': bool(used_lu), 'cores': os.cpu_count(), 'kind': 'port (oracle/gdml_oracle.py)',
       'cpu_model': next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?')}
print(json.dumps(out))
