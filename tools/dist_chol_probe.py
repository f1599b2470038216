"""gdml_dist_chol_solve with ONE rank (no communicator) at the benchmark size: phase times of the block-row-cyclic
factorisation with look-ahead, next to the single-GPU factorisation of the same system.
    python tools/dist_chol_probe.py [M] [nb]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgdml_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = 21
R, E, F = bench.synth_geometries(N, M, seed=0)
y = F.ravel() / np.std(F)
c = _lib.Context(0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
xd, gd = c.desc_from_R(R.reshape(M, -1), N)
c.train_upload(xd, gd, tp)
c.set_option('dist.nb', nb)
c.set_option('dist.force_panels', 1)  # round 6: the panel schedule itself (a one-rank call otherwise takes the single-GPU path)
for rep in range(2):
    c.sync(); t0 = time.perf_counter()
    a = c.dist_chol_solve(20.0, 1e-10, y)
    c.sync(); t1 = time.perf_counter()
    print('dist (1 rank) n=%d nb=%d: wall %.3f s  assemble %.1f ms  factor %.1f ms  solve %.1f ms' % (
        M * 3 * N, nb, t1 - t0, c.phase_ms('assemble')[0], c.phase_ms('factor')[0], c.phase_ms('solve')[0]), flush=True)
c.predict_upload_model(xd, np.zeros_like(xd), tp, 20.0, None)
r = c.kernel_matvec(1e-10, False, -a)
print('residual %.2e' % (np.linalg.norm(-r - y) / np.linalg.norm(y)))
c.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
c.chol_set_rhs(y)
c.chol_factor(1e-10)
print('single-GPU path: factor %.1f ms' % c.phase_ms('factor')[0])
c.close()
