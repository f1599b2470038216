"""K_nm assembly for index-list columns at the configs[3] / [4] shapes (the two assemblies of an iterative run: the
leverage-score sample, 3N * 10 random columns, and the k * 3N inducing columns), compact column-atom strips on / off:
    python tools/asm_list_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgdml_amd import _lib
from sgdml_amd.utils.desc import Desc

for N, M, kind, k in [(42, 2000, 'c3x3', 182), (100, 3000, None, 25), (21, 2000, 'c2x2', 100)]:
    R, E, F = bench.synth_trajectory(N, M, seed=3)
    perms = bench.perm_group(N, kind)
    tril = np.array([Desc.perm(p) for p in perms])
    n = 3 * N * M
    rng = np.random.default_rng(1)
    for label, m in (('leverage sample', 3 * N * 10), ('inducing columns', 3 * N * k)):
        idx = np.sort(rng.choice(n, size=m, replace=False))
        for compact in (0, 1):
            c = _lib.Context(0)
            c.set_option('asm.perm_compact', compact)
            xd, gd = c.desc_from_R(R.reshape(M, -1), N)
            c.train_upload(xd, gd, tril)
            ts = []
            for rep in range(2):
                c.assemble_K(20.0, False, idx=idx, alloc_extra_rows=m)
                ts.append(c.phase_ms('assemble')[0])
            print('N=%-3d M=%-4d P=%-2d %-17s m=%-6d compact=%d  %8.1f ms  (%.1f GB)' % (N, M, len(perms), label, m, compact, min(ts), n * m * 8 / 1e9), flush=True)
            c.close()
