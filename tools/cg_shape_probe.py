"""One fixed-iteration PCG step at an arbitrary shape for a kernel trace (where the iteration time of the large-molecule
configurations goes):  rocprofv3 --kernel-trace -d DIR -- python tools/cg_shape_probe.py <N> <M> <perms_kind|-> <k> <sig> [iters]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgdml_amd import _lib
from sgdml_amd.utils.desc import Desc

N, M = int(sys.argv[1]), int(sys.argv[2])
kind = None if sys.argv[3] == '-' else sys.argv[3]
k, sig = int(sys.argv[4]), float(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
R, E, F = bench.synth_trajectory(N, M, seed=3)
perms = bench.perm_group(N, kind)
tril = np.array([Desc.perm(p) for p in perms])
y = F.ravel() / np.std(F)
ctx = _lib.Context(0)
xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
ctx.train_upload(xd, gd, tril)
ctx.predict_upload_model(xd, np.zeros_like(xd), tril, sig, None)
rng = np.random.default_rng(1)
idx = np.sort(rng.choice(3 * N * M, size=3 * N * k, replace=False))
for rep in range(2):
    ctx.assemble_K(sig, False, idx=idx, alloc_extra_rows=len(idx))
    ctx.nystroem_factor(1e-10, idx)
    x, info, it, resid = ctx.pcg(1e-10, False, y, rtol=0.0, maxiter=iters)
    print(rep, {p: round(ctx.phase_ms(p)[0], 1) for p in ('assemble', 'precon', 'pcg')}, 'ms per iteration %.2f' % (ctx.phase_ms('pcg')[0] / iters), flush=True)
