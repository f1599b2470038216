"""The matrix-free K v of the iterative solver (gdml_kernel_matvec: set_alphas + training-set prediction) at the three
iterative configurations: wall time per application and algorithmic rate (10 D flops per (query, table row)).
  python tools/matvec_probe.py [cfg2|cfg3|cfg4 ...]      (under rocprofv3 --kernel-trace for the kernel split)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgdml_amd import _lib
from sgdml_amd.utils.desc import Desc

CFG = {'cfg2': (21, 5000, None, 20.0), 'cfg3': (42, 2000, 'c3x3', 60.0), 'cfg4': (100, 3000, None, 100.0)}
for name in (sys.argv[1:] or ['cfg2', 'cfg3', 'cfg4']):
    N, M, kind, sig = CFG[name]
    R, E, F = bench.synth_trajectory(N, M, seed=3, n_modes=8, amp=0.15, noise=0.01)
    perms = bench.perm_group(N, kind)
    tril = np.array([Desc.perm(p) for p in perms])
    ctx = _lib.Context(0)
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tril)
    ctx.predict_upload_model(xd, np.zeros_like(xd), tril, sig, None)
    v = np.random.RandomState(0).normal(size=M * 3 * N)
    ctx.kernel_matvec(1e-10, False, v)
    ctx.sync()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.kernel_matvec(1e-10, False, v)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    D, P = N * (N - 1) // 2, len(perms)
    flops = 10.0 * D * M * M * P
    print('%s N=%d M=%d P=%d D=%d: %.2f ms per K v (host-to-host, v up and K v down included)  %.1f TFLOP/s algorithmic = %.2f of 78.6' % (
        name, N, M, P, D, dt * 1e3, flops / dt / 1e12, flops / dt / 78.6e12), flush=True)
    ctx.close()
