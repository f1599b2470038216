"""Round 6: assemble_big1_kernel (P = 1, N > 21) against the general kernel: parity with the oracle at a small size, then time and
HBM fraction of the lower form (A = -K + lam I, what gdml_assemble_A writes) and of a dense column range.
    python tools/big1_probe.py [N ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import synth_geometries  # noqa: E402
from oracle import gdml_oracle as orc  # noqa: E402
from sgdml_amd import _lib  # noqa: E402

for N in ([int(a) for a in sys.argv[1:]] or [100, 60, 30]):
    # parity at a small size
    M = 5
    ds = orc.synth_dataset(N, M, seed=3, jitter=0.3)
    xd, gd = orc.desc_from_R(ds['R'].reshape(M, -1))
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    Ko = orc.assemble_K(xd, gd, orc.tril_perms_lin_from_tril_perms(tp), 30.0)
    c = _lib.Context(0)
    c.train_upload(xd, gd, tp)
    for opt in (1, 0):
        c.set_option('asm.big1', opt)
        K = c.assemble_K(30.0, False, to_host=True)
        Ks = c.assemble_K(30.0, False, points=(1, 4), to_host=True)
        c.assemble_K(30.0, False, alloc_extra_rows=1, for_cholesky=1e-6)
        A = c.K_to_host()[:3 * N * M]
        low = np.kron(np.tril(np.ones((M, M))), np.ones((3 * N, 3 * N))).astype(bool)
        Ao = -Ko + 1e-6 * np.eye(3 * N * M)
        print('N=%d asm.big1=%d parity: full %.1e  points(1,4) %.1e  lower A %.1e   (of max|K|)' % (
            N, opt, np.abs(K - Ko).max() / np.abs(Ko).max(), np.abs(Ks - Ko[:, 3 * N:12 * N]).max() / np.abs(Ko).max(),
            np.abs((A - Ao)[low]).max() / np.abs(Ko).max()), flush=True)
    c.close()
    # timing: lower form at a size of ~60 GB
    M = int(np.sqrt(2 * 60e9 / 8) / (3 * N))
    R, E, F = synth_geometries(N, M, seed=0)
    c = _lib.Context(0)
    xd, gd = c.desc_from_R(R.reshape(M, -1), N)
    c.train_upload(xd, gd, tp)
    n = 3 * N * M
    for opt in (0, 1, 2, 3, 4):  # 2: no phase 1, 3: no phase 2, 4: phase 2 without its gathers (timing only)
        c.set_option('asm.big1', opt)
        ts = []
        for rep in range(3):
            c.assemble_K(30.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
            ts.append(c.phase_ms('assemble')[0])
        by = 8.0 * (3 * N) ** 2 * M * (M + 1) / 2
        t = min(ts)
        print('N=%d M=%d n=%d lower form asm.big1=%d: %s ms -> %.2f TB/s = %.2f of HBM' % (N, M, n, opt, ' '.join('%.1f' % x for x in ts), by / t / 1e9, by / t / 1e9 / 8.0), flush=True)
    c.close()
