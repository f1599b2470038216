"""Round 4 (second session): molecules beyond 128 atoms and the LDS-row variant of assemble_perm_kernel.
  python tools/large_n_probe.py
Times (a) the general assembly kernel at N = 100 with the permutation rows held in lanes (two v_readlane + select) against the
LDS-row variant that serves N > 128 (asm.perm_lds_rows = 1), (b) assembly and prediction at N = 150 / 200 (new sizes)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_perm_check import time_case  # noqa: E402
from bench import synth_geometries  # noqa: E402
from oracle import gdml_oracle as orc  # noqa: E402  (table builders only)
from sgdml_amd import _lib  # noqa: E402

for rep in range(2):
    for N, M, kind in [(100, 120, 'id'), (70, 170, 'c2xc2'), (128, 90, 'id')]:
        time_case(N, M, kind, {'asm.perm_lds_rows': 0}, label='lanes')
        time_case(N, M, kind, {}, label='ldsrow')
for N, M, kind in [(150, 80, 'id'), (150, 80, 'c2xc2'), (200, 60, 'id')]:
    time_case(N, M, kind, {}, label='big')
    time_case(N, M, kind, {}, lower=True, label='big')

# prediction beyond D = 8192: every batch through the GEMM pipeline
for N, M, B in [(150, 200, 1), (150, 200, 16), (150, 200, 512), (200, 100, 256)]:
    R, E, F = synth_geometries(N, M + B, seed=1)
    tp = orc.tril_perms_from_atom_perms(np.arange(N)[None])
    c = _lib.Context(0)
    xd, gd = c.desc_from_R(R[:M].reshape(M, -1), N)
    ja = np.random.RandomState(0).normal(size=xd.shape)
    c.predict_upload_model(xd, ja, tp, 60.0, None)
    q = R[M:].reshape(B, -1)
    c.predict(q)
    t0 = time.time()
    for _ in range(3):
        c.predict(q)
    dt = (time.time() - t0) / 3
    print('predict N=%d D=%d M=%d B=%d: %.2f ms per call (host to host), kernels %.2f ms' % (
        N, N * (N - 1) // 2, M, B, dt * 1e3, c.phase_ms('predict')[0] if hasattr(c, 'phase_ms') else float('nan')), flush=True)
    c.close()
