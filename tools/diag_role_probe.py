"""Duration of the single-workgroup diagonal-block factorisation (diag_block_role) in isolation: a 1088 x 1088 SPD matrix
factored with the fused schedule forced (chol.fused_min_rows = 0): the fused launch of the second panel carries a 64-row
SYRK only, so its duration is the block factorisation.  Run under rocprofv3 --kernel-trace (tools/kstat_opts.sh style)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from sgdml_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))

ctx = _lib.Context(0)
ctx.set_option('chol.fused_min_rows', 0)
ctx.set_option('chol.outer', 512)
M = 182  # 2 atoms -> 6 columns per point: n = 1092
rs = np.random.RandomState(0)
R = rs.normal(size=(M, 6)) * 2
xd, gd = ctx.desc_from_R(R, 2)
ctx.train_upload(xd, gd, np.zeros((1, 1), dtype=np.int64))
ctx.assemble_K(10.0, False, alloc_extra_rows=1)
n = ctx.K_shape()[0]
B = rs.normal(size=(n, n + 30))
A = B @ B.T / n + 0.5 * np.eye(n)
p, ld = C.c_void_p(), C.c_int64()
ctx._check(ctx._lib.gdml_K_dev(ctx._h, C.byref(p), C.byref(ld)))
buf = np.zeros((n + 1, ld.value))
buf[:n, :n] = -A
for rep in range(3):
    ctx._check(ctx._lib.gdml_memcpy_h2d(ctx._h, p, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    ctx.assemble_K  # (matrix state flags stay valid: plain K semantics)
    ctx._check(ctx._lib.gdml_K_dev(ctx._h, C.byref(p), C.byref(ld)))
    try:
        info = ctx.chol_factor(0.0)
    except Exception as e:
        print('factor failed', e)
        break
    print('n', n, 'info', info, 'factor ms', ctx.phase_ms('factor')[0])
