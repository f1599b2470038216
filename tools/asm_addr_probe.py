"""Does the assembly time depend on where the 31.75 GB matrix lands in the address space?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

N, M = 21, 1000
R, E, F = synth_geometries(N, M, seed=0)
Rf = R.reshape(M, -1)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
for extra, pre in ((0, 0), (1, 0), (2, 0), (64, 0), (0, 1 << 20), (0, 3 << 20), (1, 5 << 20), (0, 1 << 30), (1, 0), (0, 0)):
    ctx = _lib.Context(0)
    if pre:
        d = C.c_void_p()
        ctx._check(ctx._lib.gdml_dev_alloc(ctx._h, pre, C.byref(d)))
    xd, gd = ctx.desc_from_R(Rf, N)
    ctx.train_upload(xd, gd, tp)
    ts = []
    for rep in range(3):
        ctx.assemble_K(20.0, False, alloc_extra_rows=extra)
        ts.append(ctx.phase_ms('assemble')[0])
    p, ld = C.c_void_p(), C.c_int64()
    ctx._check(ctx._lib.gdml_K_dev(ctx._h, C.byref(p), C.byref(ld)))
    print('extra_rows=%d pre_alloc=%d: K at 0x%x (mod 2MB = 0x%x, mod 1GB = 0x%x) ld=%d: assemble %s ms' % (
        extra, pre, p.value, p.value % (2 << 20), p.value % (1 << 30), ld.value, ' '.join('%.2f' % t for t in ts)), flush=True)
    ctx.close()
