"""Assembly time as a function of what the GPU did just before (clock / power state?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

N, M = 21, 1000
R, E, F = synth_geometries(N, M, seed=0)
Rf = R.reshape(M, -1)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
y = F.ravel() / np.std(F)
ctx = _lib.Context(0)
xd, gd = ctx.desc_from_R(Rf, N)
ctx.train_upload(xd, gd, tp)
def asm(tag):
    ctx.assemble_K(20.0, False, alloc_extra_rows=1)
    print('%-44s assemble %.2f ms' % (tag, ctx.phase_ms('assemble')[0]), flush=True)
asm('cold (first launch)')
asm('right after an assembly')
asm('right after an assembly')
for rep in range(2):
    ctx.chol_set_rhs(y); ctx.chol_factor(1e-10); ctx.chol_solve(None)
    asm('right after factor + solve')
    asm('right after that assembly')
    time.sleep(0.5)
    asm('after 0.5 s idle')
    ctx.chol_set_rhs(y); ctx.chol_factor(1e-10)
    asm('right after factor (no solve)')
    time.sleep(0.05)
    asm('after 50 ms idle')
