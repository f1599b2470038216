"""One assembly shape, a few repetitions (profiling target):  python tools/asm_perm_one.py N M kind [lower] [key=val ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_perm_check import group_perms
from oracle import gdml_oracle as orc
from bench import synth_geometries
from sgdml_amd import _lib

N, M, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lower = 'lower' in sys.argv[4:]
R, E, F = synth_geometries(N, M, seed=0)
tp = orc.tril_perms_from_atom_perms(group_perms(N, kind))
c = _lib.Context(0)
for kv in sys.argv[4:]:
    if '=' in kv:
        k, v = kv.split('='); c.set_option(k, float(v))
xd, gd = c.desc_from_R(R.reshape(M, -1), N)
c.train_upload(xd, gd, tp)
for _ in range(3):
    if lower: c.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
    else: c.assemble_K(20.0, False)
    print('assemble %.3f ms' % c.phase_ms('assemble')[0], flush=True)
c.close()
