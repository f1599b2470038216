"""A/B of the LDS-only phase barriers (common.h lds_phase_barrier) in assemble_strip / assemble_pts / assemble_perm: the shipped library
against build/nolds/libgdml_hip.so (the same sources with -DGDML_LDS_BARRIERS=0 = __syncthreads()), same box, alternating.
    python tools/barrier_ab.py            (spawns itself once per library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)

SHAPES = [(21, 1000, 'id', True, {}), (21, 1000, 'id', False, {}), (21, 1000, 'c2xc2', False, {}), (21, 1000, 'c2xc2', True, {}),
          (12, 1500, 'c3xc2', False, {}), (100, 120, 'id', False, {}), (100, 400, 'id', True, {}), (60, 200, 'id', False, {}),
          (42, 300, 'c3^3', False, {'asm.perm2': 0}), (30, 400, 'c3xc2', False, {})]

if __name__ == '__main__':
    if len(sys.argv) > 1:
        from asm_perm_check import time_case
        for N, M, kind, lower, opts in SHAPES:
            time_case(N, M, kind, opts, lower=lower, label=sys.argv[1])
        sys.exit(0)
    for rep in range(2):
        for label, lib in (('sync', os.path.join(ROOT, 'build', 'nolds', 'libgdml_hip.so')), ('lds ', None)):
            env = dict(os.environ)
            if lib:
                env['GDML_HIP_LIB'] = lib
            subprocess.run([sys.executable, os.path.abspath(__file__), label], env=env, check=False)
