"""assemble_pts.hip on the GPU: parity against the oracle (all column modes go through check_case of asm_perm_check: the
dense ones take the new kernel), then timings next to assemble_perm.hip (asm.pts = 0).
    python tools/asm_pts_check.py check | time [key=val ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_perm_check import check_case, time_case

mode = sys.argv[1] if len(sys.argv) > 1 else 'check'
opts = {}
for kv in sys.argv[2:]:
    k, v = kv.split('='); opts[k] = float(v)
if mode == 'check':
    ok = True
    for N, M, kind in [(9, 9, 'c3xc2'), (12, 8, 'c3^3'), (21, 7, 'c2xc2'), (21, 5, 'id'), (24, 4, 'c3xc2'), (8, 11, 'c2xc2'),
                       (16, 9, 'c3xc2'), (21, 3, 'c2xc2'), (21, 1, 'c2xc2'), (10, 20, 'c3^3')]:
        ok &= check_case(N, M, kind, dict(opts))
    for o in [{'asm.pts_i_chunk': 2}, {'asm.pts_nv': 2}, {'asm.pts_nv': 3}, {'asm.pts_nt': 0}]:
        ok &= check_case(21, 7, 'c2xc2', dict(opts, **o))
        ok &= check_case(9, 9, 'c3xc2', dict(opts, **o))
    print('ALL OK' if ok else 'SOME FAILED')
    sys.exit(0 if ok else 1)
for N, M, kind in [(21, 1000, 'c2xc2'), (12, 1500, 'c3xc2'), (9, 2000, 'c3xc2'), (21, 1000, 'id'), (24, 800, 'c3^3')]:
    time_case(N, M, kind, dict(opts, **{'asm.pts': 0}), label='perm')
    time_case(N, M, kind, dict(opts, **{'asm.wave': 0, 'asm.strip': 0, 'asm.pts': 2}), label='pts')
time_case(21, 1000, 'c2xc2', dict(opts), lower=True, label='pts')
for d in (1, 2, 4, 3, 6, 7):
    time_case(21, 1000, 'c2xc2', dict(opts, **{'asm.pts_debug': d}), label='pts')
