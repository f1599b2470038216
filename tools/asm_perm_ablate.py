"""Phase ablation of assemble_perm_kernel (option asm.perm_debug: 1 no stores, 2 no V tasks, 4 no O phase, 8 no image
prefetch; timing only, results wrong) at the shapes the review names:  python tools/asm_perm_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_perm_check import time_case

quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
for N, M, kind in [(42, 300, 'c3^3'), (100, 120, 'id'), (42, 300, 'id'), (21, 1000, 'c2xc2'), (60, 200, 'id')]:
    for dbg, label in ([(0, 'all')] if quick else [(0, 'all'), (1, '-stores'), (2, '-V'), (4, '-O'), (6, '-V-O'), (7, 'skel'), (8, '-pref')]):
        time_case(N, M, kind, {'asm.perm_debug': dbg, 'asm.pts': 0}, label=label)
    for extra in [{'asm.perm_pg': 1}, {'asm.perm_pg': 3}, {'asm.perm_pg': 4}, {'asm.perm_level': 1}, {'asm.perm_level': 0}]:
        if kind == 'c3^3' and not quick:
            time_case(N, M, kind, dict(extra, **{'asm.pts': 0}), label='var')
