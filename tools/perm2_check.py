"""assemble_perm2.hip on the GPU: parity against the oracle (all column modes of tools/asm_perm_check.py; the dense ones run on
the new kernel, the others on assemble_perm_kernel) and the A/B against assemble_perm_kernel at configs[3]'s shape.
    python tools/perm2_check.py check
    python tools/perm2_check.py time [full]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asm_perm_check import check_case as _cc, time_case as _tc  # noqa: E402

FORCE = {'asm.perm2_min_n': 25, 'asm.perm2_min_p': 2}   # by default the kernel only takes N >= 40, P >= 16


def check_case(N, M, kind, opts):
    return _cc(N, M, kind, dict(FORCE, **opts))


def time_case(N, M, kind, opts, **kw):
    return _tc(N, M, kind, dict(FORCE, **opts), **kw)


def do_check():
    ok = True
    for N, M, kind in [(26, 4, 'c3xc2'), (30, 5, 'c3xc2'), (36, 3, 'c3^3'), (42, 4, 'c3^3'), (33, 3, 'c2xc2'), (25, 3, 'c3^3'),
                       (42, 11, 'c3xc2')]:
        ok &= check_case(N, M, kind, {})
    for N, M, kind in [(36, 3, 'c3^3'), (42, 3, 'c3^3')]:
        for opts in [{'asm.perm2_split': 0}, {'asm.perm2_post': 0}, {'asm.perm2_ed': 0}, {'asm.perm2_es': 0}, {'asm.perm2_direct': 0}, {'asm.perm2_chunk': 5}, {'asm.perm2_chunk': 100}, {'asm.perm2_i_chunk': 2}, {'asm.perm2': 0}]:
            ok &= check_case(N, M, kind, opts)
    print('ALL OK' if ok else 'SOME FAILED')
    return ok


def do_time(full):
    for lower in (False, True):
        for opts in [{'asm.perm2': 0}, {}, {'asm.perm2_direct': 0}, {'asm.perm2_ed': 0}, {'asm.perm2_es': 0}, {'asm.perm2_post': 0}, {'asm.perm2_split': 0}]:
            time_case(42, 300, 'c3^3', opts, lower=lower, label='perm2')
    for opts in [{'asm.perm2_debug': 1}, {'asm.perm2_debug': 2}, {'asm.perm2_debug': 4}, {'asm.perm2_debug': 8}, {'asm.perm2_debug': 15},
                 {'asm.perm2_debug': 15 + 64}, {'asm.perm2_debug': 511}]:
        time_case(42, 300, 'c3^3', opts, label='perm2')
    for N, M, kind in [(42, 300, 'c3xc2'), (30, 400, 'c3xc2'), (36, 350, 'c3^3')]:
        time_case(N, M, kind, {'asm.perm2': 0}, label='perm2')
        time_case(N, M, kind, {}, label='perm2')
    if full:
        for opts in [{'asm.perm2': 0}, {'asm.perm2_direct': 0}, {}]:
            time_case(42, 1000, 'c3^3', opts, lower=True, reps=3, label='perm2')


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'check'
    if mode == 'check':
        sys.exit(0 if do_check() else 1)
    if mode == 'trace':   # shader-clock stamps (100 MHz) of one workgroup's phases, printed by the library on stderr
        time_case(42, 300, 'c3^3', {'asm.perm2_debug': 1024}, reps=1, label='trace')
        sys.exit(0)
    do_time(len(sys.argv) > 2 and sys.argv[2] == 'full')
