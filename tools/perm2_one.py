"""One parity case of csrc/assemble_perm2.hip in its own process:  python tools/perm2_one.py N M group post"""
import sys
sys.path.insert(0, 'tools')
from asm_perm_check import check_case
N, M, kind, post = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
check_case(N, M, kind, {'asm.perm2_post': post, 'asm.perm2_min_n': 25, 'asm.perm2_min_p': 2})
