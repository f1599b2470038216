import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_perm_check import time_case
for M in (120, 500):
    for opts in [{}, {'asm.perm_i_chunk': 4}, {'asm.perm_i_chunk': 8}, {'asm.perm_i_chunk': 32}, {'asm.perm_i_chunk': 64}, {'asm.perm_fast_store': 0}, {'asm.perm_lds_rows': 0}]:
        time_case(100, M, 'id', opts, label='n100', reps=3)
    time_case(100, M, 'id', {}, lower=True, label='n100', reps=3)
