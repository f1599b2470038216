"""Tolerance of the analytic solve (analytic.py:94-99): ||(-K + lam I)(-alpha) - y|| / ||y|| <= 1e-10.

One fixture cannot meet it with ANY backward-stable solver: n4_p6_pbc (lam = 1e-10, ||alpha|| ~ 1e9 ||y||) -- the reference's
own LAPACK cho_solve leaves 2.7e-9 there.  For such systems the bound is one unit roundoff of normwise backward error,
eps ||A||_2 ||x|| / ||y|| (1.9e-8 for that fixture; below 1e-10 for every other fixture, where the contract applies as stated).
"""
import numpy as np


def solve_tol(A, x, y):
    ulp = np.finfo(np.float64).eps * np.linalg.norm(A, 2) * np.linalg.norm(x) / np.linalg.norm(y)
    return max(1e-10, ulp)
