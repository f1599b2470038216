"""Comparison of two PCG residual histories (test helper shared by the CPU and GPU suites)."""
import numpy as np


def crossings(hist, norm_y, levels=(3e-1, 1e-1, 3e-2, 1e-2, 3e-3)):
    """Iteration at which a residual history first drops below level * ||y||: what can be compared between two correct
    PCG runs on a system with cond ~ 1/lam (pointwise the histories decorrelate after a few dozen steps -- a different
    summation order in one dot product is enough -- and plateaus make late crossings arbitrary)."""
    hist = np.asarray(hist)
    out = []
    for lv in levels:
        hit = np.nonzero(hist < lv * norm_y)[0]
        out.append(int(hit[0]) if len(hit) else len(hist))
    return np.array(out)


def assert_same_convergence(ours, ref, norm_y):
    a, b = crossings(ours, norm_y), crossings(ref, norm_y)
    assert np.all(np.abs(a - b) <= np.maximum(3, 0.15 * b)), (a, b)
