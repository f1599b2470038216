"""Comparison of two PCG residual histories (test helper shared by the CPU and GPU suites)."""
import numpy as np


def crossings(hist, norm_y, levels=(3e-1, 1e-1, 3e-2, 1e-2)):
    """Iteration at which a residual history first drops below level * ||y||: what can be compared between two correct
    PCG runs on a system with cond ~ 1/lam (pointwise the histories decorrelate after a few dozen steps -- a different
    summation order in one dot product is enough -- and below 1e-2 these systems sit on plateaus where the first
    crossing of a level moves by hundreds of iterations between two correct runs: the reference passed 3e-3 at step 120
    of its P = 6 run, the oracle at 128, the GPU at 272, all three converging after 523 +- 10 %)."""
    hist = np.asarray(hist)
    out = []
    for lv in levels:
        hit = np.nonzero(hist < lv * norm_y)[0]
        out.append(int(hit[0]) if len(hit) else len(hist))
    return np.array(out)


def assert_same_convergence(ours, ref, norm_y):
    a, b = crossings(ours, norm_y), crossings(ref, norm_y)
    assert np.all(np.abs(a - b) <= np.maximum(3, 0.15 * b)), (a, b)
