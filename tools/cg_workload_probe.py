#!/usr/bin/env python
"""Which synthetic configs[2] workload does the reference's iterative policy solve to tol = 1e-4?  Runs
bench.solve_config(solver='cg') over the variants given as JSON objects on the command line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for arg in sys.argv[1:]:
    v = json.loads(arg)
    M = v.pop('M', 2000)
    N = v.pop('N', 21)
    r = bench.solve_config('probe', N, M, None, 'cg', sig=v.pop('sig', 20), max_memory=v.pop('mem', None),
                           n_inducing=v.pop('k', None), traj=v.pop('traj', None))
    keep = {k: r[k] for k in ('train_wall_s', 'phases_ms_last', 'resid_over_norm_y', 'solver_iters', 'converged',
                              'inducing_pts_per_stage', 'ms_per_pcg_iteration')}
    print(arg, '->', json.dumps(keep), flush=True)
