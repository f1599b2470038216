#!/usr/bin/env python
"""One BASELINE configuration shape to a solution on the GPU (bench.solve_config), as JSON:
    python tools/cfg_solve_probe.py <solver> <n_atoms> <n_train> [perms_kind] [max_memory_GB] [sig] [traj] [n_inducing]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

solver, N, M = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != '-' else None
mem = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != '-' else None
sig = float(sys.argv[6]) if len(sys.argv) > 6 else 20
traj = {'n_modes': 8, 'amp': 0.15, 'noise': 0.01} if len(sys.argv) > 7 and sys.argv[7] == 'traj' else None
k = int(sys.argv[8]) if len(sys.argv) > 8 else None
print(json.dumps(bench.solve_config('probe', N, M, kind, solver, sig=sig, max_memory=mem, traj=traj, n_inducing=k)))
