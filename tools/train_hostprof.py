#!/usr/bin/env python
"""cProfile of GDMLTrain.train (host side around the kernels):  python tools/train_hostprof.py <n_atoms> <n_train> [solver]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sgdml_amd.train import GDMLTrain  # noqa: E402

N, M = int(sys.argv[1]), int(sys.argv[2])
solver = sys.argv[3] if len(sys.argv) > 3 else 'analytic'
R, E, F = bench.synth_geometries(N, M, seed=3)
task = {'type': 't', 'code_version': 'x', 'dataset_name': np.array('s'), 'dataset_theory': np.array('p'), 'z': np.full(N, 6),
        'R_train': R, 'F_train': F, 'E_train': E, 'idxs_train': np.arange(M), 'md5_train': 'x', 'idxs_valid': np.arange(0),
        'md5_valid': 'x', 'sig': 20, 'lam': 1e-10, 'use_E': True, 'use_E_cstr': False, 'use_sym': False,
        'perms': np.arange(N)[None]}
tr = GDMLTrain()
tr._force_solver = solver
tr._context().desc_from_R(R[:2].reshape(2, -1), N)  # context + first-kernel warm-up outside the profile
pr = cProfile.Profile()
pr.enable()
model = tr.train(task)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
