#!/usr/bin/env python
"""Where the wall-clock of the configs[0]-shaped sigma sweep goes: the sweep three times in one process (first-touch effects:
code-object load, first allocations), then a cProfile of a fourth.   python tools/sweep_hostprof.py"""
import cProfile
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for rep in range(3):
    t0 = time.perf_counter()
    r = bench.sigma_sweep_config0()
    print('sweep %d: %s  (outer wall %.3f s)' % (rep, json.dumps({k: round(r[k], 4) for k in ('wall_s', 'create_task_s', 'train_s', 'validate_s', 'test_s')}),
                                              time.perf_counter() - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
bench.sigma_sweep_config0()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
