"""Wall clock of the whole user flow at the benchmark size -- create_task (sampling, symmetry search) + train (descriptors,
assembly, solve, model assembly, integration constant) -- with the host functions that cost the most (cProfile).
  python tools/train_flow_probe.py [n_atoms] [n_train]"""
import cProfile, os, pstats, sys, time, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd.train import GDMLTrain

N = int(sys.argv[1]) if len(sys.argv) > 1 else 21
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
R, E, F = synth_geometries(N, M + 200, seed=0)
z = np.array([6] * (N // 3) + [1] * (N - N // 3))
ds = {'type': 'd', 'name': np.array('synthetic'), 'theory': np.array('none'), 'z': z, 'R': R.reshape(-1, N, 3),
      'F': F.reshape(-1, N, 3), 'E': E, 'md5': np.array('0' * 32)}
tr = GDMLTrain()
tr._context()  # context creation (first hipMalloc etc.) outside the clocks
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    task = tr.create_task(ds, M, ds, 100, sig=20, lam=1e-10)
    t1 = time.time()
    model = tr.train(task)
    t2 = time.time()
    pr.disable()
    print('rep %d: create_task %.3f s (perms %s), train %.3f s' % (rep, t1 - t0, task['perms'].shape, t2 - t1), flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
    print('\n'.join(l for l in s.getvalue().splitlines()[4:] if l.strip())[:6000], flush=True)
