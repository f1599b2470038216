"""One sigma sweep for a kernel trace: cfg0 = bench.sigma_sweep_config0 (N=9, P=6, M=200, 9 sigmas); p27 = N=12 with a 27-element
group, M=300, 6 sigmas."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sgdml_amd.sweep import sigma_sweep
from sgdml_amd.train import GDMLTrain

case = sys.argv[1]
if case == 'cfg0':
    r = bench.sigma_sweep_config0()
    print('SWEEP', json.dumps({k: r[k] for k in ('wall_s', 'train_s', 'validate_s', 'test_s')}))
else:
    N, M = 12, 300
    R, E, F = bench.synth_geometries(N, M + 1500, seed=5)
    ds = {'type': 'd', 'name': np.array('s'), 'theory': np.array('p'), 'z': np.full(N, 6), 'R': R, 'E': E, 'F': F}
    gens = []
    for a in (0, 3, 9):
        g = list(range(N)); g[a], g[a + 1], g[a + 2] = a + 1, a + 2, a; gens.append(tuple(g))
    perms = [tuple(range(N))]; fr = list(perms)
    while fr:
        nx = []
        for a in fr:
            for g in gens:
                c = tuple(a[i] for i in g)
                if c not in perms: perms.append(c); nx.append(c)
        fr = nx
    tr = GDMLTrain()
    np.random.seed(0)
    t0 = time.perf_counter()
    best, table, tm = sigma_sweep(tr, ds, M, 500, 1000, sigs=[10, 20, 30, 40, 50, 60], perms=np.array(perms), early_stop=False)
    print('SWEEP', json.dumps({'wall_s': time.perf_counter() - t0, 'train_s': tm['train_s'], 'validate_s': tm['validate_s'], 'n_perms': len(perms)}))
    tr.__del__()
