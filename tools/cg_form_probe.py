"""configs[2] / [4] run to solver_tol through GDMLTrain.train with the preconditioner applied from the stored factor and
from the fp32 copy with the Gram correction (pcg.precon_form 0 / 3), at the memory model's k and -- separately: it leaves the reference's memory model -- larger k.

    python tools/cg_form_probe.py [cfg2|cfg4|cfg3] [form ...] [k=...]   > gpurun_out/cg_form_probe.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

TRAJ = {'n_modes': 8, 'amp': 0.15, 'noise': 0.01}
SHAPES = {
    'cfg2': dict(n_atoms=21, n_train=5000, solver='cg', max_memory=32, traj=TRAJ, sig=20),
    'cfg3': dict(n_atoms=42, n_train=2000, perms_kind='c3x3', solver='cg', max_memory=64, traj=TRAJ, sig=60),
    'cfg4': dict(n_atoms=100, n_train=3000, solver='cg', max_memory=64, traj=TRAJ, sig=100),
}


def main():
    from sgdml_amd import _lib

    args = sys.argv[1:]
    shape = args[0] if args else 'cfg2'
    forms = [int(a) for a in args[1:] if a.isdigit()] or [0, 3]
    ks = [int(a[2:]) for a in args[1:] if a.startswith('k=')] or [None]
    c0 = _lib.Context(0)
    c0.mem_reserve()
    c0.close()
    for k in ks:
        for form in forms:
            kw = dict(SHAPES[shape])
            if k is not None:
                kw['n_inducing'] = k
                kw['max_memory'] = None
            r = bench.solve_config('%s form=%d k=%s' % (shape, form, k), options={'pcg.precon_form': form}, **kw)
            keep = {x: r.get(x) for x in ('config', 'time_to_tol_s', 'solver_iters', 'converged', 'restarts', 'inducing_pts_per_stage',
                                          'precon_form', 'f32_gram_min_pivot', 'ms_per_pcg_iteration', 'phases_ms_last', 'resid_over_norm_y')}
            print(json.dumps(keep), flush=True)


if __name__ == '__main__':
    main()
