"""One configs[2]-shaped step of bench.py's sharded-CG workload on one GPU (K_nm rows + Nystroem preconditioner + PCG
iterations), for a rocprofv3 kernel trace:  rocprofv3 --kernel-trace -d DIR -- python tools/cg_step_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from sgdml_amd import _lib

if os.environ.get('GDML_HIP_LIB'):  # an older build of the library (before/after traces): bind only what it exports
    import ctypes
    _old = ctypes.CDLL(os.environ['GDML_HIP_LIB'])
    for _name in list(_lib.SIGNATURES):
        if not hasattr(_old, _name):
            del _lib.SIGNATURES[_name]

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _lib.Context(0)
wl = bench.make_cg_workload(ctx, 21, 5000, 200, 20.0, 1e-10)
for rep in range(2):
    resid, ph = bench.cg_step(ctx, wl, iters)
    print(rep, ph, flush=True)
