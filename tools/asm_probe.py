"""Assembly-kernel probe: time gdml_assemble_K alone for a few shapes / ablation flags."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib

def run(N, M, P=1, reps=3):
    R, E, F = synth_geometries(N, M, seed=0)
    ctx = _lib.Context(0)
    D = N * (N - 1) // 2
    from oracle import gdml_oracle as orc
    perms = [list(range(N))]
    for q in range(1, P):  # P - 1 transpositions of disjoint atom pairs (not a group: the kernel does not care)
        p2 = list(range(N)); p2[2*q-2], p2[2*q-1] = p2[2*q-1], p2[2*q-2]; perms.append(p2)
    tp = orc.tril_perms_from_atom_perms(np.array(perms))
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tp)
    ts = []
    for _ in range(reps):
        ctx.assemble_K(20.0, False)
        ts.append(ctx.phase_ms('assemble')[0])
    n = M * 3 * N
    print('N=%d M=%d P=%d dbg=%s: %.2f ms  -> %.0f GB/s' % (N, M, P, os.environ.get('GDML_OPTIONS', '-'), min(ts), 8.0 * n * n / min(ts) / 1e6), flush=True)
    ctx.close()

if __name__ == '__main__':
    N, M, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1
    run(N, M, P)
