"""Does the assembly time depend on where the 31.75 GB matrix lands / on its row pitch?
(The row-pitch part needs a build with the GDML_K_LD_PAD experiment knob in gdml_assemble_K; it was removed again
after the probe: every pitch above the minimal multiple of 16 was slower, profiles/r01_assemble_address_probe.txt.)"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def trial():
    from bench import synth_geometries
    from sgdml_amd import _lib
    N, M = 21, 1000
    R, E, F = synth_geometries(N, M, seed=0)
    Rf = R.reshape(M, -1)
    tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
    for rep_ctx in range(3):
        ctx = _lib.Context(0)
        xd, gd = ctx.desc_from_R(Rf, N)
        ctx.train_upload(xd, gd, tp)
        ts = []
        for rep in range(3):
            ctx.assemble_K(20.0, False, alloc_extra_rows=1)
            ts.append(ctx.phase_ms('assemble')[0])
        p, ld = C.c_void_p(), C.c_int64()
        ctx._check(ctx._lib.gdml_K_dev(ctx._h, C.byref(p), C.byref(ld)))
        print('pad=%s ctx %d: K at 0x%x ld=%d: assemble %s ms' % (os.environ.get('GDML_K_LD_PAD', '0'), rep_ctx, p.value, ld.value, ' '.join('%.2f' % t for t in ts)), flush=True)
        ctx.close()

if __name__ == '__main__':
    if len(sys.argv) > 1:
        trial()
    else:
        for pad in (0, 16, 32, 48, 64, 128, 256, 496, 528, 1040, 2064):
            env = dict(os.environ, GDML_K_LD_PAD=str(pad))
            subprocess.run([sys.executable, __file__, 'x'], env=env)
