"""Factorisation time of the benchmark matrix (n = 63 000) under schedule options: which stream carries the
panel chain (hardware queue / priority), trailing SYRK split into chunks over two bulk streams.
    python tools/chol_sched_probe.py [M]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import synth_geometries
from sgdml_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = 21
R, E, F = synth_geometries(N, M, seed=0)
ctx = _lib.Context(0)
tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
ctx.train_upload(xd, gd, tp)
y = F.ravel() / np.std(F)

def run(tag, **opts):
    for k, v in opts.items():
        ctx.set_option(k.replace('__', '.'), v)
    ts = []
    for rep in range(2):
        ctx.assemble_K(20.0, False, alloc_extra_rows=1)
        ctx.chol_set_rhs(y)
        info = ctx.chol_factor(1e-10)
        ts.append(ctx.phase_ms('factor')[0])
    print('%-40s factor %s ms (info %d)' % (tag, ' '.join('%.1f' % t for t in ts), info), flush=True)

run('default')
for i in range(8):
    run('panel_stream=%d' % i, chol__panel_stream=i)
ctx.set_option('chol.panel_stream', -1)
for ch in (2, 4, 8, 16):
    run('syrk_chunks=%d' % ch, chol__syrk_chunks=ch, chol__syrk_stream=1)
for ch in (4, 8):
    run('syrk_chunks=%d panel_stream=5' % ch, chol__syrk_chunks=ch, chol__syrk_stream=1, chol__panel_stream=5)
    run('syrk_chunks=%d panel_stream=2' % ch, chol__syrk_chunks=ch, chol__syrk_stream=1, chol__panel_stream=2)
ctx.set_option('chol.syrk_chunks', 1)
ctx.set_option('chol.panel_stream', -1)
run('no lookahead', chol__lookahead=0)
run('default again', chol__lookahead=1)
