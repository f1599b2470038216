"""Symmetry search: the pairwise matching on the device (csrc/perm_match.hip) against the NumPy / SciPy form in
sgdml_amd/utils/perm.py -- same kept pairs, same assignments, same costs -- and its time at the size create_task uses
(1000 geometries).   python tools/perm_match_probe.py [M_big]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_geometries
from sgdml_amd import _lib
from sgdml_amd.utils import perm

ctx = _lib.Context(0)
ctx.profile(True)


def compare(name, R, z, lat_and_inv=None):
    t = time.time(); fh, ch = perm.bipartite_match(R, z, lat_and_inv); th = time.time() - t
    t = time.time(); fd, cd = perm.bipartite_match(R, z, lat_and_inv, ctx=ctx); td = time.time() - t
    ch, cd = ch.toarray(), cd.toarray()
    fin = np.isfinite(ch)
    same_keys = set(fh) == set(fd)
    same_perms = same_keys and all(np.array_equal(fh[k], fd[k]) for k in fh)
    print('%-28s M=%4d N=%3d  kept pairs host %6d device %6d  keys equal %s  perms equal %s  max|cost diff| %.2e (rel %.1e)  '
          'host %.2f s  device %.3f s (device %.2f ms)' % (
              name, R.shape[0], R.shape[1], len(fh), len(fd), same_keys, same_perms, np.abs(ch[fin] - cd[fin]).max(),
              np.abs(ch[fin] - cd[fin]).max() / np.abs(ch[fin]).max(), th, td, ctx.phase_ms('perm_match')[0]), flush=True)
    return same_perms


gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
g = np.load(os.path.join(gold, 'perm_c3.npz'))
ok = compare('perm_c3 R', g['R'], g['z'])
ok &= compare('perm_c3 R2', g['R2'], g['z2'])
ok &= compare('perm_c3 R2 + lattice', g['R2'], g['z2'], (g['lat'], np.linalg.inv(g['lat'])))
for name, R, z, kw, want in (('R', g['R'], g['z'], {}, g['perms']), ('R2', g['R2'], g['z2'], {}, g['perms2']),
                             ('R2+lat', g['R2'], g['z2'], {'lat_and_inv': (g['lat'], np.linalg.inv(g['lat']))}, g['perms3'])):
    got = perm.find_perms(R, z, ctx=ctx, **kw)
    print('find_perms(ctx) %-8s == the reference\'s group: %s (%d elements)' % (name, np.array_equal(got, want), len(got)))
    ok &= np.array_equal(got, want)
# eigenvectors on the device (batched Jacobi) against LAPACK
for N, M in ((8, 40), (21, 200), (42, 100), (70, 40), (100, 50), (101, 20), (130, 12), (150, 6)):
    Rg, _, _ = synth_geometries(N, M, seed=1)
    adj_ = perm._dist_matrices(Rg.reshape(M, N, 3))
    t = time.time(); w_, v_ = np.linalg.eigh(adj_); th = time.time() - t
    ref = np.abs(v_[:, :, ::-1])
    t = time.time(); got = ctx.sym_eig_absv(adj_); td = time.time() - t
    gap = np.min(np.diff(w_, axis=1)) / np.abs(w_).max()
    print('sym_eig N=%3d M=%3d: max||V|dev - |V|lapack| %.2e  (smallest relative eigenvalue gap %.1e)  LAPACK %.3f s  device %.3f s'
          % (N, M, np.abs(got - ref).max(), gap, th, td), flush=True)
    ok &= np.abs(got - ref).max() < 1e-8
rs = np.random.RandomState(0)
for N, M in ((30, 60), (70, 40), (130, 12)):
    base = rs.normal(size=(N, 3)) * 2.0
    R = base[None] + 0.05 * rs.normal(size=(M, N, 3))
    z = rs.choice([1, 6, 8], size=N)
    # half of the geometries with two same-species atoms swapped: assignments that are NOT the identity
    sw = np.where(z == z[0])[0][:2]
    if len(sw) == 2:
        R[::2][:, sw] = R[::2][:, sw[::-1]]
    ok &= compare('random 3 species', R, z)
print('ALL EQUAL' if ok else 'MISMATCH', flush=True)

Mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
R, _, _ = synth_geometries(21, Mb, seed=0)
z = np.array([6] * 7 + [1] * 14)
for rep in range(2):
    t = time.time(); grp = perm.find_perms(R.reshape(Mb, 21, 3), z, ctx=ctx); td = time.time() - t
    print('find_perms(ctx) M=%d N=21: %.3f s wall, device %.2f ms, group %s' % (Mb, td, ctx.phase_ms('perm_match')[0], grp.shape), flush=True)
R, _, _ = synth_geometries(100, 300, seed=0)
t = time.time(); grp = perm.find_perms(R.reshape(300, 100, 3), np.array([6] * 40 + [1] * 60), ctx=ctx); td = time.time() - t
print('find_perms(ctx) M=300 N=100: %.3f s wall, device %.2f ms, group %s' % (td, ctx.phase_ms('perm_match')[0], grp.shape), flush=True)
