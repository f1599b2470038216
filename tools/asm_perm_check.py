"""assemble_perm.hip on the GPU: parity against the oracle over its code paths (option sets), with a per-block error map
when something is off, then timings at the shapes the round-2 review named.
    python tools/asm_perm_check.py check          # parity (small cases, all modes)
    python tools/asm_perm_check.py time [quick]    # timings (the round-1 LDS kernel it was compared with is gone)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gdml_oracle as orc
from sgdml_amd import _lib


def group_perms(N, kind):
    """Closed permutation groups on N atoms: 'c3xc2' (6), 'c3^3' (27), 'c2xc2' (4), 'id'."""
    idt = np.arange(N)
    def rot3(a):  # 3-cycle on atoms a, a+1, a+2
        p = idt.copy(); p[[a, a + 1, a + 2]] = [a + 1, a + 2, a]; return p
    def swap(a):
        p = idt.copy(); p[[a, a + 1]] = [a + 1, a]; return p
    def close(gens):
        G = {tuple(idt)}
        frontier = [idt]
        while frontier:
            nxt = []
            for g in frontier:
                for h in gens:
                    c = tuple(np.asarray(g)[h])
                    if c not in G:
                        G.add(c); nxt.append(np.array(c))
            frontier = nxt
        out = sorted(G)
        out.remove(tuple(idt))
        return np.array([tuple(idt)] + out)
    if kind == 'id': return idt[None]
    if kind == 'c2xc2': return close([swap(0), swap(N - 2)])
    if kind == 'c3xc2': return close([rot3(0), swap(N - 2)])
    if kind == 'c3^3': return close([rot3(0), rot3(3), rot3(N - 3)])
    raise ValueError(kind)


def err_map(K, Ko, M, N3, mask=None):
    d = np.abs(K - Ko)
    if mask is not None: d = np.where(mask, d, 0.0)
    bl = d[:M * N3, :M * N3].reshape(M, N3, M, N3).max(axis=(1, 3))
    return bl


def check_case(N, M, kind, opts, seed=0, sig=13.0):
    ds = orc.synth_dataset(N, M, seed=seed, jitter=0.3)
    xo, go = orc.desc_from_R(ds['R'].reshape(M, -1))
    perms = group_perms(N, kind)
    tp = orc.tril_perms_from_atom_perms(perms)
    lin = orc.tril_perms_lin_from_tril_perms(tp)
    N3 = 3 * N
    n = M * N3
    Ko = orc.assemble_K(xo, go, lin, sig)
    KoE = orc.assemble_K(xo, go, lin, sig, use_E_cstr=True)
    scale = np.abs(Ko).max()
    c = _lib.Context(0)
    c.set_option('asm.wave', 0); c.set_option('asm.strip', 0)   # force the general kernel also where P = 1, N <= 21
    for k, v in opts.items(): c.set_option(k, v)
    c.train_upload(xo, go, tp)
    res = {}
    K = c.assemble_K(sig, False, to_host=True)
    res['full'] = np.abs(K - Ko).max() / scale
    bad = None
    if res['full'] > 1e-12: bad = ('full', err_map(K, Ko, M, N3))
    lam = 1e-7
    c.assemble_K(sig, False, alloc_extra_rows=1, for_cholesky=lam)
    A = c.K_to_host()[:n]
    blk_lower = np.kron(np.tril(np.ones((M, M))), np.ones((N3, N3))).astype(bool)
    Ao = -Ko + lam * np.eye(n)
    res['lower'] = np.abs((A - Ao)[blk_lower]).max() / scale
    if res['lower'] > 1e-12 and bad is None: bad = ('lower', err_map(A, Ao, M, N3, blk_lower))
    KE = c.assemble_K(sig, True, to_host=True)
    res['ecstr'] = np.abs(KE - KoE).max() / np.abs(KoE).max()
    if res['ecstr'] > 1e-12 and bad is None:
        bad = ('ecstr rows>=n: %.2e' % (np.abs(KE - KoE)[n:].max()), err_map(KE[:n, :n], KoE[:n, :n], M, N3))
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.choice(n, size=min(n, 2 * N3 + 7), replace=False))
    Kc = c.assemble_K(sig, False, idx=idx, to_host=True)
    res['index'] = np.abs(Kc - Ko[:, idx]).max() / scale
    p0, p1 = M // 3, M // 3 + max(1, M // 2)
    Kp = c.assemble_K(sig, False, points=(p0, p1), to_host=True)
    res['points'] = np.abs(Kp - Ko[:, p0 * N3:p1 * N3]).max() / scale
    c.close()
    ok = all(v <= 1e-12 for v in res.values())
    print('%s N=%-3d M=%-3d %-6s P=%-2d %-40s %s' % ('ok  ' if ok else 'FAIL', N, M, kind, len(perms), str(opts),
                                                     ' '.join('%s=%.1e' % kv for kv in res.items())), flush=True)
    if bad is not None:
        np.set_printoptions(linewidth=200, precision=1)
        print('   first failing mode:', bad[0]); print('   per-block max error (rows i, cols j):'); print(bad[1][:8, :8])
    return ok


def do_check():
    ok = True
    cases = [(5, 7, 'c2xc2'), (9, 9, 'c3xc2'), (12, 8, 'c3^3'), (21, 7, 'c2xc2'), (21, 5, 'id'), (24, 4, 'c3xc2'),
             (30, 5, 'c3xc2'), (42, 4, 'c3^3'), (47, 3, 'id'), (60, 3, 'c2xc2'), (100, 2, 'id'), (101, 2, 'c2xc2'), (2, 40, 'id'),
             (3, 30, 'id')]
    for N, M, kind in cases:
        ok &= check_case(N, M, kind, {})
    # option sets: no image / no G_j table / group sizes / slow stores / 6 atoms per wavefront for small molecules
    for N, M, kind in [(9, 9, 'c3xc2'), (21, 7, 'c2xc2'), (42, 4, 'c3^3')]:
        for opts in [{'asm.perm_level': 0}, {'asm.perm_level': 1}, {'asm.perm_level': 2}, {'asm.perm_nimg': 1}, {'asm.perm_pg': 1},
                     {'asm.perm_pg': 2}, {'asm.perm_fast_store': 0}, {'asm.perm_na': 6}, {'asm.perm_i_chunk': 2}]:
            ok &= check_case(N, M, kind, opts)
    print('ALL OK' if ok else 'SOME FAILED')
    return ok


def time_case(N, M, kind, opts, lower=False, reps=4, label=''):
    from bench import synth_geometries
    R, E, F = synth_geometries(N, M, seed=0)
    perms = group_perms(N, kind)
    tp = orc.tril_perms_from_atom_perms(perms)
    c = _lib.Context(0)
    for k, v in opts.items(): c.set_option(k, v)
    xd, gd = c.desc_from_R(R.reshape(M, -1), N)
    c.train_upload(xd, gd, tp)
    ts = []
    try:
        for _ in range(reps):
            if lower: c.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
            else: c.assemble_K(20.0, False)
            ts.append(c.phase_ms('assemble')[0])
    except Exception as e:
        print('N=%d M=%d %s %s: %s' % (N, M, kind, opts, e)); c.close(); return None
    n = M * 3 * N
    by = 8.0 * (0.5 * M * (M + 1) if lower else M * M) * (3 * N) ** 2
    t = min(ts)
    P = len(perms)
    D = N * (N - 1) // 2
    valu = (0.5 * M * (M + 1) if lower else M * M) * P * (2.0 * (3 * N) ** 2 + 100.0 * D)   # SURVEY 8(d) flop model
    print('%-6s N=%-3d M=%-4d P=%-2d %-5s %-34s %8.2f ms  %6.0f GB/s = %.3f of HBM | %.2f TFLOP/s (model) = %.3f of fp64 VALU' % (
        label, N, M, P, 'lower' if lower else 'full', str(opts), t, by / t / 1e6, by / t / 1e6 / 8000, valu / t / 1e9, valu / t / 1e9 / 78.6), flush=True)
    c.close()
    return t


def do_time(quick):
    shapes = [(21, 1000, 'c2xc2'), (42, 300, 'id'), (12, 1500, 'c3xc2'), (42, 300, 'c3^3'), (60, 200, 'id'), (9, 2000, 'c3xc2'),
              (100, 120, 'id'), (30, 400, 'id')]
    if quick: shapes = shapes[:4]
    for N, M, kind in shapes:
        time_case(N, M, kind, {}, label='new')
    # variants at the two shapes of interest
    for N, M, kind in [(21, 1000, 'c2xc2'), (42, 300, 'c3^3')]:
        for opts in [{'asm.perm_level': 2}, {'asm.perm_level': 1}, {'asm.perm_level': 0}, {'asm.perm_nimg': 1}, {'asm.perm_pg': 2},
                     {'asm.perm_pg': 1}, {'asm.perm_na': 6}, {'asm.perm_fast_store': 0}, {'asm.perm_i_chunk': 32}, {'asm.perm_i_chunk': 4}]:
            time_case(N, M, kind, opts, label='new')
        time_case(N, M, kind, {}, lower=True, label='new')


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'check'
    if mode == 'check':
        sys.exit(0 if do_check() else 1)
    do_time(len(sys.argv) > 2 and sys.argv[2] == 'quick')
