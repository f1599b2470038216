"""Round 6: where a tile of the fused trailing-update launch spends its time.

python tools/gemm_trace.py [k] [extra options "key=val,..."]   traces the k-th fused launch (default 3) of one benchmark-size
factorisation with the traced instantiation of the production loop (option gemm.trace, chol.hip) and summarises
gemm_trace.bin: per wavefront and tile the s_memtime stamps (shader cycles) of tile entry / first LDS tile visible / end of
the k loop / stores drained, the time parked before the LDS commit (vmcnt) and in the k-tile barrier, and the hardware
slot (XCC, SE, CU).  Prints: phase shares, the distribution of tile starts over the launch (chip-wide lock step?), the
start offset of the two workgroups that share a CU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def run(k, extra):
    from bench import synth_geometries
    from sgdml_amd import _lib

    M, N = int(os.environ.get('AB_M', '1000')), 21
    R, E, F = synth_geometries(N, M, seed=0)
    tp = np.arange(N * (N - 1) // 2, dtype=np.int64)[None]
    y = F.ravel() / np.std(F)
    ctx = _lib.Context(0)
    xd, gd = ctx.desc_from_R(R.reshape(M, -1), N)
    ctx.train_upload(xd, gd, tp)
    for kv in (extra.split(',') if extra else []):
        kk, v = kv.split('=')
        ctx.set_option(kk, float(v))
    for rep in range(2):
        if rep == 1:
            ctx.set_option('gemm.trace', float(k))
        ctx.assemble_K(20.0, False, alloc_extra_rows=1, for_cholesky=1e-10)
        ctx.chol_set_rhs(y)
        info = ctx.chol_factor(1e-10)
        print('rep %d factor %.1f ms info %d' % (rep, ctx.phase_ms('factor')[0], info), flush=True)
    ctx.close()


def summarise(path):
    raw = np.fromfile(path, dtype=np.uint64)
    hdr, w = raw[:8], raw[8:].reshape(-1, 4, 8)
    blocks, tiles2, tiles_m, s_begin, n_super, col0_first, K, M = [int(x) for x in hdr]
    print('launch: %d workgroups / work items (+%d of the second problem), tiles_m %d, K %d, M %d' % (blocks, tiles2, tiles_m, K, M))
    w = w[w[:, 0, 0] > 0].astype(np.int64)  # full tiles that were computed
    nt = len(w)
    # s_memtime ticks: shader-clock cycles on gfx950 (a tile of 64 k-tiles x 64 MFMAs x 64 cycles x 2 wavefronts per SIMD =
    # 524 288 cycles takes ~560 k ticks); the counters of different XCDs are not aligned: only differences on ONE CU are used
    t0, t1, t3, t4 = [w[:, :, i].astype(float) for i in range(4)]
    vm, bar = w[:, :, 4].astype(float), w[:, :, 5].astype(float)
    mx_vm, mx_bar = (w[:, :, 6] >> 32).astype(float), (w[:, :, 6] & 0xffffffff).astype(float)
    tot = t4 - t0
    ideal = 64 * 64 * 64 * 2
    print('tiles traced %d; ideal tile at 2 wavefronts per SIMD: %d cycles' % (nt, ideal))
    for name, v in (('tile total', tot), ('prologue (entry -> first LDS tile visible: 64 C loads + the first A / B tile)', t1 - t0),
                    ('k loop', t3 - t1), ('epilogue (64 stores per lane drained)', t4 - t3),
                    ('  wait before the LDS commit (vmcnt; ~2500 of it are the stamps), sum of 63', vm),
                    ('  LDS commit + barrier, sum of 63 (~6000 of it are the stamps)', bar), ('  worst single vmcnt wait', mx_vm),
                    ('  worst single commit + barrier', mx_bar)):
        q = np.percentile(v, [5, 50, 95])
        print('%-80s mean %9.0f   p5 %9.0f  p50 %9.0f  p95 %9.0f   share of tile %.3f' % (name, v.mean(), q[0], q[1], q[2], v.mean() / tot.mean()))
    print('k loop per k-tile: %.0f cycles for 2 x 4096 of MFMA (two workgroups share a SIMD): %.3f' % ((t3 - t1).mean() / 64, 8192 * 64 / (t3 - t1).mean()))
    a0, a1, a3, a4 = t0[:, 0], t1[:, 0], t3[:, 0], t4[:, 0]
    hw, xcc = w[:, 0, 7] & 0xffffffff, w[:, 0, 7] >> 32
    cu = ((xcc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    print('distinct CUs seen: %d' % len(np.unique(cu)))
    # per CU: how many workgroups are resident / inside their k loop, how long a slot stays empty between two tiles, how a
    # tile's k loop stretches with the time it shares the CU with another k loop
    res = np.zeros(4)
    inloop = np.zeros(4)
    gaps, klen, ov, offs = [], [], [], []
    for c in np.unique(cu):
        idx = np.nonzero(cu == c)[0]
        for lo, hi, acc in ((a0[idx], a4[idx], res), (a1[idx], a3[idx], inloop)):
            ev = sorted([(x, 1) for x in lo] + [(x, -1) for x in hi])
            lvl, prev = 0, ev[0][0]
            for t, d in ev:
                acc[min(lvl, 3)] += t - prev
                prev, lvl = t, lvl + d
        s_, e_ = np.sort(a0[idx]), np.sort(a4[idx])
        for x in e_[:-2]:
            i = np.searchsorted(s_, x)
            if i < len(s_):
                gaps.append(s_[i] - x)
        if len(s_) > 6:
            offs.append(np.diff(s_)[2:-2])
        aa, bb = a1[idx], a3[idx]
        for k in range(len(idx)):
            o = np.clip(np.minimum(bb[k], bb) - np.maximum(aa[k], aa), 0, None)
            o[k] = 0
            klen.append(bb[k] - aa[k])
            ov.append(o.sum())
    print('per CU, first tile start .. last tile end: workgroups resident 0 / 1 / 2: %.3f / %.3f / %.3f;  inside their k loop 0 / 1 / 2: %.3f / %.3f / %.3f'
          % (*(res[:3] / res.sum()), *(inloop[:3] / inloop.sum())))
    gaps = np.array(gaps)
    print('a tile\'s last store drained -> the next tile starts on that CU: p10 %.0f  p50 %.0f  p90 %.0f  mean %.0f cycles' % (*np.percentile(gaps, [10, 50, 90]), gaps.mean()))
    klen, ov = np.array(klen), np.array(ov)
    A = np.vstack([np.ones_like(ov), ov]).T
    coef = np.linalg.lstsq(A, klen, rcond=None)[0]
    print('k-loop length vs the time it overlaps another k loop on the CU: %.0f + %.3f x overlap  (a workgroup ALONE needs %.0f cycles for its 262 144 of MFMA: %.2f of the pipe; two together %.2f)'
          % (coef[0], coef[1], coef[0], 262144 / coef[0], 524288 / (coef[0] + coef[1] * np.percentile(ov, 90)) if coef[1] > 0 else 0))
    for lo, hi in ((0, 0.5), (0.5, 0.9), (0.9, 0.95), (0.95, 1.01)):
        m_ = (ov / klen >= lo) & (ov / klen < hi)
        if m_.sum():
            print('   overlap share %.2f-%.2f: %6d tiles, k loop %.0f cycles' % (lo, hi, m_.sum(), klen[m_].mean()))
    T = np.median(tot)
    offs = np.concatenate(offs) if offs else np.zeros(1)
    h, e = np.histogram(offs / T, bins=[0, 0.02, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.8, 1.0, 2.0])
    print('gap between consecutive tile starts on one CU, in units of the tile time (0.5 = perfectly interleaved):')
    print('   bins %s' % ' '.join('%.2f' % x for x in e))
    print('   n    %s' % ' '.join(str(int(x)) for x in h))
    sk = t3.max(axis=1) - t3.min(axis=1)
    print('wavefront skew at the end of the k loop: mean %.0f cycles p95 %.0f' % (sk.mean(), np.percentile(sk, 95)))


if __name__ == '__main__':
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    extra = sys.argv[2] if len(sys.argv) > 2 else ''
    if k > 0:
        run(k, extra)
        summarise('gemm_trace.bin')
    else:
        summarise(sys.argv[2])  # python tools/gemm_trace.py 0 <file>
