#!/usr/bin/env python
"""From a rocprofv3 rocpd database of a run that factors the benchmark matrix: for the LAST factorisation,
how long is no GEMM kernel running (= exposed panel chain), where in the factorisation, per-kernel sums, and a
per-panel breakdown of the non-GEMM kernels for a few panels.  Usage: chol_timeline.py <db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.split('(')[0].replace('void ', '')
# the last factorisation: from the last assembly kernel to the first triangular-solve kernel after it
idx = [i for i, r in enumerate(rows) if 'assemble_wave_kernel' in r[0] or 'assemble_strip_kernel' in r[0] or r[0].startswith('negate_shift')]
i0 = idx[-1]
i1 = next((i for i in range(i0, len(rows)) if short(rows[i][0]).startswith('trsv')), len(rows))
seg = [r for r in rows[i0 + 1:i1] if not r[0].startswith('__amd_rocclr')]
t0, t1 = seg[0][1], max(r[2] for r in seg)
gem = sorted((r[1], r[2]) for r in seg if 'gemm_nt_sub' in r[0])
busy, gaps, cur_s, cur_e = 0, [], gem[0][0], gem[0][1]
for s, e in gem[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e, s - cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = t1 - t0
print('factorisation %.1f ms; some GEMM running %.1f ms; no GEMM running %.1f ms (%d gaps + head %.2f ms + tail %.2f ms)' % (
    tot / 1e6, busy / 1e6, (tot - busy) / 1e6, len(gaps), (gem[0][0] - t0) / 1e6, (t1 - cur_e) / 1e6))
dec = [0.0] * 10
for s, g in gaps:
    dec[min(9, int(10 * (s - t0) / tot))] += g / 1e6
print('gap ms per time decile:', ' '.join('%.1f' % d for d in dec))
agg = {}
for n, s, e in seg:
    a = agg.setdefault(short(n), [0, 0])
    a[0] += 1; a[1] += e - s
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('  %-34s %6d launches %10.1f ms summed %9.1f us avg' % (k[:34], a[0], a[1] / 1e6, a[1] / a[0] / 1e3))
# K = 64 in-block GEMMs vs the big ones (by duration rank is unreliable: split by launch order inside a panel)
# per-panel breakdown: a panel starts at each panel_trsm_kernel (new path) -- list what ran between big GEMMs
pt = [i for i, r in enumerate(seg) if 'panel_trsm' in r[0]]
if pt:
    print('per-panel non-GEMM chain (sampled panels): wall from first chain kernel to end of panel_trsm; pieces summed')
    for p in pt[::max(1, len(pt) // 8)]:
        # walk back over the chain kernels of the diagonal block
        j = p
        while j > 0 and (seg[j - 1][2] - seg[j - 1][1]) < 3e5 and seg[p][1] - seg[j - 1][1] < 3e6:
            j -= 1
        chain = seg[j:p + 1]
        parts = {}
        for n, s, e in chain:
            parts[short(n)] = parts.get(short(n), 0) + (e - s)
        print('  t=%7.1f ms: wall %6.1f us | %s' % ((seg[p][1] - t0) / 1e6, (seg[p][2] - chain[0][1]) / 1e3,
                                                  ', '.join('%s %.0f' % (k[:18], v / 1e3) for k, v in parts.items())))
# one panel pair in the middle of the run, launch by launch (schedule of chol_factor_device: merged SYRK half 1 with the
# diagonal block of a -> prep + row-local solve of a -> SYRK half 2 with the K = NB update of b and its diagonal block ->
# prep + row-local solve of b)
dk = [i for i, r in enumerate(seg) if 'gemm_nt_sub_diag_kernel' in r[0]]
if len(dk) > 8:
    i0 = dk[len(dk) // 2 // 2 * 2]
    print('launch sequence around t = %.1f ms:' % ((seg[i0][1] - t0) / 1e6))
    for n, s_, e_ in seg[i0:i0 + 8]:
        print('  %-34s %9.1f us' % (short(n)[:34], (e_ - s_) / 1e3))

# optional: launch-by-launch listing of the last TAIL_MS milliseconds (start offset from the end, duration, queue)
import os
tail_ms = float(os.environ.get('TAIL_MS', '0'))
if tail_ms > 0:
    try:
        rows_q = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    except Exception:
        rows_q = [(n, s_, e_, -1) for n, s_, e_ in rows]
    print('last %.1f ms of the factorisation (t relative to its end, us):' % tail_ms)
    for n, s_, e_, q in rows_q:
        if s_ >= t1 - tail_ms * 1e6 and e_ <= t1 + 1000 and not n.startswith('__amd_rocclr'):
            print('  t=%9.1f  %-30s %8.1f us  q%s' % ((s_ - t1) / 1e3, short(n)[:30], (e_ - s_) / 1e3, q))
