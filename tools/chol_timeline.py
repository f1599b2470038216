#!/usr/bin/env python
"""From a rocprofv3 rocpd database of one factorisation: how long is no GEMM kernel running
(= exposed panel chain), and where in the factorisation.  Usage: chol_timeline.py <db> [n_steps_to_skip]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# isolate the last factorisation: from the last negate_shift kernel to the first trsv after it
idx = [i for i, r in enumerate(rows) if r[0].startswith('negate_shift')]
i0 = idx[-1]
i1 = next(i for i in range(i0, len(rows)) if rows[i][0].startswith('trsv_fwd'))
seg = rows[i0 + 1:i1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
gem = sorted((r[1], r[2]) for r in seg if r[0].startswith('gemm_nt_sub'))
# union of GEMM intervals
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
# gaps by position in time (deciles of the factorisation)
dec = [0.0] * 10
for s, g in gaps:
    dec[min(9, int(10 * (s - t0) / tot))] += g / 1e6
print('gap ms per time decile:', ' '.join('%.1f' % d for d in dec))
big = sorted(gaps, key=lambda x: -x[1])[:5]
print('largest gaps (ms at t ms):', ', '.join('%.2f@%.0f' % (g / 1e6, (s - t0) / 1e6) for s, g in big))
# per-kernel busy sums on the panel side
agg = {}
for n, s, e in seg:
    k = n.split('(')[0]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('  %-28s %6d launches %10.1f ms summed' % (k[:28], a[0], a[1] / 1e6))
