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
# fused steps: GEMM1 (gemm_nt_sub_kernel) -> SYRK + diagonal block (gemm_nt_sub_diag_kernel) -> panel_trsm_kernel
n_order = int(sys.argv[2]) if len(sys.argv) > 2 else 63000
nbw = 512
steps = []
for i in range(1, len(seg) - 1):
    if 'gemm_nt_sub_diag_kernel' in seg[i][0] and 'gemm_nt_sub_kernel' in seg[i - 1][0] and 'panel_trsm' in seg[i + 1][0]:
        steps.append(i)
if steps:
    g1 = sum(seg[i - 1][2] - seg[i - 1][1] for i in steps)
    sy = sum(seg[i][2] - seg[i][1] for i in steps)
    tr = sum(seg[i + 1][2] - seg[i + 1][1] for i in steps)
    gap_a = sum(seg[i][1] - seg[i - 1][2] for i in steps)          # GEMM1 -> SYRK
    gap_b = sum(seg[i + 1][1] - seg[i][2] for i in steps)          # SYRK -> trsm
    gap_c = sum(seg[i + 2][1] - seg[i + 1][2] for i in steps if i + 2 < len(seg))  # trsm -> next GEMM1
    fl1 = sum(2.0 * (n_order + 1 - (s + 1) * nbw) * nbw * nbw for s in range(len(steps)))
    fl2 = sum(float(n_order + 1 - (s + 2) * nbw) * (n_order - (s + 2) * nbw) * nbw for s in range(len(steps)))
    print('%d fused steps: GEMM1 %.1f ms (%.1f TFLOP/s), SYRK+diag %.1f ms (%.1f TFLOP/s), panel_trsm %.1f ms; launch gaps '
          'GEMM1->SYRK %.2f ms, SYRK->trsm %.2f ms, trsm->GEMM1 %.2f ms' % (len(steps), g1 / 1e6, fl1 / g1 / 1e3, sy / 1e6,
          fl2 / sy / 1e3, tr / 1e6, gap_a / 1e6, gap_b / 1e6, gap_c / 1e6))
    for i in steps[::max(1, len(steps) // 6)]:
        print('  t=%7.1f ms: GEMM1 %6.1f us | gap %5.1f | SYRK %8.1f us | gap %5.1f | trsm %6.1f us | gap %5.1f' % (
            (seg[i][1] - t0) / 1e6, (seg[i - 1][2] - seg[i - 1][1]) / 1e3, (seg[i][1] - seg[i - 1][2]) / 1e3,
            (seg[i][2] - seg[i][1]) / 1e3, (seg[i + 1][1] - seg[i][2]) / 1e3, (seg[i + 1][2] - seg[i + 1][1]) / 1e3,
            (seg[i + 2][1] - seg[i + 1][2]) / 1e3 if i + 2 < len(seg) else 0.0))
