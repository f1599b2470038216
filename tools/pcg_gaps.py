#!/usr/bin/env python
"""Host gaps inside the PCG loop from a rocprofv3 kernel trace (rocpd SQLite database) of tools/cg_step_probe.py:

    rocprofv3 --kernel-trace -d DIR -- python tools/cg_step_probe.py 50
    python tools/pcg_gaps.py DIR/.../*_results.db [label]

An iteration is delimited by the preconditioner's first kernel (gemv_t_part_kernel: one launch per PCG iteration, none
elsewhere).  For the iterations of the LAST step of the probe it prints the time per iteration, the time some kernel was
running, and the remainder = time the GPU sat idle between kernels (launch gaps, host round trips, copies)."""
import sqlite3
import sys


def main(path, label=''):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    src = 'kernels' if 'kernels' in tabs else None
    if src is None:
        raise SystemExit('no kernels view in %s (tables: %s)' % (path, tabs[:8]))
    rows = cur.execute('select name, start, end from %s order by start' % src).fetchall()
    rows = [(n.split('(')[0], s, e) for n, s, e in rows]
    marks = [i for i, r in enumerate(rows) if r[0].startswith('gemv_t_part_kernel')]
    if len(marks) < 3:
        raise SystemExit('fewer than 3 PCG iterations in the trace')
    # the last contiguous run of iterations = the last step of the probe (iterations of a step are < 100 ms apart)
    run = [marks[-1]]
    for i in reversed(marks[:-1]):
        if rows[run[0]][1] - rows[i][1] < 100e6:
            run.insert(0, i)
        else:
            break
    a, b = run[0], run[-1]  # kernels [a, b): len(run) - 1 whole iterations
    iters = len(run) - 1
    span = rows[b][1] - rows[a][1]
    busy, last_end = 0, rows[a][1]
    per_kernel = {}
    for n, s, e in rows[a:b]:
        s2 = max(s, last_end)
        if e > s2:
            busy += e - s2
            last_end = e
        k = per_kernel.setdefault(n, [0, 0])
        k[0] += 1
        k[1] += e - s
    print('%s%d PCG iterations: %.1f us per iteration, some kernel running %.1f us, idle between kernels %.1f us (%.2f %%), '
          '%.1f launches per iteration' % (label + ': ' if label else '', iters, span / iters / 1e3, busy / iters / 1e3,
                                           (span - busy) / iters / 1e3, 100.0 * (span - busy) / span, (b - a) / iters))
    for n, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        print('    %-58s %6.2f per it  %9.1f us per it' % (n[:58], c / iters, t / iters / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
