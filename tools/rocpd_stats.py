#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as a per-kernel stats table
(the equivalent of `--stats` CSV output): calls, total/avg/min/max duration, share."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if 'name' in cols else []
    agg = {}
    for name, s, e in rows:
        short = name.replace('(anonymous namespace)::', '').split('(')[0]
        a = agg.setdefault(short, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = ['%-64s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct')]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('%-64s %8d %14.3f %12.2f %12.2f %12.2f %6.2f%%' % (
            k[:64], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    import os
    if os.environ.get('FULL_NAMES'):  # kernel names in full (Tensile encodes its whole configuration there)
        lines.append('')
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ['FULL_NAMES'])]:
            lines.append('%10.3f ms  %s' % (a[1] / 1e6, k))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
