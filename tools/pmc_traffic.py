#!/usr/bin/env python
"""HBM traffic per launch of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not
fit one pass) of `bench.py --steps 1`.  Corrections per MI355X_MICROARCH.md (HBM section): counters are in KiB-like
units of 1 KB?  No -- rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, i.e. reports HALF the bytes: it is doubled here.  WRITE_SIZE
is taken as reported (calibration: the assembly kernel writes a known byte count, printed next to it).
Usage: pmc_traffic.py fetch.csv write.csv out.json"""
import collections, csv, json, sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    seen = set()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        a = agg[k]
        a[1] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            a[0] += 1
            a[2] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return agg


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
print('%-40s %8s %16s %16s %14s' % ('kernel', 'launches', 'read GB (x2 corr.)', 'written GB', 'GB / launch'))
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0, 0])[1] + write.get(k, [0, 0, 0])[1])):
    n = max(fetch.get(k, [0])[0], write.get(k, [0])[0])
    rd = 2.0 * fetch.get(k, [0, 0.0])[1] * 1024.0  # KB -> bytes, gfx950 half-count correction
    wr = write.get(k, [0, 0.0])[1] * 1024.0
    if rd + wr < 1e6:
        continue
    print('%-40s %8d %16.3f %16.3f %14.4f' % (k[:40], n, rd / 1e9, wr / 1e9, (rd + wr) / max(1, n) / 1e9))
    out[k] = {'launches': n, 'read_bytes': rd, 'written_bytes': wr, 'hbm_bytes_per_launch': (rd + wr) / max(1, n)}
res = {}
for key, pat in (('gemm_nt_sub_diag', 'gemm_nt_sub_diag_kernel'), ('gemm_nt_sub', 'gemm_nt_sub_kernel'), ('assemble', 'assemble_strip_kernel')):
    sel = [v for k, v in out.items() if pat in k]  # all instantiations (plain + fused diagonal-block launch)
    if sel:
        n = sum(v['launches'] for v in sel)
        rd, wr = sum(v['read_bytes'] for v in sel), sum(v['written_bytes'] for v in sel)
        res[key] = {'hbm_bytes_per_launch': (rd + wr) / n, 'read_bytes_per_launch': rd / n, 'written_bytes_per_launch': wr / n,
                    'launches': n,
                    'source': 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950 half-count) + WRITE_SIZE, separate passes of '
                              'bench.py --steps 1 (tools/profile_round.sh); memory-side requests incl. Infinity-Cache hits'}
json.dump(res, open(sys.argv[3], 'w'), indent=1)
