#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (optionally only the largest
dispatches), sum of each counter and a few derived ratios."""
import csv, sys, collections
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
top = int(sys.argv[3]) if len(sys.argv) > 3 else 0
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    if pat and pat not in r['Kernel_Name']: continue
    d = disp[(r['Kernel_Name'].split('(')[0], int(r['Dispatch_Id']))]
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    d['_grid'] = int(r['Grid_Size']); d['_dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
byk = collections.defaultdict(list)
for (k, i), d in disp.items(): byk[k].append(d)
for k, ds in byk.items():
    ds.sort(key=lambda d: -d['_dur'])
    if top: ds = ds[:top]
    tot = collections.defaultdict(float)
    for d in ds:
        for c, v in d.items(): tot[c] += v
    print('%s  dispatches=%d  dur_ms=%.3f' % (k, len(ds), tot['_dur'] / 1e6))
    for c in sorted(tot):
        if not c.startswith('_'): print('   %-28s %.4g' % (c, tot[c]))
    wc = tot.get('SQ_WAVE_CYCLES', 0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"):
            if c in tot: print('   %-28s %.1f%% of WAVE_CYCLES' % (c + '/WC', 100 * tot[c] / wc))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in tot and 'GRBM_GUI_ACTIVE' in tot:
        # MFMA busy is summed over SIMDs (1024); GUI_ACTIVE is per-XCD-summed? report raw ratio
        print('   MFMA_BUSY / (GUI_ACTIVE) = %.2f' % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / tot['GRBM_GUI_ACTIVE']))
