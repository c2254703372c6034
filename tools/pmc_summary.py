#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc.sh (rocprofv3 csv, one row per
dispatch and counter).  FETCH_SIZE / WRITE_SIZE are reported in KB as rocprofv3 gives them;
on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes (MI355X_MICROARCH.md, HBM):
`read_bytes_corrected` doubles it."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, 'pass*', '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name']
            short = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('prcg::', '')
            a = acc[short][row['Counter_Name']]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
            d = acc[short]['duration_us_under_pmc']
            d[0] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3
            d[1] += 1
out = {}
for k, cs in acc.items():
    d = {c: v[0] / v[1] for c, v in cs.items()}
    d['dispatches'] = max(v[1] for v in cs.values())
    if 'FETCH_SIZE' in d:
        d['read_bytes_corrected'] = 2 * 1024 * d['FETCH_SIZE']
    if 'WRITE_SIZE' in d:
        d['write_bytes'] = 1024 * d['WRITE_SIZE']
    if 'TCC_HIT_sum' in d and d['TCC_HIT_sum'] + d.get('TCC_MISS_sum', 0) > 0:
        d['l2_hit_rate'] = d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum'])
    out[k] = d
keep = {k: v for k, v in out.items() if 'k_spmv_tiles' in k or 'k_pipe_update' in k or 'k_win_tiles' in k or 'k_sell_tiles' in k or 'k_sell_win' in k}
json.dump(keep or out, sys.stdout, indent=1)
print()
