"""Splits the dispatches of the iteration kernel recorded by `rocprofv3 --pmc X -- python3 tools/sell_time.py ...` into the
configurations of that run (each issued `launches` dispatches of it, in order) and prints the counter means per configuration:
   sell_pmc.py <dir with *counter_collection.csv> <kernel name substring> <launches per configuration> [skip]"""
import csv, glob, json, os, sys
from collections import defaultdict
root, sub, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = defaultdict(dict)
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if sub in r['Kernel_Name']:
                rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
                rows[int(r['Dispatch_Id'])]['_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
ids = sorted(rows)
out = []
for c in range(len(ids) // per):
    chunk = ids[c * per:(c + 1) * per][per // 3:]          # the first third: warm-up
    acc = defaultdict(float)
    for i in chunk:
        for k, v in rows[i].items():
            acc[k] += v
    d = {k: v / len(chunk) for k, v in acc.items()}
    if 'FETCH_SIZE' in d:
        d['read_GB_corrected'] = 2 * 1024 * d['FETCH_SIZE'] * 1e-9
    if 'WRITE_SIZE' in d:
        d['write_GB'] = 1024 * d['WRITE_SIZE'] * 1e-9
    d['config'] = c
    d['dispatches'] = len(chunk)
    out.append(d)
print(json.dumps(out, indent=1))
