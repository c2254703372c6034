#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/pmc.sh: HBM bytes per launch of the one-launch
iteration kernel = FETCH_SIZE x 2 (gfx950 counts a 128-byte read request as 64 bytes: MI355X_MICROARCH.md,
HBM) + WRITE_SIZE, collected in separate passes; stamped with the hash of the kernel sources so that bench.py
quotes it only while the kernels are the ones that were measured.
usage: make_traffic.py <workload>:<dict|plain>=<pmc summary json> ..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

entries = {}
for arg in sys.argv[1:]:
    key, path = arg.split('=', 1)
    wl, enc = key.split(':')
    try:
        d = json.load(open(path))
    except Exception:
        continue
    # the one-launch pipelined iteration: NV = 2, epilogue 3 of whichever kernel family the operator runs on
    fused = [(k, v) for k, v in d.items() if ('k_win_tiles<2, 3' in k or 'k_spmv_tiles<2, 3' in k or 'k_sell_tiles<2, 3' in k or 'k_sell_win<2, 3' in k)
             and 'read_bytes_corrected' in v and 'write_bytes' in v]
    if not fused:
        continue
    k, v = max(fused, key=lambda kv: kv[1].get('dispatches', 0))
    entries[f'{wl}:pipe_pr_cg:fused:1:{enc}'] = {
        'bytes_per_launch': v['read_bytes_corrected'] + v['write_bytes'], 'read_bytes': v['read_bytes_corrected'],
        'write_bytes': v['write_bytes'], 'kernel': k, 'duration_us_under_pmc': v['duration_us_under_pmc']}
json.dump({'kernel_sha': bench.kernel_source_sha(), 'entries': entries}, sys.stdout, indent=1)
print()
