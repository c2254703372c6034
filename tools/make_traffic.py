#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/pmc.sh: HBM bytes per launch of the one-launch
iteration kernel = FETCH_SIZE x 2 (gfx950 counts a 128-byte read request as 64 bytes: MI355X_MICROARCH.md,
HBM) + WRITE_SIZE, collected in separate passes; stamped with the hash of the kernel sources so that bench.py
quotes it only while the kernels are the ones that were measured."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

entries = {}
for path, suffix in zip(sys.argv[1:3], (':dict', ':plain')):
    d = json.load(open(path))
    fused = [(k, v) for k, v in d.items() if 'k_win_tiles<2, 3' in k or 'k_spmv_tiles<2, 3' in k]
    k, v = max(fused, key=lambda kv: kv[1].get('dispatches', 0))
    entries['s3:pipe_pr_cg:fused:1' + suffix] = {
        'bytes_per_launch': v['read_bytes_corrected'] + v['write_bytes'], 'read_bytes': v['read_bytes_corrected'],
        'write_bytes': v['write_bytes'], 'kernel': k, 'duration_us_under_pmc': v['duration_us_under_pmc']}
if len(sys.argv) > 3:      # config 5's stand-in: the sliced-row kernels
    d = json.load(open(sys.argv[3]))
    fused = [(k, v) for k, v in d.items() if 'k_sell_tiles<2, 3' in k or 'k_spmv_tiles<2, 3' in k]
    if fused:
        k, v = max(fused, key=lambda kv: kv[1].get('dispatches', 0))
        entries['s4b:pipe_pr_cg:fused:1:plain'] = {
            'bytes_per_launch': v['read_bytes_corrected'] + v['write_bytes'], 'read_bytes': v['read_bytes_corrected'],
            'write_bytes': v['write_bytes'], 'kernel': k, 'duration_us_under_pmc': v['duration_us_under_pmc']}
if len(sys.argv) > 4:      # config 4 at N = 1: S2
    d = json.load(open(sys.argv[4]))
    fused = [(k, v) for k, v in d.items() if 'k_win_tiles<2, 3' in k]
    if fused:
        k, v = max(fused, key=lambda kv: kv[1].get('dispatches', 0))
        entries['s2:pipe_pr_cg:fused:1:dict'] = {
            'bytes_per_launch': v['read_bytes_corrected'] + v['write_bytes'], 'read_bytes': v['read_bytes_corrected'],
            'write_bytes': v['write_bytes'], 'kernel': k, 'duration_us_under_pmc': v['duration_us_under_pmc']}
json.dump({'kernel_sha': bench.kernel_source_sha(), 'entries': entries}, sys.stdout, indent=1)
print()
