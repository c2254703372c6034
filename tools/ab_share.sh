#!/bin/bash
# same-box A/B: tools/ab_share.sh <out> <workloads> <lib>...   (each lib with PRCG_WIN_SHARE=1 and 0, dict and plain)
out=$1; wls=$2; shift 2
: > $out
for wl in $wls; do for lib in "$@"; do for sh in 1 0; do
  PRCG_LIB=$PWD/$lib PRCG_WIN_SHARE=$sh timeout -k 10 300 python bench.py --workload $wl --steps 300 --warmup 50 --no-cpu-baseline --no-multi-rank-leg --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r.get('plain_values') or {}
print('$wl $lib share=$sh', 'dict it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'spmv_ms', round(r['spmv']['spmv_ms'],4), '| plain it/s', round(p.get('value',0),1), 'spmv_ms', round((p.get('spmv') or {}).get('spmv_ms',0),4))" >> $out
done; done; done
cat $out
