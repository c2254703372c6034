#!/bin/bash
# like ab_knobs.sh, for the plain-values leg (PRCG_VALDICT=0)
out=$1; wl=$2; libs=$3; shift 3
: > $out
for lib in $libs; do for kv in "$@"; do
  env PRCG_VALDICT=0 PRCG_LIB=$PWD/$lib $(echo $kv | tr ',' ' ') timeout -k 10 300 python bench.py --workload $wl --steps 300 --warmup 50 --no-cpu-baseline --no-multi-rank-leg --no-workloads --no-plain-values 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl plain $lib $kv', 'it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'spmv_ms', round(r['spmv']['spmv_ms'],4), 'spmm2_ms', round(r['spmv']['spmm2_ms'],4))" >> $out
done; done
cat $out
