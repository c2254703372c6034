#!/bin/bash
# A/B of library builds on one box: tools/ab_win.sh <workload> <out> <per_cu list> <lib>...
wl=$1; out=$2; grids=$3; shift 3
: > $out
for vd in 1 0; do for g in $grids; do for lib in "$@"; do
  PRCG_LIB=$PWD/$lib PRCG_VALDICT=$vd PRCG_WIN_GRID_PER_CU=$g python bench.py --workload $wl --steps 200 --warmup 50 --no-cpu-baseline --no-plain-values --no-multi-rank-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl $lib vd=$vd per_cu=$g', 'it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'spmv_ms', round(r['spmv']['spmv_ms'],4), 'spmm2_ms', round(r['spmv']['spmm2_ms'],4))" >> $out
done; done; done
cat $out
