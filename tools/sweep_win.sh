#!/bin/bash
# sweep workgroups per CU of the window kernels: tools/sweep_win.sh <workload> <out>
wl=${1:-s3}; out=${2:-gpurun_out/r2/sweep_win.txt}
: > $out
for vd in 1 0; do
for g in 2 3 4 5 6 8 10 12; do
  PRCG_VALDICT=$vd PRCG_WIN_GRID_PER_CU=$g python bench.py --workload $wl --steps 200 --warmup 50 --no-cpu-baseline --no-plain-values --no-multi-rank-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl vd=$vd per_cu=$g', 'it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'spmv_ms', round(r['spmv']['spmv_ms'],4), 'spmm2_ms', round(r['spmv']['spmm2_ms'],4))" >> $out
done; done
cat $out
