#!/bin/bash
# same-box A/B of builds over variants: tools/ab_variants.sh <out> <workload> "<lib> ..." "<variant> ..."
out=$1; wl=$2; libs=$3; vars=$4
: > $out
for v in $vars; do for lib in $libs; do
  env PRCG_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $wl --variant $v --steps 300 --warmup 50 --no-cpu-baseline --no-multi-rank-leg --no-workloads --no-plain-values 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl $v $lib', 'it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'upd_ms', round(r['update_kernel_ms'],4))" >> $out
done; done
cat $out
