#!/bin/bash
# same-box A/B of environment knobs over variants: tools/ab_env.sh <out> <workload> "<variant> ..." "<ENV=val,ENV2=val ...>" ...
out=$1; wl=$2; vars=$3; shift 3
: > $out
for v in $vars; do for kv in "$@"; do
  env $(echo $kv | tr ',' ' ') timeout -k 10 300 python bench.py --workload $wl --variant $v --steps 300 --warmup 50 --no-cpu-baseline --no-multi-rank-leg --no-workloads --no-plain-values 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl $v $kv', 'it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'upd_ms', round(r['update_kernel_ms'],4), 'frac', round(r['frac'],3))" >> $out
done; done
cat $out
