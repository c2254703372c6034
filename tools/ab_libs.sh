#!/bin/bash
# same-box A/B of builds over workloads (dictionary and plain-value legs): tools/ab_libs.sh <out> "<workload> ..." "<lib> ..." [ENV=val,ENV2=val]
out=$1; wls=$2; libs=$3; kv=${4:-X_=0}
: > $out
for wl in $wls; do for lib in $libs; do
  env PRCG_LIB=$PWD/$lib $(echo $kv | tr ',' ' ') timeout -k 10 300 python bench.py --workload $wl --steps 300 --warmup 50 --no-cpu-baseline --no-multi-rank-leg --no-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r.get('plain_values') or {}
print('$wl $lib $kv', 'it/s', round(d['value'],1), 'launch_ms', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), '| plain it/s', round(p.get('value',0),1), 'launch_ms', round(p.get('avg_launch_ms',0),4), 'moved_frac', round(p.get('moved_frac',0),3))" >> $out
done; done
cat $out
