#!/bin/bash
# usage: tools/profile_round.sh <tag> [legs]     (on the GPU box, from the repo root; legs default: all)
# The evidence behind bench.py's roofline object, written to gpurun_out/prof_<tag>/ :
#   bench.json                   the default bench line (no profiler)
#   <leg>_bench_under_rocprof.json / <leg>_kernel_stats.csv    the leg's bench command under rocprofv3 --kernel-trace --stats
#   <leg>_pmc.json               per-kernel counter means (tools/pmc.sh: one pass per counter group, never combined with tracing)
#   traffic.json                 HBM bytes per launch of the dominant kernel of every leg, stamped with the kernel-source hash
#                                bench.py checks before quoting it
# legs: s3 (headline, value dictionary), s3plain (PRCG_VALDICT=0), s1, s1plain, s2, s2plain, s4b, s4c, s4 -- BASELINE configs 3, 4, 5
tag=${1:-x}
legs=${2:-"s3 s3plain s1 s1plain s2 s2plain s4b s4c s4"}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd $root
timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
targs=""
for leg in $legs; do
  wl=${leg%plain}; enc=dict; envs=""
  if [ "$leg" != "$wl" ]; then enc=plain; envs="PRCG_VALDICT=0"; fi
  case $wl in s4b|s4c|s4) enc=plain;; esac
  cd /tmp && export TMPDIR=/tmp
  # (the leg's workload only: the other legs of a default run launch the same kernel instantiation on other operators, which would mix into its average)
  ( export $envs X_=0; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$leg -- python3 $root/bench.py --workload $wl --no-cpu-baseline --no-multi-rank-leg --no-workloads --no-plain-values > $out/${leg}_bench_under_rocprof.json 2> $out/trace_$leg.err )
  cp $(find $out/trace_$leg -name "*kernel_stats.csv" | head -1) $out/${leg}_kernel_stats.csv 2>/dev/null
  rm -rf $out/trace_$leg
  cd $root
  ( export $envs X_=0; ./tools/pmc.sh ${tag}_$leg --workload $wl --steps 30 --warmup 5 > $out/pmc_$leg.log 2>&1 )
  cp gpurun_out/pmc_${tag}_$leg.json $out/${leg}_pmc.json
  rm -rf gpurun_out/pmc_${tag}_$leg
  targs="$targs $wl:$enc=$out/${leg}_pmc.json"
  echo "leg $leg done"
done
python3 tools/make_traffic.py $targs > $out/traffic.json
head -c 600 $out/bench.json; echo; cat $out/traffic.json
