#!/bin/bash
# usage: tools/profile_round.sh <tag>     (on the GPU box, from the repo root)
# The evidence behind bench.py's roofline object, written to gpurun_out/prof_<tag>/ :
#   bench.json                 the default bench line (no profiler)
#   bench_under_rocprof.json   the same command under rocprofv3 --kernel-trace --stats
#   kernel_stats.csv           rocprofv3's per-kernel summary of that run
#   pmc.json                   per-kernel counter means (tools/pmc.sh: one pass per counter group)
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd $root
timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --no-plain-values --no-multi-rank-leg > $out/bench_under_rocprof.json 2> $out/trace.err || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
cd $root
./tools/pmc.sh $tag --workload s3 --steps 30 --warmup 5 > $out/pmc.log 2>&1
cp gpurun_out/pmc_$tag.json $out/pmc.json
rm -rf gpurun_out/pmc_$tag
head -c 1500 $out/bench.json; echo; head -4 $out/kernel_stats.csv
