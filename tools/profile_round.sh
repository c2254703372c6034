#!/bin/bash
# usage: tools/profile_round.sh <tag>     (on the GPU box, from the repo root)
# The evidence behind bench.py's roofline object, written to gpurun_out/prof_<tag>/ :
#   bench.json                 the default bench line (no profiler)
#   bench_under_rocprof.json   the same command under rocprofv3 --kernel-trace --stats
#   kernel_stats.csv           rocprofv3's per-kernel summary of that run
#   pmc_dict.json / pmc_plain.json   per-kernel counter means (tools/pmc.sh: one pass per counter group), value
#                              dictionary on (the default) and off (PRCG_VALDICT=0)
#   s4b_kernel_stats.csv / pmc_s4b.json   the same for --workload s4b (FEM-like stand-in for Queen_4147)
#   s2_kernel_stats.csv / pmc_s2.json     the same for --workload s2 (BASELINE config 4 at N = 1)
#   traffic.json               HBM bytes per launch of the dominant kernel from those counters, stamped with the
#                              kernel-source hash bench.py checks before quoting it
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd $root
timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
# (the headline workload only: the other legs launch the same kernel instantiation on smaller operators, which would mix into its average)
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --no-multi-rank-leg --no-workloads > $out/bench_under_rocprof.json 2> $out/trace.err || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
cd $root
./tools/pmc.sh ${tag}_dict --workload s3 --steps 30 --warmup 5 > $out/pmc_dict.log 2>&1
cp gpurun_out/pmc_${tag}_dict.json $out/pmc_dict.json
PRCG_VALDICT=0 ./tools/pmc.sh ${tag}_plain --workload s3 --steps 30 --warmup 5 > $out/pmc_plain.log 2>&1
cp gpurun_out/pmc_${tag}_plain.json $out/pmc_plain.json
rm -rf gpurun_out/pmc_${tag}_dict gpurun_out/pmc_${tag}_plain
# config 5's stand-in (FEM-like rows, sliced-row kernels): per-kernel statistics and counters
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_s4b -- python3 $root/bench.py --workload s4b --no-cpu-baseline --no-multi-rank-leg --no-workloads > $out/bench_s4b_under_rocprof.json 2> $out/trace_s4b.err
cp $(find $out/trace_s4b -name "*kernel_stats.csv" | head -1) $out/s4b_kernel_stats.csv
rm -rf $out/trace_s4b
cd $root
./tools/pmc.sh ${tag}_s4b --workload s4b --steps 30 --warmup 5 > $out/pmc_s4b.log 2>&1
cp gpurun_out/pmc_${tag}_s4b.json $out/pmc_s4b.json
rm -rf gpurun_out/pmc_${tag}_s4b
# config 4 at N = 1 (S2, pattern tiles in sweep order): per-kernel statistics and counters
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_s2 -- python3 $root/bench.py --workload s2 --no-cpu-baseline --no-multi-rank-leg --no-workloads --no-plain-values > $out/bench_s2_under_rocprof.json 2> $out/trace_s2.err
cp $(find $out/trace_s2 -name "*kernel_stats.csv" | head -1) $out/s2_kernel_stats.csv
rm -rf $out/trace_s2
cd $root
./tools/pmc.sh ${tag}_s2 --workload s2 --steps 30 --warmup 5 > $out/pmc_s2.log 2>&1
cp gpurun_out/pmc_${tag}_s2.json $out/pmc_s2.json
rm -rf gpurun_out/pmc_${tag}_s2
python3 tools/make_traffic.py $out/pmc_dict.json $out/pmc_plain.json $out/pmc_s4b.json $out/pmc_s2.json > $out/traffic.json
head -c 1200 $out/bench.json; echo; head -5 $out/kernel_stats.csv; cat $out/traffic.json
