#!/bin/bash
# copy what tools/profile_round.sh <tag> left in gpurun_out/prof_<tag>/ into profiles/ under the round's names: tools/copy_profiles.sh r04
tag=${1:-r04}; P=gpurun_out/prof_$tag
cp $P/bench.json profiles/${tag}_s3_bench.json
for f in $P/*_bench_under_rocprof.json $P/*_kernel_stats.csv; do b=$(basename $f); cp $f profiles/${tag}_$b; done
for f in $P/*_pmc.json; do b=$(basename $f _pmc.json); cp $f profiles/${tag}_${b}_pmc_summary.json; done
cp $P/traffic.json profiles/traffic.json; cp $P/traffic.json profiles/${tag}_traffic_at_profile_time.json
