#!/bin/bash
# copy what tools/profile_round.sh <tag> left in gpurun_out/prof_<tag>/ into profiles/ under the round's names: tools/copy_profiles.sh r03
tag=${1:-r03}; P=gpurun_out/prof_$tag
cp $P/bench.json profiles/${tag}_s3_bench.json; cp $P/bench_under_rocprof.json profiles/${tag}_s3_bench_under_rocprof.json
cp $P/kernel_stats.csv profiles/${tag}_s3_kernel_stats.csv; cp $P/pmc_dict.json profiles/${tag}_s3_dict_pmc_summary.json
cp $P/pmc_plain.json profiles/${tag}_s3_plain_pmc_summary.json; cp $P/bench_s4b_under_rocprof.json profiles/${tag}_s4b_bench_under_rocprof.json
cp $P/s4b_kernel_stats.csv profiles/${tag}_s4b_kernel_stats.csv; cp $P/pmc_s4b.json profiles/${tag}_s4b_pmc_summary.json
cp $P/bench_s2_under_rocprof.json profiles/${tag}_s2_bench_under_rocprof.json; cp $P/s2_kernel_stats.csv profiles/${tag}_s2_kernel_stats.csv
cp $P/pmc_s2.json profiles/${tag}_s2_pmc_summary.json; cp $P/traffic.json profiles/traffic.json; cp $P/traffic.json profiles/${tag}_traffic_at_profile_time.json
