#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r4
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "sliced" > $out/t_sliced2.log 2>&1 || { tail -30 $out/t_sliced2.log; exit 1; }
tail -3 $out/t_sliced2.log
timeout -k 10 300 python tools/sell_time.py s4b_80 - PRCG_SELL_PLANES=8 PRCG_SELL_NT=1 PRCG_SELL_GRID_PER_CU=3 > $out/s4b80_2.jsonl 2> $out/s4b80_2.err || { tail $out/s4b80_2.err; exit 1; }
cut -c1-330 $out/s4b80_2.jsonl
timeout -k 10 600 python tools/sell_time.py s4b - PRCG_SELL_PLANES=8 PRCG_SELL_NT=1 PRCG_SELL_GRID_PER_CU=3 PRCG_SELL=0 > $out/s4b_2.jsonl 2> $out/s4b_2.err || { tail $out/s4b_2.err; exit 1; }
cut -c1-330 $out/s4b_2.jsonl
timeout -k 10 600 python tools/sell_time.py s4c - PRCG_SELL_PLANES=8 PRCG_SELL_SIGMA=256 PRCG_SELL_SIGMA=4096 PRCG_SELL_NT=1 PRCG_SELL=0 > $out/s4c_2.jsonl 2> $out/s4c_2.err || { tail $out/s4c_2.err; exit 1; }
cut -c1-330 $out/s4c_2.jsonl
grep generated $out/*_2.err
