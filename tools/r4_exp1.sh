#!/bin/bash
# round 4, experiment 1: the sliced-row kernels with the interleaved slice table / nontemporal streams / sorting windows
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r4
mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "sliced" > $out/t_sliced.log 2>&1 || { tail -30 $out/t_sliced.log; exit 1; }
tail -3 $out/t_sliced.log
timeout -k 10 300 python tools/sell_time.py s4b_80 PRCG_SELL_PLANES=0 - PRCG_SELL_NT=1 PRCG_SELL_PLANES=0,PRCG_SELL_NT=1 PRCG_SELL_PLANES=4 PRCG_SELL_PLANES=16 > $out/s4b80.jsonl 2> $out/s4b80.err || { tail $out/s4b80.err; exit 1; }
cat $out/s4b80.jsonl
PRCG_LIB=$root/build_ab/libprcg_nogather.so timeout -k 10 300 python tools/sell_time.py s4b_80 PRCG_SELL_PLANES=0 - > $out/s4b80_nogather.jsonl 2> $out/s4b80_nogather.err
cat $out/s4b80_nogather.jsonl
timeout -k 10 600 python tools/sell_time.py s4b PRCG_SELL_PLANES=0 - PRCG_SELL_NT=1 PRCG_SELL_PLANES=0,PRCG_SELL_NT=1 PRCG_SELL_PLANES=4 PRCG_SELL_PLANES=16 PRCG_SELL_GRID_PER_CU=3 PRCG_SELL_GRID_PER_CU=4 PRCG_SELL=0 > $out/s4b.jsonl 2> $out/s4b.err || { tail $out/s4b.err; exit 1; }
cat $out/s4b.jsonl
PRCG_LIB=$root/build_ab/libprcg_nogather.so timeout -k 10 400 python tools/sell_time.py s4b PRCG_SELL_PLANES=0 - > $out/s4b_nogather.jsonl 2> $out/s4b_nogather.err
cat $out/s4b_nogather.jsonl
timeout -k 10 600 python tools/sell_time.py s4c - PRCG_SELL_PLANES=0 PRCG_SELL_SIGMA=256 PRCG_SELL_SIGMA=1024 PRCG_SELL_SIGMA=4096 PRCG_SELL_NT=1 PRCG_SELL=0 > $out/s4c.jsonl 2> $out/s4c.err || { tail $out/s4c.err; exit 1; }
cat $out/s4c.jsonl
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $out/pmc_s4b_$c -- python3 $root/tools/sell_time.py s4b iters=30 warm=12 PRCG_SELL_PLANES=0 - PRCG_SELL_NT=1 PRCG_SELL_PLANES=0,PRCG_SELL_NT=1 > $out/pmc_s4b_$c.jsonl 2> $out/pmc_s4b_$c.err || echo "pmc $c failed"
done
cd $root
mkdir -p $out/pmc_s4b && cp -r $out/pmc_s4b_FETCH_SIZE $out/pmc_s4b/pass1 && cp -r $out/pmc_s4b_WRITE_SIZE $out/pmc_s4b/pass2
python3 tools/sell_pmc.py $out/pmc_s4b "k_sell_tilesILi2ELi3" 42 > $out/pmc_s4b.json 2>&1
python3 tools/sell_pmc.py $out/pmc_s4b "k_sell_tiles<2, 3" 42 >> $out/pmc_s4b.json 2>&1
cat $out/pmc_s4b.json
rm -rf $out/pmc_s4b_FETCH_SIZE $out/pmc_s4b_WRITE_SIZE $out/pmc_s4b/pass*/*/*.db 2>/dev/null
du -sh $out
