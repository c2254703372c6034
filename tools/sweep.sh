#!/bin/bash
# usage: tools/sweep.sh <workload> "<steps list (0=auto)>" "<grid/CU list (0=auto)>"
w=${1:-s3}; steps=${2:-"1 2 4"}; grids=${3:-"0"}
mkdir -p gpurun_out
for s in $steps; do for g in $grids; do
  if [ $g = 0 ]; then unset PRCG_GRID_PER_CU; else export PRCG_GRID_PER_CU=$g; fi
  if [ $s = 0 ]; then unset PRCG_TILE_STEPS; else export PRCG_TILE_STEPS=$s; fi
  timeout -k 10 200 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-plain-values > gpurun_out/sw.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/sw.json"))
print("$w steps=$s grid/CU=$g", round(d["value"],1), "it/s spmm2_ms", round(d["roofline"]["avg_launch_ms"],4), "GB/s", round(d["roofline"]["achieved"]), "upd_ms", round(d["roofline"]["update_kernel_ms"],4), "spmv GB/s", round(d["spmv"]["spmv_GBps"]), "spmm2 alone", round(d["spmv"]["spmm2_GBps"]))
PY
done; done
