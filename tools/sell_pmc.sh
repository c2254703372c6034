#!/bin/bash
# usage: tools/sell_pmc.sh <label> <workload> <config> : counters of the one-launch iteration of tools/sell_time.py, one rocprofv3 pass per group
label=$1; wl=$2; cfg=$3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r4/pmc_$label
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $out/pass$i -- python3 $root/tools/sell_time.py $wl iters=20 warm=10 $cfg > $out/pass$i.jsonl 2> $out/pass$i.err || echo "pass $i ($group) failed"
done
cd $root
python3 tools/pmc_summary.py $out > $root/gpurun_out/r4/pmc_$label.json
rm -rf $out/pass*/
python3 - <<PY
import json
d=json.load(open('$root/gpurun_out/r4/pmc_$label.json'))
for k,v in d.items():
    if 'k_sell' in k or 'k_spmv_tiles<2, 3' in k:
        if v.get('dispatches',0) >= 20: print(k[:60], json.dumps({a: (round(b,1) if isinstance(b,float) else b) for a,b in v.items()}))
PY
