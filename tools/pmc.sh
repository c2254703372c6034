#!/bin/bash
# usage: tools/pmc.sh <label> <bench.py args...>
# Collects hardware counters for the bench command, ONE rocprofv3 pass per counter group
# (gfx950 slot limits; --pmc is never combined with the tracing domains), and prints per-kernel
# means.  Output: gpurun_out/pmc_<label>/pass*/ (raw csv) and gpurun_out/pmc_<label>.json.
label=$1; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$label
mkdir -p $out
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $group --output-format csv -d $out/pass$i -- python3 $root/bench.py "$@" --no-cpu-baseline --no-plain-values --no-multi-rank-leg --no-workloads > $out/pass$i.json 2> $out/pass$i.err || echo "pass $i ($group) failed"
done
python3 $root/tools/pmc_summary.py $out > $root/gpurun_out/pmc_$label.json
cat $root/gpurun_out/pmc_$label.json
