#!/bin/bash
# usage: tools/pmc2.sh <label> <bench.py args...>   -- memory-path counters (latency, TLB, stalls), one pass per group
label=$1; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$label
mkdir -p $out
i=0
for group in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
             "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
             "TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $group --output-format csv -d $out/pass$i -- python3 $root/bench.py "$@" --no-cpu-baseline --no-plain-values --no-multi-rank-leg --no-workloads > $out/pass$i.json 2> $out/pass$i.err || echo "pass $i ($group) failed"
done
python3 $root/tools/pmc_summary.py $out > $root/gpurun_out/pmc_$label.json
rm -rf $out
