#!/bin/bash
# Register / LDS / scratch use of every kernel of one translation unit (no GPU needed):
#   tools/kernel_resources.sh new_cg_variants_amd/csrc/prcg_win.hip [extra -D flags] > out.txt
# one line per kernel: name  VGPRs  AGPRs  SGPRs  scratch  LDS  occupancy (waves per SIMD)
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude --offload-device-only -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | awk '
  /Function Name:/ {name=$NF}
  / VGPRs:/ {v=$NF}
  / AGPRs:/ {a=$NF}
  /TotalSGPRs:/ {s=$NF}
  /ScratchSize/ {sc=$NF}
  /Occupancy/ {o=$NF}
  /LDS Size/ {print name, "vgpr="v, "agpr="a, "sgpr="s, "scratch="sc, "lds="$NF, "occ="o}'
