#!/bin/bash
# A/B builds of the sliced-row kernels: tools/build_sell_variants.sh name "-DFLAG=.." [name flags]...
# -> build_ab/libprcg_<name>.so (select one with PRCG_LIB=...; build_ab/ travels to the GPU box, not into git)
set -e
mkdir -p build_ab
make -s
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude $flags -c new_cg_variants_amd/csrc/prcg_sell.hip -o build_ab/sell_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libprcg_$name.so new_cg_variants_amd/csrc/prcg_kernels.o new_cg_variants_amd/csrc/prcg_win.o build_ab/sell_$name.o new_cg_variants_amd/csrc/prcg_medium.o \
      new_cg_variants_amd/csrc/prcg_engine.o new_cg_variants_amd/csrc/prcg_plan.o new_cg_variants_amd/csrc/prcg_rccl.o -ldl
  echo built build_ab/libprcg_$name.so
done
