#!/bin/bash
# usage: tools/ab_lib.sh <variant .so> "<workloads>" "<steps list (0 = the engine's choice)>"
# A/B of an alternative build of libprcg.so (PRCG_LIB) against the default one, same box.
alt=$1; ws=${2:-"s3"}; steps=${3:-"0"}
for w in $ws; do for lib in "" $alt; do for s in $steps; do
  if [ -z "$lib" ]; then unset PRCG_LIB; else export PRCG_LIB=$PWD/$lib; fi
  echo -n "lib=${lib:-default} "; ./tools/sweep.sh $w "$s" "0"
done; done; done
