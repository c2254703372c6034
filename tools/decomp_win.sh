# what the window kernels' one-launch iteration pays for, by timing builds (tools/build_variants.sh; wrong results): one process per library
#   tools/decomp_win.sh "<stencil_time args>" lib1 lib2 ...      e.g.  tools/decomp_win.sh "216 216 216" w_base w_nostore w_nowalk w_nopages w_bare
mkdir -p gpurun_out/r4
args=$1; shift
log=gpurun_out/r4/decomp_win.log; : > $log
for rep in 1 2; do
  for v in "$@"; do
    echo "## $v" >> $log
    PRCG_LIB=$PWD/build_ab/libprcg_$v.so timeout -k 10 200 python ${TOOL:-tools/stencil_time.py} $args >> $log 2>&1 || exit 1
  done
done
grep -v "^/opt\|^$" $log | paste - - | cut -c1-220
