# what the sliced-row iteration pays for, by timing builds (tools/build_sell_variants.sh; wrong results): one process per library
mkdir -p gpurun_out/r4
wl=${1:-s4b}
for rep in 1 2; do
  for v in base nog nog_noepi nog_noepi_noown g_noepi_noown; do
    echo "## $v" >> gpurun_out/r4/decomp_$wl.log
    PRCG_LIB=$PWD/build_ab/libprcg_$v.so timeout -k 10 200 python tools/sell_time.py $wl iters=150 - >> gpurun_out/r4/decomp_$wl.log 2>&1 || exit 1
  done
done
python - <<PY
import json
v=None
for l in open("gpurun_out/r4/decomp_$wl.log"):
    if l.startswith("## "): v=l[3:].strip()
    if l.startswith("{"):
        d=json.loads(l); print(v, round(d["us_per_iteration"],1), "TB/s", round(d["moved_TBps"],2))
PY
