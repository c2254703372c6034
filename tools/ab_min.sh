# Cross-process A/B at Queen size is disturbed by the two speeds a process can get (r04_sweeps.md D): N processes per library,
# the fastest and all of them printed.   tools/ab_min.sh <workload> <reps> lib1 lib2 ...   (names under build_ab/)
wl=$1; reps=$2; shift 2
mkdir -p gpurun_out/r4
log=gpurun_out/r4/abmin_$wl.log; : > $log
for rep in $(seq $reps); do
  for v in "$@"; do
    echo "## $v" >> $log
    PRCG_LIB=$PWD/build_ab/libprcg_$v.so timeout -k 10 200 python tools/sell_time.py $wl iters=120 - >> $log 2>&1 || exit 1
  done
done
python - <<PY
import json, collections
v=None; r=collections.OrderedDict()
for l in open("$log"):
    if l.startswith("## "): v=l[3:].strip()
    if l.startswith("{"):
        r.setdefault(v, []).append(round(json.loads(l)["us_per_iteration"],1))
for k, t in r.items(): print(k, "min", min(t), "all", t)
PY
