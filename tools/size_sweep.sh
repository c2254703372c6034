mkdir -p gpurun_out/r4
for m in 90 100 111 124; do
  for rep in 1 2; do
    timeout -k 10 200 python tools/sell_time.py fem:$m iters=150 - - >> gpurun_out/r4/sizes.log 2>&1 || exit 1
  done
done
python - <<PY
import json
for l in open("gpurun_out/r4/sizes.log"):
    if l.startswith("{"):
        d=json.loads(l); nnz=d["operator_bytes"]/d["bytes_per_nnz"]; print(d["workload"], round(d["us_per_iteration"],1), "ps/nnz", round(d["us_per_iteration"]*1e6/nnz,3), "TB/s", round(d["moved_TBps"],2))
PY
