# count and pa63 bench lines with the rows in file order (1) and not (0): one summary line each
mkdir -p gpurun_out/$1
for fo in ${FOS:-1 0}; do for wl in ${WLS:-count pa63}; do KMX_FILE_ORDER=$fo python bench.py --workload $wl --no-cpu-baseline --steps 10 $EXTRA > gpurun_out/$1/${wl}_fo$fo.json 2> gpurun_out/$1/${wl}_fo$fo.err; done; done
python - $1 <<PY
import json, sys, os
for w in os.environ.get("WLS", "count pa63").split():
  for fo in os.environ.get("FOS", "1 0").split():
    try:
      d=json.load(open(f"gpurun_out/{sys.argv[1]}/{w}_fo{fo}.json")); r=d["roofline"]
      print(w,fo,"ms_step",round(d["ms_per_step"],3),"kernel_ms",round(r["kernel_ms"],3),"frac",round(r["frac"],3),"gather",round(r["file_order_gather_ms"],2),"order",round(r["row_order_ms"],2), "rows", d["config"]["rows_out_per_step_per_gpu"], r["kernel"])
    except Exception as e: print(w, fo, "failed", e, open(f"gpurun_out/{sys.argv[1]}/{w}_fo{fo}.err").read()[-600:])
PY
