# count and pa63 bench lines: the pair with the rows where the kernels leave them, and with the rows in file order (bench.py times both)
mkdir -p gpurun_out/$1
for wl in ${WLS:-count pa63}; do python bench.py --workload $wl --no-cpu-baseline --no-whole-job --steps 10 $EXTRA > gpurun_out/$1/${wl}.json 2> gpurun_out/$1/${wl}.err; done
python - $1 <<PY
import json, sys, os
for w in os.environ.get("WLS", "count pa63").split():
    try:
      d=json.load(open(f"gpurun_out/{sys.argv[1]}/{w}.json")); r=d["roofline"]; fo=r.get("file_order") or {}
      print(w, "arena: step %.3f kernel %.3f frac %.3f | file order: step %.3f kernel %.3f frac %.3f | gather %.2f" % (d["ms_per_step"], r["kernel_ms"], r["frac"], fo.get("ms_per_step", 0), fo.get("kernel_ms", 0), r.get("frac_with_file_order") or 0, r["file_order_gather_ms"]))
    except Exception as e: print(w, "failed", e, open(f"gpurun_out/{sys.argv[1]}/{w}.err").read()[-600:])
PY
