# one workload's kernel time under several tuning builds of libkmx (scripts/dev/build_variant.sh): bench_variants.sh <dir> <workload> <name>...
d=$1; wl=$2; shift; shift
mkdir -p gpurun_out/$d
for v in "$@"; do
  lib=""; [ "$v" != "product" ] && lib=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx_$v.so
  KMX_LIB=$lib python bench.py --workload $wl --no-cpu-baseline --steps 10 $EXTRA > gpurun_out/$d/${wl}_$v.json 2> gpurun_out/$d/${wl}_$v.err
  python - gpurun_out/$d/${wl}_$v.json $v <<PY
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(sys.argv[2], "ms_step %.3f kernel_ms %.3f frac %.3f" % (d["ms_per_step"], r["kernel_ms"], r["frac"]), "fo", (r.get("file_order") or {}).get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
