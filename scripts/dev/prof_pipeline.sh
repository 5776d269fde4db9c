# kernel stats of `kmx pipeline` runs: $1 = output tag, rest = bench_pipeline.py arguments
export TMPDIR=/tmp; O=gpurun_out/$1; shift; mkdir -p $O
timeout 1500 python scripts/bench_pipeline.py --env "KMX_SLOW_EXIT=1" --prof $O "$@" > $O/lines.jsonl 2> $O/err.txt
python - $O <<'PY'
import csv, sys, glob, json
for l in open(sys.argv[1] + "/lines.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print({k: d[k] for k in ("samples", "genome", "count_wall_s", "count_s", "merge_wall_s", "total_s", "resident_count_calls")}, d["flags"][-50:])
for f in sorted(glob.glob(sys.argv[1] + "/kernel_stats_*.csv")):
    rows = list(csv.DictReader(open(f))); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "kernel total %.1f ms" % (tot / 1e6))
    for r in rows[:26]: print("  %-72s calls %6s avg %8.1f us tot %8.1f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
