# stalls in the count calls (KMX_TRACE stage clocks) for several environments: ENVS = ';'-separated sets
O=gpurun_out/stall; mkdir -p $O
N=$(echo "$ENVS" | awk -F';' '{print NF}'); V=$(printf ';%.0s' $(seq 2 $N))
timeout 1200 python scripts/bench_pipeline.py --samples ${SAMPLES:-200} --genome 5e6 --partitions 256 --tmp /dev/shm --trace --variants "$V" --env "$ENVS" > $O/lines.jsonl 2> $O/err.txt
python - $O <<'PY'
import re, sys, json, glob
lines = [json.loads(l) for l in open(sys.argv[1] + "/lines.jsonl") if l.startswith("{")]
for i, d in enumerate(lines):
    vals = []
    for l in open("gpurun_out/pipeline_trace_%d.txt" % i):
        if l.startswith('[kmx superk_partition]'): vals.append(sum(float(x) for x in re.findall(r'=([\d.]+)ms', l)))
    st = [v for v in vals if v > 10]
    print("count_wall %.3f total %.3f | calls %d sum %.0f ms, stalled %d sum %.0f ms, median %.2f | %s" % (d["count_wall_s"], d["total_s"], len(vals), sum(vals), len(st), sum(st), sorted(vals)[len(vals) // 2] if vals else 0, d["flags"][-60:]))
PY
