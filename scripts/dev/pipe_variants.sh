# $1 = tag; SAMPLES GENOME; VARIANTS (';'-separated flags) ENVS (';'-separated env sets)
O=gpurun_out/$1; mkdir -p $O
timeout ${TMO:-1500} python scripts/bench_pipeline.py --samples ${SAMPLES:-1000} --genome ${GENOME:-1e6} --partitions 256 ${TMPD:+--tmp $TMPD} --variants "${VARIANTS:-}" --env "${ENVS:-}" > $O/lines.jsonl 2> $O/err.txt
python - $O <<'PY'
import sys, json
for l in open(sys.argv[1] + "/lines.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print("count_wall %.3f count_s %.3f merge_wall %.3f total %.3f calls %d | %s" % (d["count_wall_s"], d["count_s"], d["merge_wall_s"], d["total_s"], d["resident_count_calls"], d["flags"][-70:]))
PY
tail -2 $O/err.txt
