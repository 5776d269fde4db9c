#!/bin/bash
# round 6: A/B of count-stage settings (environment) on one box: kernel times of one 30-Mbase sample (rocprofv3 --kernel-trace --stats), interleaved
# SETS="NAME=VALUE;NAME=VALUE ..." (the empty set first = the product's defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6ceab; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra SETS_A <<< ";${SETS}"
for rep in 1 2; do i=0; for S in "${SETS_A[@]}"; do i=$((i+1))
  env $S timeout 600 rocprofv3 --kernel-trace --stats -d $O/t_${i}_$rep --output-format csv -- python $R/scripts/bench_count_stage.py --genome 5e6 --partitions 256 --reps 20 --skip-streams ${EXTRA:-} > $O/b_${i}_$rep.json 2> $O/log_${i}_$rep.txt
  python - <<PY
import csv, glob, json
f = glob.glob("$O/t_${i}_$rep/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 23 / 1e3
pick = {}
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "").replace("kmx::", "")[:24]
    pick[n] = round(float(r["TotalDurationNs"]) / 23 / 1e3, 1)
try: call = json.loads(open("$O/b_${i}_$rep.json").read().strip().splitlines()[-1])["count_reads_dev_ms_median"]
except Exception: call = None
print("[${S:-defaults}] rep $rep: kernels us", round(tot, 1), "call ms", call, {k: v for k, v in sorted(pick.items(), key=lambda kv: -kv[1])[:6]})
PY
  rm -rf $O/t_${i}_$rep
done; done
