# rocprofv3 kernel stats of the count stage (scripts/bench_count_stage.py); arguments go to the script
cd /tmp && export TMPDIR=/tmp
TAG=${TAG:-count_stage}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
python $GRAFT_REPO_ROOT/scripts/bench_count_stage.py $* > $OUT/line.json 2>$OUT/err.log
cat $OUT/line.json
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_count_stage.py $* > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
python - <<PY
import csv
rows = list(csv.reader(open("$OUT/kernel_stats.csv")))
tot = 0
for r in rows[1:24]:
    print(r[0][:80].ljust(80), r[1].rjust(5), "avg_us", round(float(r[3]) / 1e3, 1), "tot_ms", round(float(r[2]) / 1e6, 2))
PY
