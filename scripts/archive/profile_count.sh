# rocprofv3 kernel stats of the super-k-mer split + count stages (scripts/bench_count.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_count
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_count.py > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
tail -1 $OUT/stats.log | cut -c1-1500
python - <<'PY'
import csv, os
rows = list(csv.reader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_count/kernel_stats.csv")))
for r in rows[1:16]:
    print(r[0][:90].ljust(90), r[1].rjust(5), "avg_us", round(float(r[3]) / 1e3, 1), "tot_ms", round(float(r[2]) / 1e6, 2))
PY
