# end-of-round evidence for the merge bench: rocprofv3 kernel stats (HBM traffic counters: scripts/pmc_traffic.sh)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
tail -1 $OUT/stats.log > $OUT/bench_line_under_rocprof.json
head -5 $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +20M -delete
