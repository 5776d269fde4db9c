# round 2: rocprofv3 kernel stats of a bench workload ($1 = label, rest = bench.py arguments)
L=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$L
rm -rf $OUT; mkdir -p $OUT
timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
tail -1 $OUT/stats.log > $OUT/bench_line_under_rocprof.json
head -12 $OUT/kernel_stats.csv | cut -c1-200
cut -c1-400 $OUT/bench_line_under_rocprof.json
find $OUT -name "*.csv" -size +20M -delete
