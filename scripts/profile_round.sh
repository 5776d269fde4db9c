# end-of-round evidence for the merge bench: rocprofv3 kernel stats + HBM traffic counters (separate passes)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
tail -1 $OUT/stats.log > $OUT/bench_line_under_rocprof.json
i=0
for G in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $G -d $OUT/p$i --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - <<'PY'
import glob, csv, collections, os
acc = collections.defaultdict(list)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_merge" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Kernel_Name"].split("<")[0].split("(")[0][-16:], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[(k, c)].append(v)
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:18s} {k[1]:16s} mean={sum(v)/len(v):.6g}  n={len(v)}")
PY
head -5 $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +20M -delete
