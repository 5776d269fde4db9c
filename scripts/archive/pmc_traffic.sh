# HBM traffic of the merge kernels on the bench workload: FETCH_SIZE and WRITE_SIZE in passes of their own
# (they do not fit one pass together on gfx950), no trace domains.  Writes gpurun_out/pmc_traffic.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $OUT/$C --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
done
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.txt
import glob, csv, collections, os
acc = collections.defaultdict(list)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_traffic/*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_merge" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[(k, c)].append(v)
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:18s} {k[1]:12s} mean={sum(v)/len(v):.8g}  n={len(v)}")
PY
find $OUT -name "*.csv" -size +20M -delete
