#!/bin/bash
# round 5: HBM-side traffic of the merge kernels on the headline workload (rows in file order, then left in the arena): FETCH_SIZE and
# WRITE_SIZE in passes of their own (they do not fit one pass together on gfx950), no trace domains -> gpurun_out/r5pmc/traffic.txt
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5pmc; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $O/$C --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r5_grid.py --wl ${WL:-count} --steps 3 --settings "x=1" > $O/$C.log 2>&1
done
python - <<'PY' | tee $O/traffic.txt
import glob, csv, collections, os
acc = collections.defaultdict(list)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5pmc/*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_merge_cols" in kn or "k_cols_sparse" in kn:
            per[(r["Dispatch_Id"], kn.split("(")[0].split("::")[-1], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[(k, c)].append(v)
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:44s} {k[1]:12s} mean_KiB={sum(v)/len(v):.8g}  n={len(v)}")
PY
find $O -name "*.csv" -size +5M -delete
