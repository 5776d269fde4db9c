# round 2: HBM traffic counters of the bft and pa63 workloads (separate --pmc passes, no trace domains) -> gpurun_out/r2c/pmc_traffic_extra.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { L=$1; C=$2; shift; shift; mkdir -p $O/pmc_${L}_$C
  timeout 900 rocprofv3 --pmc $C -d $O/pmc_${L}_$C --output-format csv -- "$@" > $O/pmc_${L}_$C.log 2>&1; }
for C in FETCH_SIZE WRITE_SIZE; do
  pmc bft $C python $R/bench.py --workload bft --steps 2 --warmup 1 --no-cpu-baseline
  pmc pa63 $C python $R/bench.py --workload pa63 --steps 2 --warmup 1 --no-cpu-baseline
done
python - <<'PY' | tee $O/pmc_traffic_extra.txt
import glob, csv, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2c"
for L in ("bft", "pa63"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{L}_*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "kmx::" in n or n.startswith("k_"):
                per[(r["Dispatch_Id"], n.split("(")[0].split("::")[-1], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, k, c), v in per.items():
            acc[(k, c)].append(v)
    for k in sorted(acc):
        v = acc[k]
        if max(v) > 1000: print(f"{L:7s} {k[0]:28s} {k[1]:11s} mean={sum(v)/len(v):.8g} max={max(v):.8g} n={len(v)}")
PY
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
