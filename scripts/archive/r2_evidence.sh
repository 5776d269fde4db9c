# round-2 evidence: bench lines of every workload, rocprofv3 kernel stats, HBM traffic counters (separate --pmc passes, no trace
# domains), the kernel grid and the count-stage timing.  Everything lands under gpurun_out/r2/ (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; rm -rf $O; mkdir -p $O
cd $R
line() { grep '^{"metric' | tail -1; }
python bench.py 2>$O/err_default.log | line > $O/bench_count_counted.json
python bench.py --lists random --no-cpu-baseline 2>/dev/null | line > $O/bench_count_random.json
python bench.py --rec-min 1 --partitions-per-gpu 8 --no-cpu-baseline 2>/dev/null | line > $O/bench_count_recmin1.json
python bench.py --workload bf 2>/dev/null | line > $O/bench_bf.json
python bench.py --workload pa63 --no-cpu-baseline 2>/dev/null | line > $O/bench_pa63.json
python bench.py --workload pa63 --lists random --partitions-per-gpu 16 --no-cpu-baseline 2>/dev/null | line > $O/bench_pa63_random16.json
python bench.py --lists random --rec-min 1 --partitions-per-gpu 8 --no-cpu-baseline 2>/dev/null | line > $O/bench_count_random_recmin1.json
python bench.py --workload bft --no-cpu-baseline 2>/dev/null | line > $O/bench_bft.json
for K in pivot rows; do KMX_MERGE_KERNEL=$K python bench.py --no-cpu-baseline 2>/dev/null | line > $O/bench_count_counted_$K.json; done
python scripts/bench_count.py > $O/bench_count_stage.json 2>$O/err_count.log
cd /tmp && export TMPDIR=/tmp
prof() { L=$1; shift; rm -rf $O/prof_$L; mkdir -p $O/prof_$L
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$L --output-format csv -- "$@" > $O/prof_$L.log 2>&1
  find $O/prof_$L -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$L.csv \; ; grep '^{"metric' $O/prof_$L.log | tail -1 > $O/bench_under_rocprof_$L.json; rm -rf $O/prof_$L; }
prof count_counted python $R/bench.py --no-cpu-baseline
prof bf python $R/bench.py --workload bf --no-cpu-baseline
prof bft python $R/bench.py --workload bft --no-cpu-baseline
prof pa63 python $R/bench.py --workload pa63 --no-cpu-baseline
prof count_stage python $R/scripts/bench_count.py
pmc() { L=$1; C=$2; shift; shift; rm -rf $O/pmc_$L_$C; mkdir -p $O/pmc_${L}_$C
  timeout 900 rocprofv3 --pmc $C -d $O/pmc_${L}_$C --output-format csv -- "$@" > $O/pmc_${L}_$C.log 2>&1; }
for C in FETCH_SIZE WRITE_SIZE; do
  pmc count $C python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline
  pmc bf $C python $R/bench.py --workload bf --steps 2 --warmup 1 --no-cpu-baseline
  pmc cstage $C python $R/scripts/bench_count.py
done
python - <<'PY' | tee $O/pmc_traffic.txt
import glob, csv, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2"
for L in ("count", "bf", "cstage"):
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
        print(f"{L:7s} {k[0]:28s} {k[1]:11s} mean={sum(v)/len(v):.8g} max={max(v):.8g} n={len(v)}")
PY
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
cd $R
( for cfg in "1000 0.001" "1000 0.002" "1000 0.003" "1000 0.005" "500 0.001" "200 0.001" "128 0.001"; do set -- $cfg; for K in rows pivot cols; do
  echo -n "N=$1 d=$2 $K: "
  KMX_TRACE=1 KMX_MERGE_KERNEL=$K timeout 300 python bench.py --lists random --samples $1 --subst-rate $2 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
back=0
for l in sys.stdin:
    if 'handed back' in l: back+=1
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s  hand-backs', back)"
done; done ) > $O/kernel_grid.txt 2>&1
ls $O
