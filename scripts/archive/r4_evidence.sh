# round-4 evidence: the driver's bench command, rocprofv3 kernel stats of every merge workload (rows in file order and where the kernels
# leave them), HBM traffic counters (separate --pmc passes, no trace domains), the pipeline end to end, full-size parity, the GPU tests.
# Everything lands under gpurun_out/r4/ (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; rm -rf $O; mkdir -p $O
cd $R
line() { grep '^{"metric' | tail -1; }
( time python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/err_all.log | line > $O/bench_all.json ) 2> $O/time_all.txt
python bench.py --workload count --share-min 1 --steps 20 --warmup 5 --no-cpu-baseline --no-whole-job 2>/dev/null | line > $O/bench_count_sharemin1.json
python bench.py --workload count --rec-min 1 --partitions-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline --no-whole-job 2>/dev/null | line > $O/bench_count_recmin1.json
python scripts/bench_pipeline.py --samples 1000 --genome 1e6 --partitions 256 --sync --variants ";;--skip-partiinfo;--samples-per-call 4" > $O/pipeline_1000x1Mbp.jsonl 2>$O/err_pipe.log
KMX_FILE_ORDER=0 python scripts/bench_pipeline.py --samples 1000 --genome 1e6 --partitions 256 --sync > $O/pipeline_1000x1Mbp_gather.jsonl 2>>$O/err_pipe.log
# (the first run of a box pays its page faults in /dev/shm: three runs of the default, then the count stage as it was before this round's
#  last changes -- statistics by atomics, no uploads ahead)
python scripts/bench_pipeline.py --samples 1000 --genome 5e6 --partitions 256 --tmp /dev/shm --variants ";;;" --env "A=1;A=1;A=1;KMX_STATS_ATOMICS=1 KMX_READS_AHEAD=0" > $O/pipeline_1000x5Mbp_count_shm.jsonl 2>>$O/err_pipe.log
KMX_SLOW_EXIT=1 bash scripts/dev/prof_pipeline.sh r4/prof_count_stage --samples 200 --genome 5e6 --partitions 256 --tmp /dev/shm > $O/count_stage_5Mbp_kernels.txt 2>&1; cp $O/prof_count_stage/kernel_stats_0.csv $O/kernel_stats_count_stage_5Mbp.csv; rm -rf $O/prof_count_stage
python scripts/bench_pipeline.py --samples 500 --genome 5e6 --partitions 256 --kmer-size 63 --mode kmer:pa:bin --extra "--recurrence-min 1" --tmp /dev/shm > $O/pipeline_500x5Mbp_k63_pa.jsonl 2>>$O/err_pipe.log
df -h /tmp /dev/shm > $O/box.txt; nproc >> $O/box.txt; free -g >> $O/box.txt
python scripts/verify_bench_parity.py --workload count > $O/verify_count.json 2>$O/err_verify.log
python scripts/verify_bench_parity.py --workload pa63 > $O/verify_pa63.json 2>>$O/err_verify.log
cd /tmp && export TMPDIR=/tmp
prof() { L=$1; shift; rm -rf $O/prof_$L; mkdir -p $O/prof_$L
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$L --output-format csv -- "$@" > $O/prof_$L.log 2>&1
  find $O/prof_$L -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$L.csv \; ; grep '^{' $O/prof_$L.log | tail -1 > $O/line_under_rocprof_$L.json; rm -rf $O/prof_$L; }
prof count_counted python $R/bench.py --workload count --no-cpu-baseline --no-whole-job
prof bf python $R/bench.py --workload bf --no-cpu-baseline
prof bft python $R/bench.py --workload bft --no-cpu-baseline
prof pa63 python $R/bench.py --workload pa63 --no-cpu-baseline
pmc() { L=$1; C=$2; shift; shift; rm -rf $O/pmc_${L}_$C; mkdir -p $O/pmc_${L}_$C
  timeout 900 rocprofv3 --pmc $C -d $O/pmc_${L}_$C --output-format csv -- "$@" > $O/pmc_${L}_$C.log 2>&1; }
for C in FETCH_SIZE WRITE_SIZE; do
  pmc count $C python $R/bench.py --workload count --steps 2 --warmup 2 --no-cpu-baseline --no-whole-job
  pmc bft $C python $R/bench.py --workload bft --steps 2 --warmup 1 --no-cpu-baseline
  pmc pa63 $C python $R/bench.py --workload pa63 --steps 2 --warmup 1 --no-cpu-baseline
done
python - <<'PY' | tee $O/pmc_traffic.txt
import glob, csv, collections, os, re
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4"
for L in ("count", "bft", "pa63"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{L}_*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "kmx::" in n:
                m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", n)
                per[(r["Dispatch_Id"], m.group(0) if m else n[:40], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, k, c), v in per.items():
            acc[(k, c)].append(v)
    for k in sorted(acc):
        v = acc[k]
        print(f"{L:7s} {k[0]:44s} {k[1]:11s} mean={sum(v)/len(v):.8g} max={max(v):.8g} n={len(v)}")
PY
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
cd $R
( for cfg in "1000 0.001" "500 0.001" "200 0.001" "128 0.001"; do set -- $cfg; for K in rows cols; do
  echo -n "N=$1 d=$2 $K: "
  KMX_MERGE_KERNEL=$K timeout 300 python bench.py --workload count --lists random --samples $1 --subst-rate $2 --steps 4 --warmup 2 --no-cpu-baseline --no-whole-job 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s frac', round(r['frac'],3), '| file order', round(r['file_order']['kernel_ms'],3), 'ms frac', round(r['frac_with_file_order'],3))"
done; done ) > $O/kernel_grid.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/gpu_tests.txt
ls $O
