# round-3 evidence: the driver's bench command, bench lines and rocprofv3 kernel stats of every workload, the count stage, the pipeline
# end to end (variants), HBM traffic counters (separate --pmc passes, no trace domains), the kernel grid, full-size parity.
# Everything lands under gpurun_out/r3/ (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; rm -rf $O; mkdir -p $O
cd $R
line() { grep '^{"metric' | tail -1; }
( time python bench.py --steps 20 --warmup 5 2>$O/err_all.log | line > $O/bench_all.json ) 2> $O/time_all.txt
python bench.py --workload count --share-min 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $O/bench_count_sharemin1.json
python bench.py --workload count --rec-min 1 --partitions-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $O/bench_count_recmin1.json
python bench.py --workload count --lists random --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $O/bench_count_random.json
for K in pivot rows; do KMX_MERGE_KERNEL=$K python bench.py --workload count --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line > $O/bench_count_counted_$K.json; done
python scripts/bench_count_stage.py > $O/bench_count_stage_5Mbp_32.json 2>$O/err_cs.log
python scripts/bench_count_stage.py --genome 1e6 --partitions 256 > $O/bench_count_stage_1Mbp_256.json 2>>$O/err_cs.log
python scripts/bench_count_stage.py --hash > $O/bench_count_stage_hash.json 2>>$O/err_cs.log
python scripts/bench_count_stage.py --kmer-size 63 > $O/bench_count_stage_k63.json 2>>$O/err_cs.log
for B in hash sort; do KMX_COUNT_BUCKETS=$B python scripts/bench_count_stage.py > $O/bench_count_stage_5Mbp_32_buckets_$B.json 2>>$O/err_cs.log; done
KMX_COUNT_BUCKETS=sort python scripts/bench_count_stage.py --kmer-size 63 > $O/bench_count_stage_k63_buckets_sort.json 2>>$O/err_cs.log
python scripts/bench_pipeline.py --samples 1000 --genome 1e6 --partitions 256 --sync --variants ";;--skip-partiinfo;--no-resident;--gpu-workers 1;--gpu-workers 3" > $O/pipeline_1000x1Mbp.jsonl 2>$O/err_pipe.log
KMX_RING_PREFILL=1 python scripts/bench_pipeline.py --samples 1000 --genome 1e6 --partitions 256 --sync > $O/pipeline_1000x1Mbp_ring_prefill.jsonl 2>>$O/err_pipe.log
KMX_OUT_ORDER=1 python scripts/bench_pipeline.py --samples 1000 --genome 1e6 --partitions 256 --sync > $O/pipeline_1000x1Mbp_out_order.jsonl 2>>$O/err_pipe.log
python scripts/bench_pipeline.py --samples 100 --genome 5e6 --partitions 32 --mode hash:bf:bin --bloom 1e8 --sync > $O/pipeline_100x5Mbp_bf.jsonl 2>>$O/err_pipe.log
python scripts/bench_pipeline.py --samples 200 --genome 5e6 --partitions 256 --sync > $O/pipeline_200x5Mbp_count.jsonl 2>>$O/err_pipe.log
python scripts/bench_pipeline.py --samples 1000 --genome 5e6 --partitions 256 --mode kmer:pa:bin --sync > $O/pipeline_1000x5Mbp_pa.jsonl 2>>$O/err_pipe.log
df -h /tmp > $O/df.txt; nproc >> $O/df.txt; free -g >> $O/df.txt
python scripts/verify_bench_parity.py --workload count > $O/verify_count.json 2>$O/err_verify.log
python scripts/verify_bench_parity.py --workload pa63 > $O/verify_pa63.json 2>>$O/err_verify.log
cd /tmp && export TMPDIR=/tmp
prof() { L=$1; shift; rm -rf $O/prof_$L; mkdir -p $O/prof_$L
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$L --output-format csv -- "$@" > $O/prof_$L.log 2>&1
  find $O/prof_$L -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$L.csv \; ; grep '^{' $O/prof_$L.log | tail -1 > $O/line_under_rocprof_$L.json; rm -rf $O/prof_$L; }
prof count_counted python $R/bench.py --workload count --no-cpu-baseline
prof count_sharemin1 python $R/bench.py --workload count --share-min 1 --no-cpu-baseline
prof bf python $R/bench.py --workload bf --no-cpu-baseline
prof bft python $R/bench.py --workload bft --no-cpu-baseline
prof pa63 python $R/bench.py --workload pa63 --no-cpu-baseline
prof count_stage python $R/scripts/bench_count_stage.py
prof count_stage_1Mbp_256 python $R/scripts/bench_count_stage.py --genome 1e6 --partitions 256
prof count_stage_k63 python $R/scripts/bench_count_stage.py --kmer-size 63
prof pipeline_200x1Mbp python $R/scripts/bench_pipeline.py --samples 200 --genome 1e6 --partitions 256
pmc() { L=$1; C=$2; shift; shift; rm -rf $O/pmc_${L}_$C; mkdir -p $O/pmc_${L}_$C
  timeout 900 rocprofv3 --pmc $C -d $O/pmc_${L}_$C --output-format csv -- "$@" > $O/pmc_${L}_$C.log 2>&1; }
for C in FETCH_SIZE WRITE_SIZE; do
  pmc count $C python $R/bench.py --workload count --steps 2 --warmup 2 --no-cpu-baseline
  pmc bft $C python $R/bench.py --workload bft --steps 2 --warmup 1 --no-cpu-baseline
  pmc pa63 $C python $R/bench.py --workload pa63 --steps 2 --warmup 1 --no-cpu-baseline
  pmc cstage $C python $R/scripts/bench_count_stage.py --reps 3
done
python - <<'PY' | tee $O/pmc_traffic.txt
import glob, csv, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3"
for L in ("count", "bft", "pa63", "cstage"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{L}_*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "kmx::" in n or n.startswith("k_") or "k_pack" in n:
                per[(r["Dispatch_Id"], n.split("(")[0].split("::")[-1], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, k, c), v in per.items():
            acc[(k, c)].append(v)
    for k in sorted(acc):
        v = acc[k]
        print(f"{L:7s} {k[0]:40s} {k[1]:11s} mean={sum(v)/len(v):.8g} max={max(v):.8g} n={len(v)}")
PY
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
cd $R
( for cfg in "1000 0.001" "1000 0.003" "500 0.001" "200 0.001" "128 0.001"; do set -- $cfg; for K in rows pivot cols; do
  echo -n "N=$1 d=$2 $K: "
  KMX_TRACE=1 KMX_MERGE_KERNEL=$K timeout 300 python bench.py --workload count --lists random --samples $1 --subst-rate $2 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
back=0
for l in sys.stdin:
    if 'handed back' in l: back+=1
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s frac', round(r['frac'],3), ' hand-backs', back)"
done; done ) > $O/kernel_grid.txt 2>&1
( for PP in 32 128 256; do echo -n "N=128 partitions per launch $PP: "; timeout 300 python bench.py --workload count --lists random --samples 128 --partitions-per-gpu $PP --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s frac', round(r['frac'],3))"
done ) > $O/small_cohort_batches.txt 2>&1
bash $R/scripts/r3_pipeline_api_profile.sh > $O/pipeline_api_profile.txt 2>&1; cp $R/gpurun_out/r3_api/hip_api_stats.csv $O/pipeline_hip_api_stats.csv; cp $R/gpurun_out/r3_api/kernel_stats.csv $O/pipeline_300x1Mbp_kernel_stats.csv
ls $O
