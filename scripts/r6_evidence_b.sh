#!/bin/bash
# round-6 evidence, part B: rocprofv3 kernel stats of every merge workload, HBM traffic counters (separate --pmc passes, no trace domains),
# the pipeline end to end at full size, full-size parity.  -> gpurun_out/r6e/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
python scripts/bench_pipeline.py --samples 1000 --genome 5e6 --partitions 256 --tmp /dev/shm --variants ";" > $O/pipeline_1000x5Mbp_shm.jsonl 2>$O/err_pipe.log
python scripts/verify_bench_parity.py --workload count > $O/verify_count.json 2>$O/err_verify.log
python scripts/verify_bench_parity.py --workload pa63 > $O/verify_pa63.json 2>>$O/err_verify.log
cd /tmp && export TMPDIR=/tmp
prof() { L=$1; shift; rm -rf $O/prof_$L; mkdir -p $O/prof_$L
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$L --output-format csv -- "$@" > $O/prof_$L.log 2>&1
  find $O/prof_$L -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$L.csv \; ; grep '^{' $O/prof_$L.log | tail -1 > $O/line_under_rocprof_$L.json; rm -rf $O/prof_$L; }
prof count python $R/bench.py --workload count --no-cpu-baseline --no-whole-job
prof bf python $R/bench.py --workload bf --no-cpu-baseline
prof bft python $R/bench.py --workload bft --no-cpu-baseline
prof pa63 python $R/bench.py --workload pa63 --no-cpu-baseline
cd $R
bash scripts/r5_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/r5pmc/traffic.txt $O/pmc_traffic_count.txt
WL=pa63 bash scripts/r5_pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/r5pmc/traffic.txt $O/pmc_traffic_pa63.txt
rm -rf gpurun_out/r5pmc
df -h /tmp /dev/shm > $O/box.txt; nproc >> $O/box.txt; free -g >> $O/box.txt
ls -la $O; cat $O/pmc_traffic_count.txt $O/pmc_traffic_pa63.txt; cut -c1-600 $O/pipeline_1000x5Mbp_shm.jsonl; head -c 600 $O/verify_count.json; head -5 $O/kernel_stats_count.csv | cut -c1-250
