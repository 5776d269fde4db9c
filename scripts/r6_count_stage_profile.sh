#!/bin/bash
# round 6 (as round 5): the count stage's kernels for one 30-Mbase sample (scripts/bench_count_stage.py: configs[2]'s sample shape through
# kmx_count_reads_dev): rocprofv3 --kernel-trace --stats (times, calls) and --pmc passes of their own (instructions issued, wave
# cycles) -> gpurun_out/r6cs/{kernel_stats.csv, counters.txt}; scripts/r6_count_stage_table.py turns them into the table bench.py reads
set -x
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6cs${TAG:-}
rm -rf $O; mkdir -p $O
ARGS="--genome 5e6 --partitions ${PARTS:-256} --reps 20 ${EXTRA:-}"
python $GRAFT_REPO_ROOT/scripts/bench_count_stage.py $ARGS > $O/bench.json 2> $O/bench.log      # (unprofiled: the call's wall clock, the sample's numbers)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_count_stage.py $ARGS --skip-streams > $O/bench_profiled.json 2> $O/trace.log
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
P2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P -d $O/p$i --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_count_stage.py --genome 5e6 --partitions ${PARTS:-256} --reps 4 --skip-streams ${EXTRA:-} > $O/p$i.log 2>&1
done
python - <<'PY' > $O/counters.txt
import glob, csv, collections, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r6cs" + os.environ.get("TAG", "")
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        a = acc[(k, c)]; a[0] += v; a[1] += 1
for (k, c) in sorted(acc):
    s, n = acc[(k, c)]
    print(f"{k}\t{c}\t{s / n:.6g}\t{n}")
PY
find $O -name "*.csv" -size +3M -delete; rm -rf $O/trace $O/p1 $O/p2 $O/p3 $O/p4
head -30 $O/kernel_stats.csv; cat $O/bench.json
