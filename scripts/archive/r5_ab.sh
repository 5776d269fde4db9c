#!/bin/bash
# round 5: (1) the r03 build of k_merge_cols against HEAD's on ONE box (rocprofv3 kernel trace of the same bench command);
# (2) what the pair's kernels lose on a part of the chip (scripts/r5_grid.py); (3) the merge parity tests on the build
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5a
mkdir -p $O
cd $GRAFT_REPO_ROOT/_r03 && (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/r03 -o r03 -- python $GRAFT_REPO_ROOT/_r03/bench.py --workload count --steps 10 --warmup 3 --no-cpu-baseline) > $O/r03_bench.json 2> $O/r03_bench.err
cd $GRAFT_REPO_ROOT && (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/head -o head -- python $GRAFT_REPO_ROOT/bench.py --workload count --steps 10 --warmup 3 --no-cpu-baseline --no-whole-job) > $O/head_bench.json 2> $O/head_bench.err
cd $GRAFT_REPO_ROOT && python scripts/r5_grid.py > $O/grid.jsonl 2> $O/grid.err
cd $GRAFT_REPO_ROOT && timeout 900 python -m pytest tests/test_merge_gpu.py -m gpu -x -q > $O/test_merge.txt 2>&1
find $O -name "*.csv" | head -20
for f in $O/r03/*kernel_stats.csv $O/head/*kernel_stats.csv; do echo $f; head -8 $f; done
cat $O/grid.jsonl; tail -3 $O/test_merge.txt; tail -3 $O/grid.err
# keep only the stats (the traces are large)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
