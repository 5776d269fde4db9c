#!/bin/bash
# round 6: host phases of the count calls inside `kmx pipeline` (KMX_COUNT_PHASES=1), 1000 x 5 Mbp
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6phases; mkdir -p $O
python $GRAFT_REPO_ROOT/scripts/bench_pipeline.py --samples ${SAMPLES:-400} --genome 5e6 --partitions 256 --tmp /dev/shm --extra "--hard-min 2 --recurrence-min 2 --static-repart ${EXTRA:-}" \
  --variants ";" --env "KMX_COUNT_PHASES=1 ${ENVX:-};KMX_COUNT_PHASES=1 ${ENVX:-}" --trace > $O/lines.jsonl 2> $O/err.log
grep -h "count phases" $GRAFT_REPO_ROOT/gpurun_out/pipeline_trace_1.txt | tail -2
grep -h "count_wall_s" $O/lines.jsonl | tail -1 | cut -c200-700
