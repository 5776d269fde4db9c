#!/bin/bash
# round 5: the GPU test suite, then every workload of bench.py (the line the driver records), then the count stage's kernel profile
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5r${TAG:-}; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt; fi
if [ -z "$SKIP_BENCH" ]; then timeout 1500 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -3 $O/bench.err; fi
if [ -n "$COUNT_PROFILE" ]; then bash scripts/r5_count_stage_profile.sh > $O/count_profile.log 2>&1; tail -5 $O/count_profile.log; fi
