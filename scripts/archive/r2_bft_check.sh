python -m pytest tests/test_merge_gpu.py -m gpu -x -q -k "bft" 2>&1 | tail -5
python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "bft" 2>&1 | tail -3
python bench.py --workload bft --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
KMX_TRACE=1 python bench.py --no-cpu-baseline --steps 3 2>gpurun_out/trace_counted.log | tail -1 | cut -c1-600
grep -v "superk_partition\|count_batch" gpurun_out/trace_counted.log | tail -30
