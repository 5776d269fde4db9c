#!/bin/bash
# round 5: keys of three / four words through the partition-local sample sort -- tests, then the count call by itself (one 5 Mbp sample,
# 32 partitions) at k = 63 / 96 / 127, the wide ones also with the word-by-word library passes of round 4 (KMX_COUNT_SORT=library)
cd /root/repo; O=gpurun_out/r5w; mkdir -p $O
timeout 900 python -m pytest tests/test_count_gpu.py -m gpu -x -q 2>&1 | tail -4 > $O/tests_count.txt
timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "wide" 2>&1 | tail -4 > $O/tests_pipeline_wide.txt
for k in 63 96 127; do
  python scripts/bench_count_stage.py --kmer-size $k 2>/dev/null | tail -1 | sed "s/^/k=$k sample-sort: /" >> $O/count_call.txt
  if [ $k != 63 ]; then KMX_COUNT_SORT=library python scripts/bench_count_stage.py --kmer-size $k 2>/dev/null | tail -1 | sed "s/^/k=$k library: /" >> $O/count_call.txt; fi
done
cat $O/tests_count.txt $O/tests_pipeline_wide.txt; cut -c1-400 $O/count_call.txt
