#!/bin/bash
# Where does the column-blocked pair overtake k_merge_rows when the rows leave in file order (the library's default)?
# r04_kernel_grid.txt (random lists, 32 partitions per launch) says between 200 and 500 lists; the selection in kmx_api.hip
# (192 lists) was set before file order existed.  Both kinds of lists, 32 partitions per launch, forced kernels.
O=${1:-gpurun_out/r4x}; mkdir -p $O
for L in counted random; do for N in 192 256 320 400; do for K in rows cols; do
  echo -n "lists=$L N=$N $K: "
  KMX_MERGE_KERNEL=$K timeout 200 python bench.py --workload count --lists $L --samples $N --steps 6 --warmup 2 --no-cpu-baseline --no-whole-job 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r['file_order']; print(r['kernel'], 'unordered', round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step | file order', round(f['kernel_ms'],3), 'ms kernel', round(f['ms_per_step'],3), 'ms/step frac', round(r['frac_with_file_order'],3))"
  echo
done; done; done > $O/crossover.txt 2>&1
cat $O/crossover.txt
