#!/bin/bash
# round 6: A/B of k_merge_bft builds on one box, interleaved (the clocks of a box drift by 10 % over a minute of work: one pass says nothing)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6bft; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/ab.txt
for rep in 1 2 3; do for V in product ${VARIANTS}; do
  L=""; [ "$V" != "product" ] && L="$R/kmtricks_amd/libkmx_$V.so"
  KMX_LIB=$L python $R/bench.py --workload ${WL:-bft} --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline 2>$O/err_$V.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('rep $rep $V kernel_ms', round(r['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3))
except Exception as e: print('$V failed', e)" >> $O/ab.txt
done; done
cat $O/ab.txt
