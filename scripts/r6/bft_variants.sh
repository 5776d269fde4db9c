#!/bin/bash
# round 6: k_merge_bft builds (scripts/dev/build_variant.sh bft_<name> "-D...", FILES=merge_bft) on configs[3], one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6bft; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/bft.txt
for V in product $(ls $R/kmtricks_amd/libkmx_bft_*.so 2>/dev/null | sed 's/.*libkmx_//; s/\.so//') product; do
  L=""; [ "$V" != "product" ] && L="$R/kmtricks_amd/libkmx_$V.so"
  KMX_LIB=$L python $R/bench.py --workload ${WL:-bft} --steps 10 --warmup 3 --no-cpu-baseline 2>$O/err_$V.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('$V kernel', r['kernel'], 'kernel_ms', round(r['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3))
except Exception as e: print('$V failed', e)" >> $O/bft.txt
done
cat $O/bft.txt
