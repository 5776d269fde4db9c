#!/bin/bash
# round 6: cohorts of 64-256 lists (configs[2]'s lists, rows in file order): k_merge_rows builds with smaller tiles / more workgroups a CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6mid; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/mid.txt
for N in ${NS:-64 128 200 256}; do for V in ${VARIANTS:-product rows512 rows512b rows256}; do
  L=""; [ "$V" != "product" ] && L="$R/kmtricks_amd/libkmx_$V.so"
  KMX_LIB=$L KMX_MERGE_KERNEL=${KERNEL:-rows} python $R/bench.py --workload count --samples $N --steps 10 --warmup 2 --no-cpu-baseline --no-whole-job 2>$O/err_${N}_$V.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('N=$N $V kernel', r['kernel'], 'kernel_ms', round(r['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'algo GB', round(r['algo_bytes_per_launch']/1e9,3))" >> $O/mid.txt
done; done
cat $O/mid.txt
