#!/bin/bash
# round 5: what the kernels of a small cohort (128 / 200 lists, configs[2]'s lists, rows in file order) cost one by one, both merges forced
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for N in ${NS:-128 200}; do for K in rows cols; do
  rm -rf $O/p; mkdir -p $O/p
  KMX_MERGE_KERNEL=$K timeout 600 rocprofv3 --kernel-trace --stats -d $O/p --output-format csv -- python $R/bench.py --workload count --samples $N --steps 10 --warmup 2 --no-cpu-baseline --no-whole-job > $O/log_${N}_$K.txt 2>&1
  echo "== N=$N $K" >> $O/small.txt
  grep '^{' $O/log_${N}_$K.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('rows_left_in_arena',{})
print('line: kernel', r['kernel'], 'kernel_ms', round(r['kernel_ms'],3), 'ms/step', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'algo GB', round(r['algo_bytes_per_launch']/1e9,3), 'records', d['config'].get('records_per_step_per_gpu'), 'arena', f.get('kernel_ms'))" >> $O/small.txt
  find $O/p -name "*kernel_stats.csv" | head -1 | xargs -r python -c "
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'merge' in r['Name'] or 'cols' in r['Name'] or 'range' in r['Name'] or 'rows' in r['Name'] or 'order' in r['Name']]
for r in rows[:12]: print('  %-60s calls %5s avg %9.1f us' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3))" >> $O/small.txt
  rm -rf $O/p
done; done
cat $O/small.txt
