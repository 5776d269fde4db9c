# small cohorts: the merge step at 32 / 128 / 256 partitions per launch (rows / cols / libkmx's own choice)
for cfg in ${CFGS:-"128 32" "128 128" "128 256" "200 128"}; do set -- $cfg; for K in ${KS:-rows cols auto}; do
  echo -n "N=$1 partitions=$2 $K: "
  if [ $K = auto ]; then unset KMX_MERGE_KERNEL; else export KMX_MERGE_KERNEL=$K; fi
  timeout 300 python bench.py --workload count --lists random --samples $1 --partitions-per-gpu $2 --subst-rate 0.001 --steps 4 --warmup 2 --no-cpu-baseline --no-whole-job 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s frac', round(r['frac'],3), '| file order', round(r['file_order']['kernel_ms'],3), 'ms frac', round(r['frac_with_file_order'],3))"
done; done
