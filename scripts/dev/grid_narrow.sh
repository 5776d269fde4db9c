for v in 1 0; do for N in 500 200 1000 300; do
  echo -n "narrow=$v N=$N: "
  KMX_DENSE_NARROW=$v KMX_MERGE_KERNEL=cols timeout 300 python bench.py --workload count --lists random --samples $N --subst-rate 0.001 --steps 4 --warmup 2 --no-cpu-baseline --no-whole-job 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(round(r['kernel_ms'],3), 'ms arena frac', round(r['frac'],3), '| file order', round(r['file_order']['kernel_ms'],3), 'ms frac', round(r['frac_with_file_order'],3))"
done; done
