# the three COUNT kernels over sample counts and divergence (bench workload generator): ms per step / kernel ms; a forced
# kernel that hands tasks back shows the time of the whole chain
for cfg in "1000 0.001" "1000 0.002" "1000 0.003" "700 0.001" "500 0.001" "384 0.001" "300 0.001" "200 0.001" "128 0.001"; do set -- $cfg; for K in rows pivot cols; do
  echo -n "N=$1 d=$2 $K: "
  KMX_TRACE=1 KMX_MERGE_KERNEL=$K timeout 300 python bench.py --samples $1 --subst-rate $2 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
back=0
for l in sys.stdin:
    if 'handed back' in l: back+=1
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s  hand-backs', back)"
done; done
