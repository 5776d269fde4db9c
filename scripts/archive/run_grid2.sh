# rows vs pivot kernel over divergence at N=1000 (bench workload generator)
for D in 0.001 0.002 0.003 0.005; do for K in rows pivot; do
  echo -n "N=1000 d=$D $K: "
  KMX_TRACE=1 KMX_MERGE_KERNEL=$K timeout 300 python bench.py --subst-rate $D --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -c "re-run" | tr '\n' ' '
  KMX_MERGE_KERNEL=$K timeout 300 python bench.py --subst-rate $D --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('fallbacks;', r['kernel'], round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step')"
done; done
