# rows vs pivot kernel over sample counts and divergence (bench workload generator)
for N in 1000 300 100 30; do for D in 0.001 0.01; do for K in rows pivot; do
  echo -n "N=$N d=$D $K: "
  KMX_TRACE=1 KMX_MERGE_KERNEL=$K timeout 300 python bench.py --samples $N --subst-rate $D --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -c "re-run" | tr '\n' ' '
  KMX_MERGE_KERNEL=$K timeout 300 python bench.py --samples $N --subst-rate $D --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('fallbacks;', round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,1), 'Gk/s rows', d['config']['rows_out_per_step_per_gpu'])"
done; done; done
