# BASELINE configs[4] (k = 63, PA rows): recurrence-min 1 (kmtricks' default) against 2, round-1 generator (quick set-up)
cd $GRAFT_REPO_ROOT
for rm in 1 2; do echo -n "pa63 rec-min $rm: "; python bench.py --workload pa63 --lists random --rec-min $rm --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['ms_per_step'],2), r['kernel'], round(r['kernel_ms'],2), round(r['frac'],3), d['config']['rows_out_per_step_per_gpu'])"; done
