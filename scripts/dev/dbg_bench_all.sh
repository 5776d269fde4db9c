cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
( time timeout 1500 python -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(100, repeat=True)
sys.argv=['bench.py','--gpus','1','--steps','20','--warmup','5']
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/r5e/bench_all_2.out 2> gpurun_out/r5e/bench_all_2.err ) 2> gpurun_out/r5e/time_all_2.txt
grep '^{"metric' gpurun_out/r5e/bench_all_2.out | tail -1 > gpurun_out/r5e/bench_all_2.json
grep -v "counted lists" gpurun_out/r5e/bench_all_2.err | tail -80 | cut -c1-180; cat gpurun_out/r5e/time_all_2.txt
