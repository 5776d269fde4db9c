cd $GRAFT_REPO_ROOT
timeout 600 python -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(60, repeat=True)
sys.argv=['bench.py','--workload','pa63','--steps','5','--warmup','2']
runpy.run_path('bench.py', run_name='__main__')
" 2>&1 | grep -v "^\[bench\] counted" | tail -60 | cut -c1-200
