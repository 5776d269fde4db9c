#!/bin/bash
# round-6 evidence, part A: the driver's bench command, the pair's two kernels apart (HIP events around each), the GPU tests.
# -> gpurun_out/r6e/ (copied to profiles/ by hand)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
line() { grep '^{"metric' | tail -1; }
( time python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/err_all.log | line > $O/bench_all.json ) 2> $O/time_all.txt
python scripts/r5_grid.py --wl count --steps 10 --settings "x=1" > $O/pair_parts.jsonl 2>$O/err_parts.log
python scripts/r5_grid.py --wl pa63 --steps 10 --settings "x=1" >> $O/pair_parts.jsonl 2>>$O/err_parts.log
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/gpu_tests.txt
python -c "
import json; d=json.load(open('$O/bench_all.json')); r=d['roofline']
print('headline', r['kernel'], r['kernel_ms'], 'frac', r['frac'], 'value', d['value'], 'ms/step', d['ms_per_step'])
for k,v in d.get('workloads',{}).items():
    rr=v.get('roofline',{}) if isinstance(v,dict) else {}
    print(k, rr.get('kernel_ms'), rr.get('frac'), rr.get('frac_with_file_order'))
"
cat $O/pair_parts.jsonl $O/gpu_tests.txt $O/time_all.txt
