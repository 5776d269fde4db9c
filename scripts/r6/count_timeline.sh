#!/bin/bash
# round 6: kernel + memcpy timeline of kmx_count_reads_dev on one 30-Mbase sample: where the device idles between launches
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6tl${TAG:-}
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r6/count_calls.py ${ARGS:-} > $O/bench.json 2> $O/trace.log
python $GRAFT_REPO_ROOT/scripts/r6/timeline.py $O/trace > $O/timeline.txt
rm -rf $O/trace
tail -80 $O/timeline.txt; cat $O/bench.json
