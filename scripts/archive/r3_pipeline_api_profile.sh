# HIP API and kernel statistics of `kmx pipeline` (count stage only: --until count would write files; here the whole run on 300 samples)
cd /tmp && export TMPDIR=/tmp
T=$(mktemp -d); O=$GRAFT_REPO_ROOT/gpurun_out/r3_api; rm -rf $O; mkdir -p $O
python - "$T" <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from multiprocessing import Pool
T = sys.argv[1]
with Pool(32) as p:
    paths = p.map(bench._pipeline_make_sample, [(s, 1000000, 0.001, 6, T) for s in range(300)])
with open(f"{T}/in.fof", "w") as f:
    for s, pth in enumerate(paths): f.write(f"S{s:04d}: {pth}\n")
PY
KMX_SLOW_EXIT=1 timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -d $O/prof --output-format csv -- $GRAFT_REPO_ROOT/kmtricks_amd/kmx pipeline --file $T/in.fof --run-dir $T/run --nb-partitions 256 --static-repart --recurrence-min 2 -t 64 > $O/log.txt 2>&1
tail -2 $O/log.txt | cut -c1-600
find $O/prof -name "*hip_api_stats.csv" -exec cp {} $O/hip_api_stats.csv \;
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*trace.csv" -delete
head -25 $O/hip_api_stats.csv | cut -c1-160
head -12 $O/kernel_stats.csv | cut -c1-160
rm -rf $T
