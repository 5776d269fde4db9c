# per-kernel durations of the bench step with nothing else in flight (scripts/serial_steps.py under rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/serial
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python $GRAFT_REPO_ROOT/scripts/serial_steps.py > $OUT/log.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "kmx::" in n or "k_ctrl" in n:
        print(f"{(n.split('kmx::')[1] if 'kmx::' in n else n).split('(')[0][:28]:30s} calls {r['Calls']:>3s}  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
