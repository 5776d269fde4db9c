export TMPDIR=/tmp; O=gpurun_out/gap; rm -rf $O; mkdir -p $O
python scripts/bench_pipeline.py --samples ${SAMPLES:-120} --genome 5e6 --partitions 256 --tmp /dev/shm --env "KMX_SLOW_EXIT=1" --prof $O --prof-flags=--memory-copy-trace --keep-trace > $O/lines.jsonl 2> $O/err.txt
python - $O <<'PY'
import csv, sys, glob
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rd = csv.DictReader(open(f)); 
    for r in rd: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s " % r.get("Queue_Id", "?") + r["Kernel_Name"].replace("void ", "").replace("kmx::", "")[:34]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "")[12:] ))
ev.sort(); t0 = ev[0][0]
mid = t0 + int(0.35 * (ev[-1][1] - t0))
for s, e, n in ev:
    if mid <= s < mid + 9_000_000: print("%9.3f +%7.3f  %s" % ((s - mid) / 1e6, (e - s) / 1e6, n))
PY
