# GPU timeline of a `kmx pipeline --until count`-free run: the longest kernels / copies and the longest idle gaps
export TMPDIR=/tmp; O=gpurun_out/gap; rm -rf $O; mkdir -p $O
python scripts/bench_pipeline.py --samples ${SAMPLES:-120} --genome 5e6 --partitions 256 --tmp /dev/shm --env "KMX_SLOW_EXIT=1" --prof $O --prof-flags=--memory-copy-trace --keep-trace > $O/lines.jsonl 2> $O/err.txt
python - $O <<'PY'
import csv, sys, glob
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
ev.sort()
print(len(ev), "events, span %.1f ms" % ((ev[-1][1] - ev[0][0]) / 1e6))
print("longest events:")
for s, e, n in sorted(ev, key=lambda x: x[0] - x[1])[:15]: print("  %.2f ms at %.1f ms  %s" % ((e - s) / 1e6, (s - ev[0][0]) / 1e6, n))
# idle gaps
end = ev[0][1]; gaps = []
for s, e, n in ev[1:]:
    if s > end: gaps.append((s - end, end, n))
    end = max(end, e)
busy = 0; end = ev[0][0]
for s_, e_, n in ev:
    if e_ > end: busy += e_ - max(s_, end); end = e_
print("busy (union of kernels and copies) %.1f ms of %.1f ms; gaps < 0.2 ms: %d sum %.1f ms; gaps 0.2-2 ms: %d sum %.1f ms" % (busy / 1e6, (ev[-1][1] - ev[0][0]) / 1e6,
      sum(1 for g in gaps if g[0] < 2e5), sum(g[0] for g in gaps if g[0] < 2e5) / 1e6, sum(1 for g in gaps if 2e5 <= g[0] < 2e6), sum(g[0] for g in gaps if 2e5 <= g[0] < 2e6) / 1e6))
kb = 0; end = ev[0][0]
for s_, e_, n in ev:
    if n.startswith("COPY"): continue
    if e_ > end: kb += e_ - max(s_, end); end = e_
print("busy with kernels alone %.1f ms" % (kb / 1e6))
print("idle gaps > 2 ms: %d, sum %.1f ms" % (sum(1 for g in gaps if g[0] > 2e6), sum(g[0] for g in gaps if g[0] > 2e6) / 1e6))
for g, at, n in sorted(gaps, reverse=True)[:15]: print("  %.2f ms idle at %.1f ms, then %s" % (g / 1e6, (at - ev[0][0]) / 1e6, n))
PY
