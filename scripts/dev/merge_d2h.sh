# the merge stage's device-to-host copies on the GPU's timeline: how busy is the link between the first and the last of them?
export TMPDIR=/tmp; O=gpurun_out/d2h; rm -rf $O; mkdir -p $O
python scripts/bench_pipeline.py --samples ${SAMPLES:-1000} --genome ${GENOME:-5e6} --partitions 256 --tmp /dev/shm --env "KMX_SLOW_EXIT=1 ${ENVX:-}" --prof $O --prof-flags=--memory-copy-trace --keep-trace > $O/lines.jsonl 2> $O/err.txt
python - $O <<'PY'
import csv, sys, glob, json
for l in open(sys.argv[1] + "/lines.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print({k: d[k] for k in ("count_wall_s", "merge_wall_s", "total_s", "matrix_bytes")})
cp = []; kn = []
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd: cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), r))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "merge" in r["Kernel_Name"] or "cols" in r["Kernel_Name"]: kn.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
print("columns:", list(cp[0][3].keys()) if cp else None)
d2h = sorted(c for c in cp if "DEVICE_TO_HOST" in c[2] and c[1] - c[0] > 200_000)      # (the pieces: copies of more than 0.2 ms)
if d2h:
    t0, t1 = d2h[0][0], max(c[1] for c in d2h)
    busy = 0; end = t0
    for s, e, _, _ in d2h:
        if e > end: busy += e - max(s, end); end = e
    print("pieces: %d, first to last %.1f ms, link busy %.1f ms (%.0f %%), mean piece %.2f ms" % (len(d2h), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(c[1] - c[0] for c in d2h) / len(d2h) / 1e6))
    gaps = []; end = d2h[0][1]
    for s, e, _, _ in d2h[1:]:
        if s > end: gaps.append((s - end, end - t0))
        end = max(end, e)
    gaps.sort(reverse=True)
    print("largest gaps between pieces (ms, at ms):", [(round(g / 1e6, 1), round(a / 1e6)) for g, a in gaps[:16]])
    kn.sort()
    print("merge kernels: %d, %.1f ms in all; first at %.1f ms before the first piece" % (len(kn), sum(e - s for s, e, _ in kn) / 1e6, (t0 - kn[0][0]) / 1e6 if kn else 0))
PY
rm -rf $O/run0
