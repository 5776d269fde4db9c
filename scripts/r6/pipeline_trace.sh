#!/bin/bash
# round 6: kernel + copy trace of `kmx pipeline` (SAMPLES x 5 Mbp in a RAM file system): how busy the GPU is during the count stage
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6ptrace${TAG:-}
rm -rf $O; mkdir -p $O
python $GRAFT_REPO_ROOT/scripts/bench_pipeline.py --samples ${SAMPLES:-200} --genome 5e6 --partitions 256 --tmp /dev/shm --extra "--hard-min 2 --recurrence-min 2 --static-repart ${EXTRA:-}" \
  --variants ";" --env "KMX_SLOW_EXIT=1 ${ENVX:-};KMX_SLOW_EXIT=1 ${ENVX:-}" --prof $O/prof --prof-flags=--memory-copy-trace --keep-trace > $O/lines.jsonl 2> $O/err.log
python - <<PY
import csv, glob, json
d = "$O/prof"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", r.get("Stream_Id", ""))))
cp = []
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), int(r.get("Bytes", r.get("Size", 0)) or 0)))
ev.sort()
# the last run's count stage: from its first k_pack_bases to its last count kernel
runs = []; cur = []
for e in ev:
    if cur and e[0] - cur[-1][1] > 300_000_000: runs.append(cur); cur = []
    cur.append(e)
runs.append(cur)
last = runs[-1]
cnt = [e for e in last if "k_cs_" in e[2] or "k_sk_" in e[2] or "k_superk" in e[2] or "k_pack" in e[2] or "k_part_stats" in e[2]]
t0, t1 = cnt[0][0], max(e[1] for e in cnt)
# union of busy intervals of all kernels in [t0, t1]
iv = sorted((max(e[0], t0), min(e[1], t1)) for e in last if e[1] > t0 and e[0] < t1)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
import collections
# the idle gaps of the union: which kernel ended in front of a gap, which one started behind it
allk = sorted((e for e in last if e[1] > t0 and e[0] < t1), key=lambda e: e[0])
gaps = collections.defaultdict(lambda: [0.0, 0]); ce = allk[0][1]; cek = allk[0][2]
for e in allk[1:]:
    if e[0] > ce:
        g = gaps[(cek[-28:], e[2][-28:])]; g[0] += (e[0] - ce) / 1e3; g[1] += 1
    if e[1] > ce: ce, cek = e[1], e[2]
# a window of the steady state: every kernel and copy with its queue, from the 150th sample's walk on (8 ms)
walks = [e for e in last if "k_superk_wave" in e[2]]
if len(walks) > 160:
    w0 = walks[150][0]
    print("window (us from the 150th walk; dur; queue; name):")
    for e in sorted([x for x in last if w0 - 200_000 <= x[0] <= w0 + 8_000_000] + [(c[0], c[1], "COPY " + c[2][12:] + " " + str(c[3]), "-") for c in cp if w0 - 200_000 <= c[0] <= w0 + 8_000_000]):
        print(f"  {(e[0] - w0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f}  q{e[3]}  {e[2][-44:]}")
allev = sorted(last + [(c[0], c[1], "COPY " + c[2][12:] + " " + str(c[3]), "-") for c in cp if c[0] >= last[0][0] - 2_000_000_000])
print("the run's first 45 kernels / copies (ms from the first; dur us):")
for e in allev[:45]: print(f"  {(e[0] - allev[0][0]) / 1e6:9.3f} {(e[1] - e[0]) / 1e3:9.1f}  q{e[3]}  {e[2][-44:]}")
wk = [e for e in last if "k_superk_wave" in e[2]]
print("walk starts (ms from the first event), every 25th:", [round((w[0] - allev[0][0]) / 1e6, 1) for w in wk[::25]])
# (the long intervals between two walks: when, and what the GPU and the link did inside)
iv_w = [(wk[i + 1][0] - wk[i][0]) / 1e3 for i in range(len(wk) - 1)]
import statistics
print("walk-to-walk us: median", round(statistics.median(iv_w), 1), "mean", round(sum(iv_w) / len(iv_w), 1), "over 3 ms:", sum(1 for x in iv_w if x > 3000), "their sum ms", round(sum(x for x in iv_w if x > 3000) / 1e3, 1))
shown = 0
for i, x in enumerate(iv_w):
    if x > 3000 and shown < 6 and i > 30:
        shown += 1
        a, b = wk[i][0], wk[i + 1][0]
        print(f" long interval {x / 1e3:.1f} ms at sample {i} ({(a - allev[0][0]) / 1e6:.1f} ms):")
        for e in [e for e in allev if a <= e[0] <= b][:60]: print(f"    {(e[0] - a) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f}  q{e[3]}  {e[2][-44:]}")
print("idle gaps (us in all, count) by kernel in front -> kernel behind:")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]: print(f"  {v[0]:10.0f} us  x{v[1]:5d}  {k[0]} -> {k[1]}")
per = collections.defaultdict(float)
for e in cnt: per[e[2]] += (e[1] - e[0]) / 1e3
h2d = sum(c[3] for c in cp if c[0] >= t0 and c[1] <= t1 and "HOST_TO_DEVICE" in c[2]); h2dt = sum(c[1] - c[0] for c in cp if c[0] >= t0 and c[1] <= t1 and "HOST_TO_DEVICE" in c[2])
npk = sum(1 for e in cnt if "k_pack_bases" in e[2])
print(json.dumps({"count_stage_span_ms": (t1 - t0) / 1e6, "gpu_busy_ms": busy / 1e6, "busy_frac": busy / (t1 - t0), "samples": npk, "kernel_ms_sum": sum(per.values()) / 1e3,
                  "h2d_GB": h2d / 1e9, "h2d_busy_ms": h2dt / 1e6, "per_kernel_us_per_sample": {k[-40:]: round(v / max(npk, 1), 1) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]}}))
PY
rm -rf $O/prof
