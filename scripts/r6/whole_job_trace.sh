#!/bin/bash
# round 6: kernel trace of the whole job of configs[2] on one GPU (8 batches of 32 partitions, two in flight): where 56 ms go when a batch's pair takes 4.8
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6wj${TAG:-}
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload count --steps 3 --warmup 1 --no-cpu-baseline ${ARGS:-} > $O/bench.json 2> $O/trace.log
python - <<PY > $O/timeline.txt
import csv, glob
ev = []
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-58:]))
for f in glob.glob("$O/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[12:] + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the whole job's timed pass: the last 8 k_merge_cols<..true, true> launches that are not followed by arena-mode ones; print everything from the 16th-last ORD cols launch on
idx = [i for i, e in enumerate(ev) if "k_merge_cols<0, false, false, true, true>" in e[2]]
# the arena steps come last; find the 16 consecutive ORD launches of the two job passes: take those before the first arena-mode launch
ar = [i for i, e in enumerate(ev) if "k_merge_cols<0, false, false, false, false>" in e[2]]
cut = ar[0] if ar else len(ev)
idx = [i for i in idx if i < cut]
start = idx[-8] if len(idx) >= 8 else 0
while start > 0 and ("k_range_bounds" in ev[start - 1][2] or "k_cols_skel" in ev[start - 1][2] or "k_cols_prep" in ev[start - 1][2] or ev[start - 1][2].startswith(("COPY", "__amd"))) and ev[start][0] - ev[start - 1][1] < 2_000_000: start -= 1
end = idx[-1]
while end + 1 < cut and ev[end + 1][0] - ev[end][1] < 3_000_000 and "k_merge_cols<0, false, false, false, false>" not in ev[end + 1][2]: end += 1
sel = ev[start:end + 1]
t0 = sel[0][0]; pe = t0
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f}  gap {(s - pe) / 1e3:8.1f}  {n}")
    pe = max(pe, e)
print(f"span {(pe - t0) / 1e3:.1f} us")
PY
rm -rf $O/trace
tail -70 $O/timeline.txt; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['roofline'].get('whole_job')); print(d['ms_per_step'], d['roofline']['frac'])"
