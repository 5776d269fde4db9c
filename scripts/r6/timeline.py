#!/usr/bin/env python3
"""rocprofv3 kernel + memcpy traces -> the last call's timeline: every launch / copy with its start, duration and the idle gap in front of it"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# a call starts at its k_pack_bases (the sync-free path: behind one fill) or, the old path, at its first k_superk_wave<false
idx = [i for i, e in enumerate(ev) if "k_pack_bases" in e[2]]
old = [i for i, e in enumerate(ev) if "k_superk_wave<false" in e[2]]
start = idx[-1] if idx else 0
if old and old[-1] < start and start - old[-1] < 25: start = old[-1]
while start > 0 and ev[start - 1][2].startswith("__amd_rocclr_fill") and ev[start][0] - ev[start - 1][1] < 100_000: start -= 1
sel = ev[start:]
t0 = sel[0][0]; busy = 0; prev_end = t0
for s, e, n in sel:
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:8.1f}  {n}")
    busy += e - s; prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, launches {len(sel)}")
