#!/usr/bin/env python3
"""kernel resource usage of one libkmx source file: scripts/dev/kres.py merge_cols.hip [name regex] [extra hipcc flags]"""
import os, re, subprocess, sys
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "."
cs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "kmtricks_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/tmp/kres.o",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], cwd=cs, capture_output=True, text=True).stderr
cur = None; rows = []
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for k, rx in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(rx, l)
        if m and cur is not None: cur[k] = m.group(1)
if not rows: print(out[-3000:])
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if re.search(pat, n): print("%-64s vgpr %4s scratch %4s occ %s lds %s" % (n, r.get("vgpr"), r.get("scratch"), r.get("occ"), r.get("lds")))
