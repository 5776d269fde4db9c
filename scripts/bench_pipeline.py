#!/usr/bin/env python3
"""End-to-end wall clock of `kmx pipeline` (reads -> super-k-mers -> counts -> matrix, files in, files out) on a
synthetic cohort: S samples of one G-bp genome with per-sample substitutions (rate d), error-free 150-bp reads at
coverage COV, plain FASTA on local disk.  Prints the driver's per-stage seconds (one line of JSON).  The driver is
a single host thread per process: file I/O, FASTA parsing and the host<->device copies are inside these numbers."""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=24); ap.add_argument("--genome", type=float, default=1e6)
ap.add_argument("--subst-rate", type=float, default=0.001); ap.add_argument("--coverage", type=int, default=6)
ap.add_argument("--mode", default="kmer:count:bin"); ap.add_argument("--partitions", type=int, default=16)
a = ap.parse_args()
rng = np.random.default_rng(11)
G, L = int(a.genome), 150
ref = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
tmp = tempfile.mkdtemp(prefix="kmx_bench_")
t0 = time.perf_counter()
with open(os.path.join(tmp, "in.fof"), "w") as fof:
    for s in range(a.samples):
        g = ref.copy()
        mut = rng.random(G) < a.subst_rate
        g[mut] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(mut.sum()))
        n_reads = G * a.coverage // L
        starts = rng.integers(0, G - L, n_reads)
        idx = starts[:, None] + np.arange(L)[None, :]
        reads = g[idx]
        rc = rng.random(n_reads) < 0.5
        reads[rc] = comp[reads[rc]][:, ::-1]
        path = os.path.join(tmp, f"S{s}.fa")
        lines = np.empty((n_reads, L + 4), np.uint8)
        lines[:, 0] = ord(">"); lines[:, 1] = ord("r"); lines[:, 2] = ord("\n"); lines[:, 3:3 + L] = reads; lines[:, 3 + L] = ord("\n")
        lines.tofile(path)
        fof.write(f"S{s}: {path}\n")
gen_s = time.perf_counter() - t0
run = os.path.join(tmp, "run")
cmd = [os.path.join(ROOT, "kmtricks_amd", "kmx"), "pipeline", "--file", os.path.join(tmp, "in.fof"), "--run-dir", run, "--kmer-size", "31",
       "--mode", a.mode, "--hard-min", "2", "--recurrence-min", "2", "--nb-partitions", str(a.partitions), "--static-repart", "--bloom-size", "1e7"]
t0 = time.perf_counter()
r = subprocess.run(cmd, capture_output=True, text=True)
wall = time.perf_counter() - t0
line = [l for l in r.stderr.splitlines() if l.startswith("[kmx pipeline]")]
if r.returncode != 0 or not line:
    print(r.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1][len("[kmx pipeline] "):])
d.update({"mode": a.mode, "wall_s": wall, "fasta_generation_s": gen_s, "genome": G, "subst_rate": a.subst_rate, "coverage": a.coverage,
          "Mbases_per_s_end_to_end": d["bases"] / wall / 1e6})
print(json.dumps(d))
shutil.rmtree(tmp, ignore_errors=True)
