#!/usr/bin/env python3
"""gpurun_out/r6cs/{kernel_stats.csv, counters.txt, bench.json} (scripts/r6_count_stage_profile.sh) -> profiles/count_stage_kernels.json:
per kernel of the count stage its calls and time per sample, the wave instructions it issues and two fractions --
  issue = (VALU wave instructions x 2 cycles on a SIMD-32 + the other classes x 1) / (1024 SIMDs x 2.4 GHz x its time): how much of the chip's
          issue capacity the kernel uses (MI355X_MICROARCH.md: a wave64 VALU instruction takes a SIMD-32 for 2 cycles),
  hbm   = (FETCH_SIZE x 2 + WRITE_SIZE) / its time / 8 TB/s (the counters' gfx950 correction as that guide prescribes),
and the stage's algorithmic bytes (super-k-mer bytes + 12 per distinct solid k-mer) over the sum of the kernels' times."""
import csv, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6cs")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "count_stage_kernels.json")
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
reps = 20 + 3      # timed calls + warm-up calls (round 6: the profiled runs skip the count_reads call that writes the record streams)
def short(n):
    n = n.split("(")[0]
    return n.replace("kmx::", "").replace("void ", "").strip()
ks = {}
for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
    ks[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ns": float(r["TotalDurationNs"]), "avg_ns": float(r["AverageNs"])}
cnt = collections.defaultdict(dict)
for l in open(os.path.join(src, "counters.txt")):
    k, c, v, n = l.rstrip("\n").split("\t")
    cnt[short(k)][c] = float(v)
rows = []
tot_ns = sum(v["total_ns"] for v in ks.values())
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ns"]):
    c = cnt.get(k, {})
    t = v["avg_ns"] * 1e-9
    valu, salu, lds = c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0)
    vmem = c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
    issue = (2 * valu + salu + lds + vmem) / (1024 * 2.4e9 * t) if t > 0 else None
    hbm = (2 * c.get("FETCH_SIZE", 0) * 1024 + c.get("WRITE_SIZE", 0) * 1024) / t / 8e12 if (t > 0 and ("FETCH_SIZE" in c or "WRITE_SIZE" in c)) else None
    rows.append({"kernel": k, "calls_per_sample": v["calls"] / reps, "us_per_sample": v["total_ns"] / reps / 1e3, "avg_us": v["avg_ns"] / 1e3,
                 "share": v["total_ns"] / tot_ns, "valu": valu, "salu": salu, "lds": lds, "vmem": vmem, "waves": c.get("SQ_WAVES"),
                 "issue_frac": issue, "hbm_frac": hbm, "library": k.startswith("rocprim") or "rocprim::" in k})
kern_us = tot_ns / reps / 1e3
doc = {"_note": "one 30-Mbase sample (5 Mbp x 6x, 150-bp reads, k = 31, m = 10) through kmx_count_reads_dev; rocprofv3 --kernel-trace --stats and --pmc passes of their own "
                "(scripts/r6_count_stage_profile.sh, scripts/r6_count_stage_table.py); issue_frac = (2 VALU + SALU + LDS + VMEM wave instructions) / (1024 SIMDs x 2.4 GHz x time), "
                "hbm_frac = (2 FETCH_SIZE + WRITE_SIZE) / time / 8 TB/s",
       "sample": {k: bench[k] for k in ("genome", "bases", "kmers", "superk_bytes", "distinct_solid", "partitions", "algorithmic_bytes", "count_reads_dev_ms_median")},
       "kernels_us_per_sample": kern_us, "launches_per_sample": sum(r["calls_per_sample"] for r in rows),
       "library_us_per_sample": sum(r["us_per_sample"] for r in rows if r["library"]),
       "hbm_frac_algorithmic": bench["algorithmic_bytes"] / (kern_us * 1e-6) / 8e12,
       "kernels": rows}
json.dump(doc, open(out, "w"), indent=1)
print(f"{kern_us:.0f} us of kernels per sample in {doc['launches_per_sample']:.1f} launches, {doc['library_us_per_sample']:.0f} us in library kernels; algorithmic bytes -> {doc['hbm_frac_algorithmic']:.4f} of the HBM roofline")
for r in rows[:12]:
    print(f"  {r['kernel'][:70]:70s} {r['us_per_sample']:8.1f} us  x{r['calls_per_sample']:5.1f}  issue {r['issue_frac'] if r['issue_frac'] is None else round(r['issue_frac'], 3)}  hbm {r['hbm_frac'] if r['hbm_frac'] is None else round(r['hbm_frac'], 3)}")
