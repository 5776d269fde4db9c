#!/usr/bin/env python3
"""First timing of the super-k-mer split and count stages (host buffers in/out through the C ABI, so
PCIe and the host-side record-offset pass are inside the numbers) against the oracle on one core.
One synthetic sample: 5 Mbp genome, 150-bp error-free reads at 6x, k=31, m=10, 32 static partitions."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kmtricks_amd import lib
import orc

rng = np.random.default_rng(20240601)
G, L, COV, K, M, P = 5_000_000, 150, 6, 31, 10, 32
genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
n_reads = G * COV // L
starts = rng.integers(0, G - L, n_reads)
strand = rng.random(n_reads) < 0.5
reads = []
for s, st in zip(starts, strand):
    r = genome[s:s + L]
    reads.append((comp[r][::-1] if st else r).tobytes())
ctx = lib.Context(0)
rep = orc.repart_static(M, P)
packed = ctx.pack_reads(reads)
ctx.superk_partition(packed, K, M, rep, P)                           # warm-up at full size (module load, pools)
t0 = time.perf_counter(); sk = ctx.superk_partition(packed, K, M, rep, P); t_sk = time.perf_counter() - t0
nk = sum(x[1] for x in sk); nb = sum(len(x[0]) for x in sk)
ctx.count_kmer(sk[0][0], K, 2)
t0 = time.perf_counter(); cnt = [ctx.count_kmer(sk[p][0], K, 2) for p in range(P)]; t_ck = time.perf_counter() - t0
t0 = time.perf_counter(); cnth = [ctx.count_hash(sk[p][0], K, 3125056, p, 2) for p in range(P)]; t_ch = time.perf_counter() - t0
streams = [x[0] for x in sk]
ctx.count_batch(streams, K, 2); ctx.count_batch(streams, K, 2, window=3125056)      # warm-up at full size (pinned pools)
t0 = time.perf_counter(); bk = ctx.count_batch(streams, K, 2); t_bk = time.perf_counter() - t0
t0 = time.perf_counter(); bh = ctx.count_batch(streams, K, 2, window=3125056); t_bh = time.perf_counter() - t0
batch_ok = all(np.array_equal(bk[p][0], cnt[p][0]) and np.array_equal(bk[p][1], cnt[p][1]) and
               np.array_equal(bh[p][0], cnth[p][0]) and np.array_equal(bh[p][1], cnth[p][1]) for p in range(P))
ctx.count_reads(packed, K, M, rep, P, 2); ctx.count_reads(packed, K, M, rep, P, 2, window=3125056)      # warm-up
t0 = time.perf_counter(); fk, fnk, _, _ = ctx.count_reads(packed, K, M, rep, P, 2); t_fk = time.perf_counter() - t0
t0 = time.perf_counter(); fh, _, _, _ = ctx.count_reads(packed, K, M, rep, P, 2, window=3125056); t_fh = time.perf_counter() - t0
fused_ok = all(np.array_equal(fk[p][0], cnt[p][0]) and np.array_equal(fk[p][1], cnt[p][1]) and
               np.array_equal(fh[p][0], cnth[p][0]) and np.array_equal(fh[p][1], cnth[p][1]) for p in range(P))
# oracle on one core, bounded sample: 1/8 of the reads for the split, 4 partitions for the counts
lut = orc.minimizer_lut(M)
t0 = time.perf_counter(); osk = orc.superk_partition(reads[: n_reads // 8], K, M, lut, rep, P); t_osk = time.perf_counter() - t0
t0 = time.perf_counter(); oc = [orc.count_kmer(sk[p][0], K, 2) for p in range(4)]; t_ock = time.perf_counter() - t0
ok = all(np.array_equal(oc[p][0], cnt[p][0]) and np.array_equal(oc[p][1], cnt[p][1]) for p in range(4))
print(json.dumps({"reads": n_reads, "bases": n_reads * L, "kmers": nk, "superk_bytes": nb, "distinct_solid": int(sum(len(c[1]) for c in cnt)),
                  "gpu_superk_s": t_sk, "gpu_superk_Mbases_per_s": n_reads * L / t_sk / 1e6,
                  "gpu_count_kmer_s": t_ck, "gpu_count_kmer_Mkmers_per_s": nk / t_ck / 1e6,
                  "gpu_count_hash_s": t_ch, "gpu_count_hash_Mkmers_per_s": nk / t_ch / 1e6,
                  "gpu_count_batch_kmer_s": t_bk, "gpu_count_batch_kmer_Mkmers_per_s": nk / t_bk / 1e6,
                  "gpu_count_batch_hash_s": t_bh, "gpu_count_batch_hash_Mkmers_per_s": nk / t_bh / 1e6,
                  "batch_equals_per_partition": bool(batch_ok),
                  "gpu_count_reads_kmer_s": t_fk, "gpu_count_reads_kmer_Mkmers_per_s": nk / t_fk / 1e6, "gpu_count_reads_kmer_Mbases_per_s": n_reads * L / t_fk / 1e6,
                  "gpu_count_reads_hash_s": t_fh, "gpu_count_reads_hash_Mkmers_per_s": nk / t_fh / 1e6, "fused_equals_per_partition": bool(fused_ok),
                  "count_algorithmic_bytes": nb + 12 * int(sum(len(c[1]) for c in cnt)),
                  "oracle_superk_Mbases_per_s_1core": (n_reads // 8) * L / t_osk / 1e6,
                  "oracle_count_kmer_Mkmers_per_s_1core": sum(sk[p][1] for p in range(4)) / t_ock / 1e6,
                  "count_bit_exact_vs_oracle_4_partitions": bool(ok)}))
