"""Seeded synthetic partitions (sorted per-sample count lists) for the parity tests."""
import numpy as np


def synth_lists(seed, n_lists, pool, p_present, n_private, kw=1, key_bits=62, count_max=50, ragged=False):
    """`pool` shared keys, each present in a sample with probability p_present, plus n_private
    keys private to each sample.  -> list of (keys uint64[n, kw], counts uint32[n]) ascending."""
    rng = np.random.default_rng(seed)
    total = pool + n_lists * n_private

    def draw(n):
        lo = rng.integers(0, 1 << min(key_bits, 62), n, dtype=np.uint64)
        if kw == 1:
            return lo.reshape(n, 1)
        if kw == 2:
            hi = rng.integers(0, 1 << max(1, min(key_bits - 64, 62)), n, dtype=np.uint64)
            return np.stack([lo, hi], axis=1)
        # three and four words (k = 64 ... 127): the upper words take one of eight values spread over all 64 bits (the top bit
        # included), so that many keys agree in some of their words and every word decides some comparisons
        ws = [lo if rng.random() < 0.9 else (lo & np.uint64(7))]
        for _ in range(kw - 1):
            ws.append(rng.integers(0, 8, n, dtype=np.uint64) * np.uint64(0x2492492492492492))
        top_bits = max(1, min(key_bits - 64 * (kw - 1), 62))
        ws[-1] = ws[-1] >> np.uint64(64 - top_bits) if top_bits < 62 else ws[-1]
        return np.stack(ws, axis=1)

    allk = np.unique(draw(int(total * 1.1) + 8), axis=0)
    rng.shuffle(allk, axis=0)
    allk = allk[:total]
    shared, priv = allk[:pool], allk[pool:]
    out = []
    for i in range(n_lists):
        pp = p_present if not ragged else p_present * rng.random()
        mask = rng.random(pool) < pp
        ks = np.concatenate([shared[mask], priv[i * n_private:(i + 1) * n_private]], axis=0)
        if ragged and i % 7 == 3:
            ks = ks[:0]
        # ascending, most significant word first
        order = np.lexsort([ks[:, j] for j in range(kw)])
        ks = np.ascontiguousarray(ks[order])
        cs = rng.integers(1, count_max, len(ks), dtype=np.uint32)
        out.append((ks, cs))
    return out


def synth_hash_lists(seed, n_lists, lower, window, density, count_max=20):
    """hash-mode lists: each sample holds ~density*window distinct hashes of [lower, lower+window)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_lists):
        n = int(window * density * (0.5 + rng.random()))
        hs = np.unique(rng.integers(lower, lower + window, n, dtype=np.uint64))
        cs = rng.integers(1, count_max, len(hs), dtype=np.uint32)
        out.append((hs.reshape(-1, 1), cs))
    return out


_NT = {"A": 0, "C": 1, "T": 2, "G": 3}


def kmer_value(s):
    """a k-mer's value as the reference encodes it: A0 C1 T2 G3, the first nucleotide in the top digit (kmer.hpp:938)"""
    v = 0
    for c in s:
        v = v * 4 + _NT[c]
    return v


def canonical_value(s):
    rc = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    return min(kmer_value(s), kmer_value(rc))


def superk_record(seq, k):
    """the super-k-mer record of the len(seq) - k + 1 consecutive k-mers of `seq` (gatb Model.hpp:1388-1433 as sorting_count.hpp:153-275
    reads it): [u8 n][ceil((k + n - 1) / 4) bytes of S, little endian], S = the first k-mer's value + the following nucleotides at
    digits k, k + 1, ..."""
    n = len(seq) - k + 1
    assert 1 <= n <= 255
    S = kmer_value(seq[:k])
    for j, c in enumerate(seq[k:]):
        S |= _NT[c] << (2 * (k + j))
    return bytes([n]) + S.to_bytes((k + n - 1 + 3) // 4, "little")


def synth_superk_stream(seed, k, n_records, max_kmers, genome=4000):
    """records cut at random places of a random genome (so that k-mers repeat), 1 .. max_kmers k-mers each -> (bytes, {canonical value: count})"""
    rng = np.random.default_rng(seed)
    g = "".join("ACGT"[i] for i in rng.integers(0, 4, genome + k + max_kmers))
    out, counts = [], {}
    for _ in range(n_records):
        n = int(rng.integers(1, max_kmers + 1)); at = int(rng.integers(0, genome))
        seq = g[at:at + k + n - 1]
        if rng.random() < 0.5:
            seq = seq[::-1].translate(str.maketrans("ACGT", "TGCA"))
        out.append(superk_record(seq, k))
        for j in range(n):
            v = canonical_value(seq[j:j + k])
            counts[v] = counts.get(v, 0) + 1
    return b"".join(out), counts
