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
