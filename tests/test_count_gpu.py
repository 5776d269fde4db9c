"""Parity of the HIP count path and bit transpose against the oracle (and the reference goldens)."""
import json, os
import numpy as np
import pytest

import orc
from test_oracle_goldens import repart_table, read_fasta, GD, G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from kmtricks_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


def random_reads(seed, n_reads, length, n_rate=0.002):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_reads):
        s = rng.choice(list("ACGT"), size=length)
        bad = rng.random(length) < n_rate
        s[bad] = "N"
        out.append("".join(s))
    return out


def test_count_reference_goldens(ctx):
    """tests/task_main.cpp:118-508 through the HIP path: k-mers and window hashes of partition 0 in file order"""
    lut = orc.minimizer_lut(10)
    rep = repart_table()
    cf = G["task_main"]["count_files"]
    for name, f in (("D1", "1.fasta"), ("D2", "2.fasta")):
        sk = orc.superk_partition(read_fasta(os.path.join(GD, f)), 31, 10, lut, rep, 4)
        for p in range(4):
            keys, counts = ctx.count_kmer(sk[p][0], 31, 1)
            exp = cf[f"counts/partition_{p}/{name}.kmer"]
            if "kmers" in exp:
                assert [[orc.kmer_to_string(k, 31), int(c)] for k, c in zip(keys, counts)] == exp["kmers"]
            else:
                assert len(counts) == exp["n"]
            hs, hc = ctx.count_hash(sk[p][0], 31, 25000000, p, 1)
            exph = cf[f"counts/partition_{p}/{name}.hash"]
            if "hashes" in exph:
                assert [[int(h), int(c)] for h, c in zip(hs, hc)] == exph["hashes"]
            else:
                assert len(hc) == exph["n"]


def test_processor_test_hard_min(ctx):
    """tests/processor_test.cpp:10-75 through the HIP count path: a 20-mer counted twice and one counted six times under
    abundance-min 3 -- only the second comes out, with its count (count_processor.hpp:61-70, 135-146), k-mer and hash mode"""
    Gp = G["processor_test"]
    for kind in ("kmer", "hash"):
        g = Gp[kind]
        k, amin = g["kmer_size"], g["abundance_min"]
        fed = g["fed_counts"] if kind == "kmer" else [c for _, c in g["fed"]]
        kept = g["kept_counts"] if kind == "kmer" else [c for _, c in g["kept"]]
        rng = np.random.default_rng(3)
        kmers = ["".join(rng.choice(list("ACGT"), size=k)) for _ in fed]
        reads = [km for km, c in zip(kmers, fed) for _ in range(c)]
        lut = orc.minimizer_lut(8); rep = orc.repart_static(8, 1)
        sk = orc.superk_partition(reads, k, 8, lut, rep, 1)
        if kind == "kmer":
            keys, counts = ctx.count_kmer(sk[0][0], k, amin)
            ek, ec = orc.count_kmer(sk[0][0], k, amin)
        else:
            keys, counts = ctx.count_hash(sk[0][0], k, 1 << 20, 0, amin)
            ek, ec = orc.count_hash(sk[0][0], k, 1 << 20, 0, amin)
        assert [int(c) for c in counts] == kept and np.array_equal(np.asarray(keys).reshape(-1), np.asarray(ek).reshape(-1))
        out, _, _, _ = ctx.count_reads(reads, k, 8, rep, 1, amin, window=(1 << 20) if kind == "hash" else 0)      # the fused split + count too
        assert [int(c) for c in out[0][1]] == kept


@pytest.mark.parametrize("k,m", [(31, 10), (21, 8), (32, 10), (47, 11), (63, 10), (20, 7)])
def test_count_random_reads_vs_oracle(ctx, k, m):
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, 8)
    reads = random_reads(100 + k, 600, 150) + ["ACGT" * 60] * 5 + ["A" * 200, "ACGTN" * 30, "ACG"]
    sk = orc.superk_partition(reads * 2, k, m, lut, rep, 8)
    tot = 0
    for p in range(8):
        for hard_min in (1, 2, 3):
            ek, ec = orc.count_kmer(sk[p][0], k, hard_min)
            gk, gc = ctx.count_kmer(sk[p][0], k, hard_min)
            assert np.array_equal(ek, gk) and np.array_equal(ec, gc)
            eh, ehc = orc.count_hash(sk[p][0], k, 1000003, p, hard_min)
            gh, ghc = ctx.count_hash(sk[p][0], k, 1000003, p, hard_min)
            assert np.array_equal(eh, gh) and np.array_equal(ehc, ghc)
        tot += len(orc.count_kmer(sk[p][0], k, 1)[1])
    assert tot > 1000


@pytest.mark.parametrize("k,m", [(31, 10), (47, 11), (20, 7)])
def test_count_batch_vs_oracle(ctx, k, m):
    """kmx_count_batch (all partition streams of a sample at once) == oracle per partition"""
    P = 16
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(7 + k, 900, 120) + ["ACGT" * 60] * 5 + ["A" * 200]
    sk = orc.superk_partition(reads * 2, k, m, lut, rep, P)
    streams = [sk[p][0] for p in range(P)]
    streams[3] = b""                          # an empty partition in the middle of the batch
    for hard_min in (1, 2):
        got = ctx.count_batch(streams, k, hard_min)
        goth = ctx.count_batch(streams, k, hard_min, window=1000003, partitions=list(range(P)))
        for p in range(P):
            ek, ec = orc.count_kmer(streams[p], k, hard_min)
            assert np.array_equal(ek, got[p][0]) and np.array_equal(ec, got[p][1])
            eh, ehc = orc.count_hash(streams[p], k, 1000003, p, hard_min)
            assert np.array_equal(eh, goth[p][0]) and np.array_equal(ehc, goth[p][1])
    assert all(len(c) == 0 for _, c in ctx.count_batch([b"", b""], k, 1))


@pytest.mark.parametrize("k", [64, 65, 80, 95, 96, 97, 112, 127])
def test_count_wide_kmers_vs_oracle(ctx, k):
    """k = 64 ... 127 (Kmer<96> / Kmer<128>; the reference's default KMER_LIST "32 64 96 128", CMakeLists.txt:25-27): kmx_count_batch on
    super-k-mer record streams of up to 92 / 124 k-mers a record == the oracle per partition (k-mers of ceil(k / 32) words, low word
    first; window hashes over those 16 / 24 / 32 key bytes), == the counts of the strings the records were cut from; then those lists merged (k_merge_rows, 3 / 4 words)"""
    from synth import synth_superk_stream
    P, mx, kw = 5, (92 if k < 96 else 124), orc.kw_of_k(k)
    streams, truth = [], []
    for p in range(P):
        recs, cnt = synth_superk_stream(1000 * k + p, k, 400 if p != 2 else 0, mx if p != 1 else 1, genome=3000)
        streams.append(recs); truth.append(cnt)
    for hard_min in (1, 2):
        got = ctx.count_batch(streams, k, hard_min)
        goth = ctx.count_batch(streams, k, hard_min, window=1000003, partitions=[7, 0, 3, 1, 12])
        for p in range(P):
            ek, ec = orc.count_kmer(streams[p], k, hard_min)
            assert got[p][0].shape == (len(ec), kw)
            assert np.array_equal(ek, got[p][0]) and np.array_equal(ec, got[p][1]), (k, p, hard_min)
            vals = [sum(int(ek[i, w]) << (64 * w) for w in range(kw)) for i in range(len(ec))]
            exp = sorted((v, n) for v, n in truth[p].items() if n >= hard_min)
            assert vals == [v for v, _ in exp] and list(map(int, ec)) == [n for _, n in exp]
            eh, ehc = orc.count_hash(streams[p], k, 1000003, [7, 0, 3, 1, 12][p], hard_min)
            assert np.array_equal(eh, goth[p][0]) and np.array_equal(ehc, goth[p][1]), (k, p, hard_min)
    keys, counts = ctx.count_kmer(streams[0], k, 1)
    ek, ec = orc.count_kmer(streams[0], k, 1)
    assert np.array_equal(keys, ek) and np.array_equal(counts, ec)      # (kmx_count_kmer: one stream)
    # the samples' lists of "one partition", merged: the matrix the oracle builds from the oracle's lists
    lists = [ctx.count_batch([s], k, 1)[0] for s in streams]
    exp_body, exp_rows, exp_stats = orc.merge_matrix([(a.reshape(-1), c) for a, c in lists], kw, [1, 2, 1, 1, 2], 1, 0, orc.MODE_COUNT)
    body, rows, stats = ctx.merge(lists, kw, [1, 2, 1, 1, 2], 1, 0, orc.MODE_COUNT)
    assert rows == exp_rows and body == exp_body and np.array_equal(stats, exp_stats)


def test_count_rejects_records_longer_than_a_super_kmer(ctx):
    """a record that claims more k-mers than a super-k-mer can hold (28 for k < 32, 60 above: Sequence2SuperKmer.hpp:90-132) is a
    malformed stream -- refused, not decoded into wrong keys (the lane-per-k-mer decode reads a record's k-mers from one window)"""
    from kmtricks_amd import lib
    for k, n in ((31, 29), (31, 255), (40, 61), (64, 93), (95, 93), (96, 125), (127, 255)):
        rec = bytes([n]) + bytes((k + n - 1 + 3) // 4)
        with pytest.raises(lib.KmxError, match="malformed super-k-mer stream"):
            ctx.count_kmer(rec, k, 1)
    ok = bytes([28]) + bytes((31 + 28 - 1 + 3) // 4)      # (28 k-mers of poly-A: one key counted 28 times)
    keys, counts = ctx.count_kmer(ok, 31, 1)
    assert list(counts) == [28]


def test_count_empty_stream(ctx):
    k_, c_ = ctx.count_kmer(b"", 31, 2)
    assert len(c_) == 0
    h_, c_ = ctx.count_hash(b"", 31, 6400, 1, 2)
    assert len(c_) == 0


@pytest.mark.parametrize("nr,nc", [(8, 8), (16, 8), (24, 40), (64, 64), (64, 128), (200, 72), (3136, 104), (4096, 1000 // 8 * 8 + 8),
                                   (520, 2504), (1000, 24), (19208, 2504), (1032, 1032), (8, 4104)])
def test_transpose_vs_oracle(ctx, nr, nc):
    rng = np.random.default_rng(nr * 131 + nc)
    m = rng.integers(0, 256, nr * nc // 8, dtype=np.uint8)
    t = ctx.transpose_bits(m, nr, nc)
    assert np.array_equal(t, orc.transpose_bits(m, nr, nc))
    assert np.array_equal(ctx.transpose_bits(t, nc, nr), m)   # bit_matrix_test.cpp:60-99


def test_superk_partition_reference_goldens(ctx):
    """tests/task_main.cpp:85-114 through the HIP partitioner: 37/46/12/43 and 20/21/58/39 k-mers per
    super-k-mer file, and byte-identical record streams vs the oracle (fixture repartition, k=31, m=10)"""
    lut = orc.minimizer_lut(10)
    rep = repart_table()
    for name, f in (("D1", "1.fasta"), ("D2", "2.fasta")):
        reads = read_fasta(os.path.join(GD, f))
        exp = orc.superk_partition(reads, 31, 10, lut, rep, 4)
        got = ctx.superk_partition(reads, 31, 10, rep, 4)
        assert [g[1] for g in got] == G["task_main"]["superk_info_" + name][1::2]
        for p in range(4):
            assert got[p][0] == exp[p][0] and got[p][1] == exp[p][1]


@pytest.mark.parametrize("k,m,P", [(31, 10, 8), (21, 8, 5), (32, 10, 16), (47, 11, 3), (63, 10, 32), (20, 7, 4), (12, 4, 2),
                                   (64, 10, 8), (64, 15, 4), (77, 11, 5), (95, 10, 16), (96, 4, 3), (111, 12, 8), (127, 10, 32), (127, 4, 2)])
def test_superk_statistics_vs_oracle(ctx, k, m, P):
    """PartiInfo<5> from the HIP split (fill_partitions.hpp:67-102): kx-mer / radix counters per partition, super-k-mers,
    k-mers and kx-mers per minimizer; long reads (strand runs and kx-mers that straddle the 64-position chunks),
    palindromes, N's; the statistics-only pass (the repartition's sampling) gives the same per-minimizer numbers"""
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(70 + k, 300, 150, n_rate=0.004) + random_reads(71 + k, 20, 3000, n_rate=0.001) + \
        ["ACGT" * 70, "A" * 300, "ACGTN" * 40, "ACG", "", "T" * k, "acgtacgtnnacgt" * 12, "AT" * 200, "GAATTC" * 50]
    epin, ems, emk, emx = orc.superk_stats(reads, k, m, lut, rep, P)
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    got, pin, ms, mk, mx = ctx.superk_partition_stats(reads, k, m, rep, P)
    for p in range(P):
        assert got[p][0] == exp[p][0] and got[p][1] == exp[p][1]
    assert np.array_equal(pin, epin), np.nonzero(pin != epin)
    assert np.array_equal(ms, ems) and np.array_equal(mk, emk) and np.array_equal(mx, emx)
    assert int(pin[:, 0].sum()) == sum(g[1] for g in got) == int(mk.sum())
    none, pin2, ms2, mk2, mx2 = ctx.superk_partition_stats(reads, k, m, np.zeros(4 ** m, np.uint16), 1, streams=False)
    assert none is None and np.array_equal(ms2, ems) and np.array_equal(mk2, emk) and np.array_equal(mx2, emx)


@pytest.mark.parametrize("path,k", [("sample-sort", 31), ("library", 31), ("overflow", 31), ("lds-hash", 31), ("lds-sort", 31),
                                    ("sample-sort", 47), ("overflow", 47), ("lds-sort", 47), ("library", 63),
                                    ("sample-sort", 80), ("library", 80), ("overflow", 80), ("sample-sort", 111), ("overflow", 127)])
def test_count_sort_paths(ctx, monkeypatch, path, k):
    """the partition-local sample sort (count_sort.hpp) with its bucket kernels -- a wave per bucket with the keys in registers
    (k_cs_wave_sort, buckets of up to 1024 keys; the LDS kernels behind it for the larger ones), the LDS kernels alone
    (KMX_COUNT_BUCKETS=hash|sort) --, the library sort it falls back to (forced, and taken by itself when a k-mer repeated thousands
    of times overflows a bucket): same counts, 64- and 128-bit keys, large enough for hundreds of buckets per partition.  Keys of three and
    four words (k = 80, 111, 127; round 5) take the same sample sort with 24- / 32-byte keys, and word-by-word radix passes behind it"""
    if path == "library":
        monkeypatch.setenv("KMX_COUNT_SORT", "library")
    if path.startswith("lds-"):
        monkeypatch.setenv("KMX_COUNT_BUCKETS", path[4:])
    m, P = 10, 4
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(4242, 1500, 150, n_rate=0.002) * 3
    if path == "overflow":
        rep_read = "ACGTTGCAAGGCTTAAGCCGATTACAGGCTAAGCTTAGGCATCGGATTCAGCATTGCAAGTCCAGTTAGCAGGATCA"
        reads = reads + [rep_read if k < 64 else rep_read + "GGATTCAAGCTTAGCCATGCAATGCCGGATAGCTTAAGCGCTAGCATTACGGATCCAGTAGCTAGGCTAATC"] * 6000
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    got = ctx.count_batch([e[0] for e in exp], k, 2)
    tot = 0
    for p in range(P):
        ek, ec = orc.count_kmer(exp[p][0], k, 2)
        assert np.array_equal(got[p][0], ek) and np.array_equal(got[p][1], ec)
        tot += len(ec)
    assert tot > (150_000 if k == 31 else 80_000 if k <= 80 else 40_000 if k <= 111 else 20_000)
    if path == "overflow":
        assert max(int(g[1].max()) for g in got if len(g[1])) >= 6000
    goth = ctx.count_batch([e[0] for e in exp], k, 1, window=100003, partitions=[7, 3, 0, 9])
    for p, wid in enumerate([7, 3, 0, 9]):
        ek, ec = orc.count_hash(exp[p][0], k, 100003, wid, 1)
        assert np.array_equal(goth[p][0], ek) and np.array_equal(goth[p][1], ec)


def _same_hist(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ("unique", "total", "oob", "sums"))


@pytest.mark.parametrize("path", ["sample-sort", "library", "overflow"])
def test_abundance_histogram_vs_oracle(ctx, monkeypatch, path):
    """--hist: KHist (histogram.hpp:48-68) over every distinct k-mer / hash of a sample BEFORE the hard-min filter
    (count_processor.hpp:61, 135), accumulated on the device by each count call while the histogram is on: per-partition
    calls, the batched call (both sorts; 'overflow' has a k-mer counted 6000 times -> the upper out-of-bounds fields) and
    the fused kmx_count_reads; bounds (1, 255) as the reference builds it and a narrow (2, 10)."""
    if path == "library":
        monkeypatch.setenv("KMX_COUNT_SORT", "library")
    k, m, P = 31, 10, 4
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(99, 700, 150, n_rate=0.002) * 3 + random_reads(98, 300, 150)
    if path == "overflow":
        reads = reads + ["ACGTTGCAAGGCTTAAGCCGATTACAGGCTAAGCTTAGGCATCG"] * 6000
    streams = [e[0] for e in orc.superk_partition(reads, k, m, lut, rep, P)]
    W = 100003
    for hashed in (False, True):
        exp, exp_n = None, None
        for p in range(P):
            c = orc.count_hash(streams[p], k, W, p, 1)[1] if hashed else orc.count_kmer(streams[p], k, 1)[1]
            exp = orc.khist(c, 1, 255, acc=exp); exp_n = orc.khist(c, 2, 10, acc=exp_n)
        assert int(exp["sums"][0]) > 50_000
        if path == "overflow" and not hashed:
            assert int(exp["oob"][1]) >= 1 and int(exp["oob"][3]) >= 6000
        # batched call (hard-min 3 on purpose: the histogram sees the k-mers the filter drops)
        ctx.hist_reset()
        ctx.count_batch(streams, k, 3, window=W if hashed else 0)
        assert _same_hist(ctx.hist_read(1, 255), exp) and _same_hist(ctx.hist_read(2, 10), exp_n)
        # per-partition calls add up
        ctx.hist_reset()
        for p in range(P):
            if hashed: ctx.count_hash(streams[p], k, W, p, 2)
            else: ctx.count_kmer(streams[p], k, 2)
        assert _same_hist(ctx.hist_read(1, 255), exp)
        # fused split + count
        ctx.hist_reset()
        ctx.count_reads(reads, k, m, rep, P, 2, window=W if hashed else 0)
        assert _same_hist(ctx.hist_read(1, 255), exp)
        # off: later calls leave it alone
        ctx.hist_off()
        ctx.count_batch(streams, k, 1, window=W if hashed else 0)
        assert _same_hist(ctx.hist_read(1, 255), exp)
    with pytest.raises(Exception):
        ctx.hist_read(1, 256)


@pytest.mark.parametrize("k,m,P,hard_min,hashed", [(31, 10, 8, 1, False), (31, 10, 8, 2, True), (63, 10, 32, 2, False), (21, 8, 5, 3, False), (32, 10, 16, 1, True),
                                                   (64, 10, 8, 1, False), (95, 11, 5, 2, True), (96, 10, 16, 2, False), (127, 12, 8, 1, True), (127, 10, 4, 1, False)])
def test_count_reads_fused_vs_oracle(ctx, k, m, P, hard_min, hashed):
    """kmx_count_reads (split + count with the streams resident in HBM) == oracle split, then oracle count of every partition;
    the streams it can hand back are the split's, the info numbers those of the skp block framing"""
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(700 + k, 400, 150, n_rate=0.003) * 2 + ["ACGT" * 70, "A" * 300, "", "T" * k, "acgtacgtnnacgt" * 12]
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    W = 6400
    got, nk, streams, info = ctx.count_reads(reads, k, m, rep, P, hard_min, window=W if hashed else 0, streams=True)
    for p in range(P):
        assert streams[p] == exp[p][0] and nk[p] == exp[p][1]
        if hashed:
            ek, ec = orc.count_hash(exp[p][0], k, W, p, hard_min)
            assert np.array_equal(got[p][0], ek) and np.array_equal(got[p][1], ec)
        else:
            ek, ec = orc.count_kmer(exp[p][0], k, hard_min)
            assert np.array_equal(got[p][0], ek) and np.array_equal(got[p][1], ec)
        # block framing of the stream (<= 32768-byte blocks): k-mers since the last flush, bytes flushed
        buf = km = fl = pos = 0
        s = exp[p][0]
        while pos < len(s):
            n = s[pos]; nb = 1 + (k + n - 1 + 3) // 4
            if buf + nb > 32768: fl += buf + 4; buf = 0; km = 0
            buf += nb; km += n; pos += nb
        assert (int(info[p, 0]), int(info[p, 1])) == (km, fl)
    got2, nk2, none, _ = ctx.count_reads(reads, k, m, rep, P, hard_min, window=W if hashed else 0)
    assert none is None and nk2 == nk and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(got, got2))


@pytest.mark.parametrize("passes", ["one", "two", "long-read"])
@pytest.mark.parametrize("k,m,P", [(31, 10, 8), (21, 8, 5), (32, 10, 16), (47, 11, 3), (63, 10, 32), (20, 7, 4),
                                   (64, 10, 8), (65, 15, 3), (90, 9, 16), (96, 10, 8), (113, 13, 5), (127, 10, 32)])
def test_superk_partition_random_reads_vs_oracle(ctx, monkeypatch, k, m, P, passes):
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(7 + k, 900, 150, n_rate=0.004) + ["ACGT" * 70, "A" * 300, "ACGTN" * 40, "ACG", "", "T" * k,
                                                        "acgtacgtnnacgt" * 12]
    # the descriptors are placed by count + scan + emit passes over the reads, or (KMX_SUPERK_ONE_PASS) in one pass with a look-back
    # over the workgroups when every read's descriptors fit the LDS (a read of more than 512 k-mers: the passes): the same streams
    if passes != "two":
        monkeypatch.setenv("KMX_SUPERK_ONE_PASS", "1")
    if passes == "long-read":
        reads = reads[:300] + [("ACGTTGCATGGA" * 200)[:2000 + k]] + reads[300:]
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    got = ctx.superk_partition(reads, k, m, rep, P)
    assert sum(g[1] for g in got) > 10000
    for p in range(P):
        assert got[p][1] == exp[p][1]
        assert got[p][0] == exp[p][0]


@pytest.mark.parametrize("stats_by", ["partition", "atomics", "two-walks"])
@pytest.mark.parametrize("k,m,P,hard_min,hashed,G", [(31, 10, 8, 1, False, 1), (31, 10, 8, 2, True, 2), (63, 10, 32, 2, False, 3), (21, 8, 5, 3, False, 2), (32, 10, 16, 1, True, 1),
                                                      (31, 10, 2, 1, False, 2), (40, 12, 3, 1, False, 2),
                                                      (64, 10, 8, 2, False, 3), (80, 10, 8, 1, False, 2), (96, 10, 4, 1, True, 2), (97, 11, 3, 1, False, 1), (127, 12, 5, 2, False, 1)])
def test_count_reads_dev_vs_oracle(ctx, k, m, P, hard_min, hashed, G, stats_by, monkeypatch):
    """kmx_count_reads_dev: the counts stay in HBM as packed .kmer-body records, partition p in store p % G (what the merge stage
    of GPU p % G reads); read back they are the oracle's counts, merged where they lie (kmx_merge_dev) they give the oracle's
    matrix; the raw PartiInfo<5> tables are the oracle's counters"""
    from kmtricks_amd import lib
    # (the sparse form of the tables, G >= 2: counted per partition from the sorted descriptors -- k_part_stats; with few partitions a
    #  partition's minimizers do not fit its LDS table and the rest goes through the dense tables --, or, KMX_STATS_ATOMICS, by atomics while
    #  the reads are walked, as the dense form always is)
    if stats_by == "atomics":
        if G < 2: pytest.skip("the dense form is counted by atomics anyway")
        monkeypatch.setenv("KMX_STATS_ATOMICS", "1")
    # (round 6: k < 64 with the statistics per partition takes the sync-free path -- one walk, own counting sort of the descriptors, the
    #  sample sort's tables made on the device; "two-walks" = the path of rounds 1-5 it falls back to, KMX_COUNT_FAST=0)
    if stats_by == "two-walks":
        if k >= 64: pytest.skip("k >= 64 takes the two-walk path anyway")
        monkeypatch.setenv("KMX_COUNT_FAST", "0")
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    reads = random_reads(900 + k, 400, 150, n_rate=0.003) * 2 + ["ACGT" * 70, "A" * 300, "", "T" * k, "acgtacgtnnacgt" * 12]
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    epin, ems, emk, _ = orc.superk_stats(reads, k, m, lut, rep, P)
    W = 6400
    kw = 1 if hashed else (k + 31) // 32
    stores = [lib.Store(0) for _ in range(G)]
    try:
        # (G == 3 / the 12-mer minimizers: the bases sent ahead of the call -- kmx_reads_upload --, the call given the device pointer)
        lists, nk, raw = ctx.count_reads_dev(reads, k, m, rep, P, hard_min, stores, window=W if hashed else 0, raw=True, sparse=(G >= 2), ahead=(G == 3 or m == 12))
        # a second sample (the reads reversed) in the same stores: lists of both must stay valid
        reads2 = reads[::-1][:300]
        lists2, nk2, _ = ctx.count_reads_dev(reads2, k, m, rep, P, hard_min, stores, window=W if hashed else 0)
        exp2 = orc.superk_partition(reads2, k, m, lut, rep, P)
        assert sum(s.used() for s in stores) >= sum(n for _, n in lists + lists2) * (kw * 8 + 4)
        pr, ms, mk, nsk = raw
        pr = pr.reshape(P, 5, 256).astype(np.uint64)
        assert np.array_equal(pr.reshape(P, 1280), epin[:, 2:]) and np.array_equal(pr.sum(axis=(1, 2)), epin[:, 1])
        assert np.array_equal((pr.sum(axis=2) * np.arange(1, 6, dtype=np.uint64)).sum(axis=1), epin[:, 0])
        assert np.array_equal(ms, ems) and np.array_equal(mk, emk) and nsk == int(ems.sum())
        for (ll, nn, ee) in ((lists, nk, exp), (lists2, nk2, exp2)):
            for p in range(P):
                assert nn[p] == ee[p][1]
                gk, gc = ctx.read_list(ll[p][0], ll[p][1], kw)
                ek, ec = orc.count_hash(ee[p][0], k, W, p, hard_min) if hashed else orc.count_kmer(ee[p][0], k, hard_min)
                assert np.array_equal(gk.reshape(ek.shape), ek) and np.array_equal(gc, ec)
        # merge straight from the stores: two samples per partition
        for p in range(P):
            e1 = orc.count_hash(exp[p][0], k, W, p, hard_min) if hashed else orc.count_kmer(exp[p][0], k, hard_min)
            e2 = orc.count_hash(exp2[p][0], k, W, p, hard_min) if hashed else orc.count_kmer(exp2[p][0], k, hard_min)
            res = ctx.merge_dev([dict(lists=[lists[p], lists2[p]], key_words=kw, soft_min=[1, 1], rec_min=1, share_min=0, mode=lib.MODE_COUNT)])
            res.wait()
            eb, er, es = orc.merge_matrix([(e1[0].reshape(-1), e1[1]), (e2[0].reshape(-1), e2[1])], kw, [1, 1], 1, 0, orc.MODE_COUNT)
            assert res.rows() == er and res.body() == eb and np.array_equal(res.stats(), es)
            res.free()
    finally:
        for s in stores:
            s.close()


def test_count_store_limit(ctx):
    """a store refuses what does not fit its limit (the pipeline then writes that sample's count files instead)"""
    from kmtricks_amd import lib
    rep = orc.repart_static(10, 4)
    reads = random_reads(5, 300, 150)
    st = lib.Store(0, limit_bytes=4096)
    try:
        assert st.limit() == 4096
        with pytest.raises(lib.KmxError, match="store is full"):
            ctx.count_reads_dev(reads * 2, 31, 10, rep, 4, 1, [st])
    finally:
        st.close()


def test_merge_host_mixed_lists(ctx):
    """kmx_merge_host with list_on_device: resident lists (a store) and host lists (count files just read) in one task"""
    import ctypes as C
    from kmtricks_amd import lib
    lut = orc.minimizer_lut(10)
    rep = orc.repart_static(10, 4)
    r1, r2, r3 = random_reads(11, 300, 150), random_reads(12, 300, 150), random_reads(11, 200, 150)
    st = lib.Store(0)
    try:
        l1, _, _ = ctx.count_reads_dev(r1, 31, 10, rep, 4, 1, [st])
        l3, _, _ = ctx.count_reads_dev(r3, 31, 10, rep, 4, 1, [st])
        e = [[orc.count_kmer(s[0], 31, 1) for s in orc.superk_partition(r, 31, 10, lut, rep, 4)] for r in (r1, r2, r3)]
        for p in range(4):
            host = lib.pack_records(e[1][p][0], e[1][p][1], 1)
            keep = []
            t = ctx._task([l1[p], (host.ctypes.data if len(host) else None, len(host)), l3[p]], 1, [1, 1, 1], 2, 0, lib.MODE_COUNT, 0, 0, 2, 0, keep)
            flags = (C.c_uint8 * 3)(1, 0, 1)
            t.list_on_device = C.cast(flags, C.c_void_p)
            arr = (lib.KmxMergeTask * 1)(t)
            res = ctx.merge_host((arr, 1, [3], keep))
            res.wait()
            eb, er, es = orc.merge_matrix([(x[0].reshape(-1), x[1]) for x in (e[0][p], e[1][p], e[2][p])], 1, [1, 1, 1], 2, 0, orc.MODE_COUNT)
            assert res.rows() == er and res.body() == eb and np.array_equal(res.stats(), es)
            res.free()
    finally:
        st.close()


def test_superk_sample_vs_oracle(ctx):
    """kmx_superk_sample (the sampling pass of the sampled repartition, gatb RepartitionAlgorithm.cpp:182-215): the shortest
    prefix of the reads that holds more than the budget of super-k-mers, its kx-mers per minimizer -> the oracle's table"""
    k, m, P = 31, 10, 8
    lut = orc.minimizer_lut(m)
    rep0 = np.zeros(4 ** m, np.uint16)
    reads = random_reads(31, 500, 150, n_rate=0.002)
    per_read = [sum(x[2] for x in orc.superk_partition([r], k, m, lut, rep0, 1)) for r in reads]
    for budget in (10, 1000, 10 ** 9):
        acc = used = 0
        while used < len(reads):
            acc += per_read[used]; used += 1
            if acc > budget: break
        g_used, g_nsk, g_mx = ctx.superk_sample(reads, k, m, budget)
        assert (g_used, g_nsk) == (used, acc)
        _, _, _, emx = orc.superk_stats(reads[:used], k, m, lut, rep0, 1)
        assert np.array_equal(g_mx, emx)
        assert np.array_equal(orc.repart_sampled(g_mx, P), orc.repart_sampled(emx, P))


@pytest.mark.parametrize("k,m,P,hard_min,hashed,G,S", [(31, 10, 8, 1, False, 1, 3), (31, 10, 16, 2, True, 3, 4), (63, 10, 32, 2, False, 2, 2), (21, 8, 5, 1, False, 2, 5)])
def test_count_reads_dev_multi_vs_single(ctx, k, m, P, hard_min, hashed, G, S):
    """kmx_count_reads_dev_multi: several samples in ONE call (partition id' = sample * P + partition inside, the statistics
    tables per sample) give, sample by sample, the oracle's counts, k-mer numbers, SuperKmerBinInfoFile numbers and PartiInfo<5>
    tables -- what S calls of kmx_count_reads_dev give; samples of different sizes, an empty one among them"""
    from kmtricks_amd import lib
    lut = orc.minimizer_lut(m)
    rep = orc.repart_static(m, P)
    W = 6400
    samples = [random_reads(3000 + 17 * i + k, 150 + 120 * i, 150, n_rate=0.003) + ["ACGT" * 70, "T" * k, ""] for i in range(S)]
    samples[S // 2] = [] if S > 2 else samples[S // 2]
    kw = 1 if hashed else (k + 31) // 32
    stores = [lib.Store(0, limit_bytes=64 << 20) for _ in range(G)]
    try:
        got = ctx.count_reads_dev_multi(samples, k, m, rep, P, hard_min, stores, window=W if hashed else 0, raw=True)
        assert len(got) == S
        for i, reads in enumerate(samples):
            lists, nk, info, raw = got[i]
            exp = orc.superk_partition(reads, k, m, lut, rep, P)
            epin, ems, emk, _ = orc.superk_stats(reads, k, m, lut, rep, P)
            pr, ms, mk, nsk = raw
            assert np.array_equal(pr.reshape(P, 1280).astype(np.uint64), epin[:, 2:]) and np.array_equal(ms, ems) and np.array_equal(mk, emk) and nsk == int(ems.sum())
            for p in range(P):
                assert nk[p] == exp[p][1]
                gk, gc = ctx.read_list(lists[p][0], lists[p][1], kw)
                ek, ec = orc.count_hash(exp[p][0], k, W, p, hard_min) if hashed else orc.count_kmer(exp[p][0], k, hard_min)
                assert np.array_equal(gk.reshape(ek.shape), ek) and np.array_equal(gc, ec)
                # block framing of the stream (<= 32768-byte blocks): k-mers since the last flush, bytes flushed
                buf = km = fl = pos = 0
                s = exp[p][0]
                while pos < len(s):
                    n = s[pos]; nb = 1 + (k + n - 1 + 3) // 4
                    if buf + nb > 32768: fl += buf + 4; buf = 0; km = 0
                    buf += nb; km += n; pos += nb
                assert (int(info[p, 0]), int(info[p, 1])) == (km, fl)
    finally:
        for s in stores: s.close()


@pytest.mark.parametrize("k,m,P", [(31, 10, 4), (47, 9, 4), (63, 10, 4), (64, 10, 3), (96, 11, 4), (127, 10, 2)])
def test_parti_info_through_hip_against_the_string_restatement(ctx, k, m, P):
    """the PartiInfo<5> counters of the HIP split against tests/test_merge_independent.py's `pinfo_from_strings` (super-k-mers and kx-mers
    cut out of the reads as strings, fill_partitions.hpp:59-105): the reference's tests never read PartiInfoFile, this is the pin"""
    from test_merge_independent import pinfo_from_strings
    rep = orc.repart_static(m, P)
    reads = random_reads(900 + k, 60, 220, n_rate=0.004) + ["A" * 200, "ACGTN" * 40, "GAATTC" * 50, "acgtacgtnnacgt" * 12]
    exp, minim = pinfo_from_strings(reads, k, m, rep, P)
    _, pin, ms, mk, _ = ctx.superk_partition_stats(reads, k, m, rep, P)
    assert pin.tolist() == exp
    assert {int(v): [int(ms[v]), int(mk[v])] for v in np.nonzero(ms)[0]} == minim


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["more records than estimated", "a partition beyond the sample sort"])
def test_count_reads_dev_hands_back_to_the_two_walk_path(ctx, what, monkeypatch, capfd):
    """round 6: the sync-free path of kmx_count_reads_dev raises a status word on the device and the call is repeated the old way --
    k = 12 with 11-mer minimizers cuts a super-k-mer per k-mer (more records than the sorted arrays are sized for); one partition of a
    million k-mers is beyond the sample sort's 2048 buckets.  (scripts/fuzz_count.py found the first: the statistics kernel, on the
    second stream, read the sorted arrays the scatter had not written -- a memory fault; it leaves at once now.)  Counts, statistics and
    the KMX_TRACE line that says so."""
    from kmtricks_amd import lib
    if what == "more records than estimated":
        k, m, P, reads = 12, 11, 3, random_reads(4242, 300, 2000, n_rate=0.001)
    else:
        k, m, P, reads = 31, 10, 1, random_reads(4243, 600, 2000, n_rate=0.0)
    lut, rep = orc.minimizer_lut(m), orc.repart_static(m, P)
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    epin, ems, emk, _ = orc.superk_stats(reads, k, m, lut, rep, P)
    monkeypatch.setenv("KMX_TRACE", "1")
    store = lib.Store(0)
    try:
        for _ in range(2):      # (twice: the second call meets the first one's estimates and pool)
            lists, nk, raw = ctx.count_reads_dev(reads, k, m, rep, P, 1, [store], raw=True, sparse=True, ahead=True)
            pr, ms, mk, nsk = raw
            assert np.array_equal(pr.reshape(P, 1280).astype(np.uint64), epin[:, 2:]) and np.array_equal(ms, ems) and np.array_equal(mk, emk) and nsk == int(ems.sum())
            for p in range(P):
                ek, ec = orc.count_kmer(exp[p][0], k, 1)
                gk, gc = ctx.read_list(lists[p][0], lists[p][1], 1)
                assert nk[p] == exp[p][1] and np.array_equal(gk.reshape(ek.shape), ek) and np.array_equal(gc, ec)
        assert capfd.readouterr().err.count("the sync-free path handed the call back") == 2
    finally:
        store.close()


@pytest.mark.gpu
def test_count_calls_side_by_side_into_one_store():
    """round 6: `kmx pipeline`'s workers -- a context and a host thread each -- count their samples side by side into the GPU's one store;
    each call reserves the tail of a chunk of its own for the lists it has not sized yet (kmx_store::try_reserve, several open at a
    time).  Four threads, six samples each, every list against the oracle; then the store still takes a plain allocation."""
    import threading
    from kmtricks_amd import lib
    k, m, P = 31, 10, 8
    lut, rep = orc.minimizer_lut(m), orc.repart_static(m, P)
    samples = [random_reads(900 + i, 260 + 7 * i, 300, n_rate=0.002) for i in range(6)]
    expected = []
    for reads in samples:
        parts = orc.superk_partition(reads, k, m, lut, rep, P)
        expected.append([orc.count_kmer(parts[p][0], k, 1) for p in range(P)])
    store = lib.Store(0)
    errors, results = [], {}

    def worker(w):
        try:
            c = lib.Context(0)
            for rnd in range(3):
                for i, reads in enumerate(samples):
                    lists, nk, _ = c.count_reads_dev(reads, k, m, rep, P, 1, [store])
                    results[(w, rnd, i)] = [c.read_list(lists[p][0], lists[p][1], 1) for p in range(P)]
            c.close()
        except Exception as e:      # (an assertion in a thread is lost: collected and raised below)
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    assert len(results) == 4 * 3 * len(samples)
    for (w, rnd, i), got in results.items():
        for p in range(P):
            ek, ec = expected[i][p]
            assert np.array_equal(got[p][0].reshape(ek.shape), ek) and np.array_equal(got[p][1], ec), (w, rnd, i, p)
    assert store.used() > 0
    store.close()


@pytest.mark.gpu
@pytest.mark.parametrize("first", ["1", "0"])
@pytest.mark.parametrize("k,m,P,hard_min,hashed,copies", [(31, 10, 2, 2, False, 6), (31, 10, 2, 1, False, 1), (32, 10, 3, 3, True, 5), (21, 8, 2, 1, False, 12), (25, 9, 2, 2, False, 2)])
def test_count_buckets_by_counting_first(ctx, k, m, P, hard_min, hashed, copies, first, monkeypatch, capfd):
    """round 6: k_cs_wave_count -- a bucket's keys into a hash table of the wave's own, the kept distinct keys sorted as (key, count) pairs --
    against the full sort of the same buckets (KMX_COUNT_HASH_FIRST=0) and the oracle: reads given `copies` times over (sequencing depth: the
    table holds a bucket's distinct keys), once (it does not: the bucket is sorted in the same kernel or listed for the LDS kernels), hard-min
    1 .. 3, k-mers and hashes, partitions of a few hundred thousand k-mers = hundreds of buckets each"""
    from kmtricks_amd import lib
    monkeypatch.setenv("KMX_COUNT_HASH_FIRST", first)
    monkeypatch.setenv("KMX_TRACE", "1")
    lut, rep = orc.minimizer_lut(m), orc.repart_static(m, P)
    base = random_reads(7000 + k + copies, 900 // max(1, copies // 2), 700, n_rate=0.001)
    reads = []
    for c in range(copies):      # (every copy's reads cut at other places: the same k-mers, other super-k-mers)
        reads += [r[(37 * c) % 90:] for r in base[: len(base) - 11 * c]]
    W = 1 << 22
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    store = lib.Store(0)
    try:
        for _ in range(2):
            lists, nk, _ = ctx.count_reads_dev(reads, k, m, rep, P, hard_min, [store], window=W if hashed else 0)
            for p in range(P):
                ek, ec = (orc.count_hash(exp[p][0], k, W, p, hard_min) if hashed else orc.count_kmer(exp[p][0], k, hard_min))
                gk, gc = ctx.read_list(lists[p][0], lists[p][1], 1)
                assert nk[p] == exp[p][1] and np.array_equal(gk.reshape(ek.shape), ek) and np.array_equal(gc, ec)
        err = capfd.readouterr().err
        assert "count_reads_fast" in err and "handed the call back" not in err      # (the sync-free path took both calls: its kernels are the ones compared)
    finally:
        store.close()


@pytest.mark.gpu
def test_counting_first_gives_way_to_the_sort_and_comes_back(monkeypatch, capfd):
    """round 6: a sample without repeats overflows the waves' hash tables (k_cs_wave_count sorts or lists those buckets and counts them);
    the context's next calls take k_cs_wave_sort, and try counting first again a dozen samples on.  The counts are the oracle's either way."""
    from kmtricks_amd import lib
    monkeypatch.delenv("KMX_COUNT_HASH_FIRST", raising=False)
    monkeypatch.setenv("KMX_TRACE", "1")
    k, m, P = 31, 10, 2
    lut, rep = orc.minimizer_lut(m), orc.repart_static(m, P)
    reads = random_reads(8111, 700, 900, n_rate=0.0)      # (630 k distinct k-mers, two partitions: ~750 buckets each, none with a repeat)
    exp = orc.superk_partition(reads, k, m, lut, rep, P)
    c = lib.Context(0); store = lib.Store(0)
    try:
        kinds = []
        for call in range(16):
            lists, nk, _ = c.count_reads_dev(reads, k, m, rep, P, 1, [store])
            err = capfd.readouterr().err
            assert "handed the call back" not in err
            kinds.append("count" if "by counting first" in err else "sort")
            if call in (0, 1, 15):
                for p in range(P):
                    ek, ec = orc.count_kmer(exp[p][0], k, 1)
                    gk, gc = c.read_list(lists[p][0], lists[p][1], 1)
                    assert np.array_equal(gk.reshape(ek.shape), ek) and np.array_equal(gc, ec)
        assert kinds[0] == "count" and kinds[1] == "sort" and "count" in kinds[2:], kinds
    finally:
        store.close(); c.close()
