"""Pins the CPU oracle (oracle/kmx_oracle.c) against the reference's own golden
vectors (tests/golden/, extracted from /root/reference/tests by make_golden.py).
CPU only."""
import json, os, struct
import numpy as np
import pytest

import orc
import kmfiles

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.json")))
GD = os.path.join(os.path.dirname(__file__), "golden")
K, M, P = 31, 10, 4


def repart_table():
    rt = G["repartition_table"]
    t = np.zeros(rt["nb_minims"], dtype=np.uint16)
    for i, v in rt["nonzero"].items():
        t[int(i)] = v
    return t


def read_fasta(path):
    seqs, cur = [], []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if cur: seqs.append("".join(cur)); cur = []
        elif line:
            cur.append(line)
    if cur: seqs.append("".join(cur))
    return seqs


@pytest.fixture(scope="module")
def superk():
    lut = orc.minimizer_lut(M)
    rep = repart_table()
    out = {}
    for name, f in (("D1", "1.fasta"), ("D2", "2.fasta")):
        out[name] = orc.superk_partition(read_fasta(os.path.join(GD, f)), K, M, lut, rep, P)
    return out


def test_xxh64_known_answers():
    # XXH64 specification test vectors (xxHash README / sanity tests)
    assert orc.xxh64(b"", 0) == 0xEF46DB3751D8E999
    assert orc.xxh64(b"a", 0) == 0xD24EC4F1A98C6E5B
    assert orc.xxh64(b"abc", 0) == 0x44BC2CF5AD770999
    try:
        import xxhash
    except ImportError:
        return
    rng = np.random.default_rng(1)
    for n in (1, 3, 4, 7, 8, 12, 16, 31, 32, 33, 64, 100):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 2 ** 63 + 5):
            assert orc.xxh64(d, seed) == xxhash.xxh64(d, seed=seed).intdigest()


def test_repartition_test_minimizers():
    """tests/repartition_test.cpp:7-18"""
    lut = orc.minimizer_lut(M)
    rep = repart_table()
    for kmer, part in G["repartition_test"]["cases"]:
        w = orc.kmer_from_string(kmer)
        assert orc.kmer_to_string(w, K) == kmer
        # km::Kmer::minimizer (kmer.hpp:848-886) works on the canonical... the gatb model on the forward value;
        # for these four k-mers both give the asserted partition
        mini = orc.minimizer_of(w, K, M, lut)
        assert rep[mini] == part


def test_superk_info_counts(superk):
    """tests/task_main.cpp:85-114: k-mers per super-k-mer file"""
    for name in ("D1", "D2"):
        exp = G["task_main"]["superk_info_" + name]
        assert exp[0] == P
        got = [superk[name][p][1] for p in range(P)]
        assert got == exp[1::2]


def test_kmer_count_goldens(superk):
    """tests/task_main.cpp:118-340: canonical 31-mers of partition 0 in file order, record counts elsewhere"""
    cf = G["task_main"]["count_files"]
    for name in ("D1", "D2"):
        for p in range(P):
            keys, counts = orc.count_kmer(superk[name][p][0], K, 1)
            exp = cf[f"counts/partition_{p}/{name}.kmer"]
            if "kmers" in exp:
                assert [[orc.kmer_to_string(k, K), int(c)] for k, c in zip(keys, counts)] == exp["kmers"]
            else:
                assert len(counts) == exp["n"]


def test_hash_count_goldens(superk):
    """tests/task_main.cpp:342-508 with tests/data/hash.info (W = 25 000 000)"""
    bloom, nparts, wbits, wbytes, msize = struct.unpack("<QQQQI", open(os.path.join(GD, "hash.info"), "rb").read())
    assert (nparts, wbits, msize) == (4, 25000000, 10)
    cf = G["task_main"]["count_files"]
    for name in ("D1", "D2"):
        for p in range(P):
            hs, counts = orc.count_hash(superk[name][p][0], K, wbits, p, 1)
            exp = cf[f"counts/partition_{p}/{name}.hash"]
            if "hashes" in exp:
                assert [[int(h), int(c)] for h, c in zip(hs, counts)] == exp["hashes"]
            else:
                assert len(counts) == exp["n"]
            assert np.all((hs >= wbits * p) & (hs < wbits * (p + 1)))


def test_count_matches_committed_partition_fixtures(superk):
    """tests/data/partitions/{kmers,hashes}: the reference's committed count files equal what the
    oracle computes from the FASTA fixtures (keys; the fixtures were written by a u8-count build)."""
    for name in ("D1", "D2"):
        for p in range(P):
            f = kmfiles.read_kmer_file(f"{GD}/partitions/kmers/partition_{p}/{name}.kmer")
            keys, counts = orc.count_kmer(superk[name][p][0], K, 1)
            assert f["k"] == K and np.array_equal(f["keys"], keys) and np.array_equal(f["counts"], counts)
            h = kmfiles.read_hash_file(f"{GD}/partitions/hashes/partition_{p}/{name}.hash")
            # the committed .hash fixtures were produced with --bloom-size 1e6: W = round_up64(1e6/4) = 250048
            # (HashWindow, include/kmtricks/hash.hpp:31-40)
            hs, hc = orc.count_hash(superk[name][p][0], K, 250048, p, 1)
            assert np.array_equal(h["keys"], hs) and np.array_equal(h["counts"], hc)


@pytest.mark.parametrize("kind", ["kmers", "hashes"])
def test_merge_test_row_counts(kind):
    """tests/merge_test.cpp:5-78: 57/67/70/82 rows (next() count = distinct keys)"""
    mt = G["merge_test"]
    for p in range(4):
        lists = []
        for s in ("D1", "D2"):
            if kind == "kmers":
                f = kmfiles.read_kmer_file(f"{GD}/partitions/kmers/partition_{p}/{s}.kmer")
            else:
                f = kmfiles.read_hash_file(f"{GD}/partitions/hashes/partition_{p}/{s}.hash")
            lists.append((f["keys"], f["counts"]))
        body, rows, stats = orc.merge_matrix(lists, 1, mt["soft_min"], mt["rec_min"], mt["share_min"], orc.MODE_COUNT)
        assert rows == mt[("kmer" if kind == "kmers" else "hash") + "_rows"][p]
        assert len(body) == rows * (8 + 2 * 4)
        m = np.frombuffer(body, dtype=np.uint8).reshape(rows, 16)
        k = m[:, :8].copy().view(np.uint64).ravel()
        assert np.all(k[1:] > k[:-1])
        # every input record is solid (soft-min 1) -> unique == list length
        assert [int(x) for x in stats[2]] == [len(l[1]) for l in lists]


def test_packc_goldens():
    """tests/packc_test.cpp:5-40"""
    for n, b, e in G["packc_test"]["byte_count_pack"]:
        assert orc.byte_count_pack(n, b) == e
    for c, w, e in G["packc_test"]["to_n_b"]:
        assert orc.to_n_b(c, w) == e


def test_static_repart_matches_xxhash_module():
    xxhash = pytest.importorskip("xxhash")
    t = orc.repart_static(6, 7)
    for m_ in (0, 1, 77, 4095):
        assert t[m_] == xxhash.xxh64(struct.pack("<I", m_), seed=0).intdigest() % 7


def test_transpose_roundtrip():
    """tests/bit_matrix_test.cpp:60-99: transpose(transpose(x)) == x, and the definition out[c][r]=in[r][c]"""
    rng = np.random.default_rng(3)
    for nr, nc in ((8, 8), (16, 8), (24, 40), (64, 128), (200, 72)):
        m = rng.integers(0, 256, nr * nc // 8, dtype=np.uint8)
        t = orc.transpose_bits(m, nr, nc)
        assert np.array_equal(orc.transpose_bits(t, nc, nr), m)
        bits = np.unpackbits(m.reshape(nr, nc // 8), axis=1, bitorder="little")
        tb = np.unpackbits(t.reshape(nc, nr // 8), axis=1, bitorder="little")
        assert np.array_equal(bits.T, tb)


def test_sampled_repartition_reproduces_the_reference_table():
    """gatb's sampled repartition (RepartitionAlgorithm.cpp:182-215 kx-mers per minimizer of the sampled reads, PartiInfo.cpp:48-103
    computeDistrib) on the reference's two test samples (k = 31, m = 10, 4 partitions) == its committed
    tests/data/repart_gatb/repartition.minimRepart, all 4^10 entries"""
    lut = orc.minimizer_lut(10)
    reads = read_fasta(os.path.join(GD, "1.fasta")) + read_fasta(os.path.join(GD, "2.fasta"))
    pin, ms, mk, mx = orc.superk_stats(reads, 31, 10, lut, repart_table(), 4)
    assert int(mk.sum()) == sum(G["task_main"]["superk_info_D1"][1::2]) + sum(G["task_main"]["superk_info_D2"][1::2])
    assert [int(x) for x in pin[:, 0]] == [a + b for a, b in zip(G["task_main"]["superk_info_D1"][1::2], G["task_main"]["superk_info_D2"][1::2])]
    assert np.array_equal(orc.repart_sampled(mx, 4), repart_table())


def test_histogram_reference_vector():
    """tests/histogram_test.cpp: KHist(0, 20, 1, 10) fed {1, 1, 3, 9, 1, 2, 2, 2, 9, 5} -> the unique / total bins it asserts, nothing out of bounds"""
    g = G["histogram_test"]
    h = orc.khist(g["counts"], g["lower"], g["upper"])
    assert h["unique"].tolist() == g["unique"] and h["total"].tolist() == g["total"]
    assert h["oob"].tolist() == [0, 0, 0, 0] and h["sums"].tolist() == [len(g["counts"]), sum(g["counts"])]
    h2 = orc.khist([0, 11, 300], 1, 10, acc=h)      # out of bounds on both sides
    assert h2["oob"].tolist() == [1, 2, 0, 311]


def test_kmer_test_vectors():
    """tests/kmer_test.cpp:71-152 on the oracle's primitives: canonical forms, the most-significant-word-first order, the m-mers and
    the minimizer of a k-mer (kmer.hpp:848-886: smallest valid m-mer value over the k-mer's m-mers and their reverse complements)"""
    Gk = G["kmer_test"]
    for a, c in Gk["canonical"]:
        k = len(a)
        w = orc.kmer_from_string(a); rc = orc.revcomp(w, k)
        less = tuple(int(x) for x in w[::-1]) < tuple(int(x) for x in rc[::-1])
        assert (a if less else orc.kmer_to_string(rc, k)) == c
    for mk, x, y in Gk["less"]:
        if len(x) > 64:
            continue      # (three words: beyond the oracle's one- and two-word keys; the shipped kmer.hpp is checked in test_tools_cpu)
        wx, wy = orc.kmer_from_string(x), orc.kmer_from_string(y)
        assert tuple(int(v) for v in wx[::-1]) < tuple(int(v) for v in wy[::-1])
    mi = Gk["minimizer"]
    s, m = mi["kmer"], mi["m"]
    assert [s[i:i + m] for i in range(len(s) - m + 1)] == mi["mmers"]
    lut = orc.minimizer_lut(m)
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    val = lambda t: sum(code[c] << (2 * (m - 1 - i)) for i, c in enumerate(t))
    best = min(int(lut[val(t)]) for t in mi["mmers"])
    assert best == val(mi["minimizer"])
    assert orc.minimizer_of(orc.kmer_from_string(s), len(s), m, lut) == best


def test_processor_test_hard_min():
    """tests/processor_test.cpp:10-75: records counted 2 and 6 times under abundance-min 3 -> only the second reaches the file"""
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
            keys, counts = orc.count_kmer(sk[0][0], k, amin)
        else:
            keys, counts = orc.count_hash(sk[0][0], k, 1 << 20, 0, amin)
        assert sorted(int(c) for c in counts) == sorted(kept)


@pytest.mark.parametrize("k", [21, 31, 32, 47, 63, 64, 80, 95, 96, 127])
def test_count_oracle_against_an_independent_restatement(k):
    """the oracle's record decode + count (pinned to the reference's vectors for k = 31 above) against counts taken from the strings
    the records were cut from (tests/synth.py: big-integer k-mer values, reverse complement by string) -- the same code path of the
    oracle for every k, so the k <= 63 cases tie the k = 64 ... 127 ones (Kmer<96> / Kmer<128>, no reference vector exists) to it"""
    from synth import synth_superk_stream
    mx = 28 if k < 32 else 60 if k < 64 else 92 if k < 96 else 124
    recs, cnt = synth_superk_stream(k, k, 300, mx)
    kw = orc.kw_of_k(k)
    for hm in (1, 2):
        keys, c = orc.count_kmer(recs, k, hm)
        vals = [sum(int(keys[i, w]) << (64 * w) for w in range(kw)) for i in range(len(c))]
        exp = sorted((v, n) for v, n in cnt.items() if n >= hm)
        assert vals == [v for v, _ in exp] and list(map(int, c)) == [n for _, n in exp]
