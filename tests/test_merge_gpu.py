"""Parity of the HIP merge (libkmx through its C ABI) against the CPU oracle: bit-exact matrix
bodies and merge statistics.  Needs an MI355X: run with -m gpu."""
import json, os
import numpy as np
import pytest

import orc
import kmfiles
from synth import synth_lists, synth_hash_lists

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from kmtricks_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["rows", "pivot", "cols", "cols-arena"])
def merge_kernel(request, monkeypatch, ctx):
    """Every merge test runs once per COUNT/PA kernel: k_merge_rows, k_merge_pivot and k_merge_cols (which hand tasks
    they do not suit -- dissimilar lists, 128-bit keys, share-min, PA rows -- down to the next kernel inside libkmx).
    The column-blocked pair twice: with its rows written at their final place (file order out of the kernels: the
    default) and with the rows where the kernels leave them + a directory ("cols-arena": kmx_set_file_order off)."""
    kern, _, how = request.param.partition("-")
    monkeypatch.setenv("KMX_MERGE_KERNEL", kern)
    monkeypatch.setenv("KMX_FILE_ORDER", "0" if how == "arena" else "1")      # (contexts the tests create themselves)
    ctx.set_file_order(how != "arena")
    return kern


def check(ctx, lists, kw, soft_min, rec_min, share_min, mode, lower=0, upper=0, bitw=2, rows_hint=0):
    exp_body, exp_rows, exp_stats = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], kw, soft_min, rec_min,
                                                     share_min, mode, lower, upper, bitw)
    body, rows, stats = ctx.merge([(k, c) for k, c in lists], kw, soft_min, rec_min, share_min, mode, lower, upper,
                                  bitw, rows_hint)
    assert rows == exp_rows
    assert len(body) == len(exp_body)
    if body != exp_body:
        a = np.frombuffer(body, np.uint8); b = np.frombuffer(exp_body, np.uint8)
        bad = np.nonzero(a != b)[0]
        raise AssertionError(f"body differs at {len(bad)} bytes, first at {bad[:8]}")
    assert np.array_equal(stats, exp_stats), (stats, exp_stats)
    return rows


def fixture_lists(kind, p):
    out = []
    for s in ("D1", "D2"):
        if kind == "kmers":
            f = kmfiles.read_kmer_file(f"{GD}/partitions/kmers/partition_{p}/{s}.kmer")
        else:
            f = kmfiles.read_hash_file(f"{GD}/partitions/hashes/partition_{p}/{s}.hash")
        out.append((f["keys"].reshape(-1, 1), f["counts"]))
    return out


@pytest.mark.parametrize("kind", ["kmers", "hashes"])
def test_reference_fixtures_row_counts(ctx, kind):
    """tests/merge_test.cpp:5-78 on the committed partitions: 57/67/70/82 rows"""
    G = json.load(open(os.path.join(GD, "reference_goldens.json")))["merge_test"]
    for p in range(4):
        lists = fixture_lists(kind, p)
        for mode in (orc.MODE_COUNT, orc.MODE_PA):
            rows = check(ctx, lists, 1, [1, 1], 1, 1, mode)
            assert rows == G["kmer_rows"][p]


def test_reference_fixtures_bf(ctx):
    W = 250048  # window of the committed .hash fixtures (bloom 1e6 / 4 partitions, hash.hpp:31-40)
    for p in range(4):
        lists = fixture_lists("hashes", p)
        check(ctx, lists, 1, [1, 1], 1, 0, orc.MODE_BF, W * p, W * (p + 1) - 1)
        check(ctx, lists, 1, [1, 1], 1, 0, orc.MODE_BFC, W * p, W * (p + 1) - 1, bitw=2)


CASES = [
    # n_lists, pool, p_present, n_private, soft_min, rec_min, share_min
    (1, 500, 1.0, 0, 1, 1, 0),
    (2, 3000, 0.7, 900, 1, 1, 0),
    (3, 2000, 0.9, 100, 3, 2, 1),
    (17, 1500, 0.8, 50, 2, 1, 2),
    (64, 900, 0.95, 20, 5, 3, 2),
    (100, 700, 0.9, 11, 1, 2, 0),
    (333, 300, 0.97, 7, 10, 50, 20),
    (1000, 120, 0.97, 4, 1, 2, 0),
    (1000, 60, 0.5, 2, 25, 1, 100),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
def test_synthetic_kmer_rows(ctx, case, mode):
    n, pool, pp, npriv, smin, rmin, share = case
    lists = synth_lists(1234 + n, n, pool, pp, npriv, kw=1)
    soft = [smin + (i % 3) for i in range(n)]
    check(ctx, lists, 1, soft, rmin, share, mode)


@pytest.mark.parametrize("case", [(2, 2500, 0.6, 700, 1, 1, 0), (50, 800, 0.9, 30, 4, 2, 3), (700, 150, 0.95, 5, 1, 2, 0)])
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
def test_synthetic_k63_rows(ctx, case, mode):
    """128-bit keys (32 <= k <= 63, Kmer<64>): compare most significant word first"""
    n, pool, pp, npriv, smin, rmin, share = case
    lists = synth_lists(77 + n, n, pool, pp, npriv, kw=2, key_bits=126)
    check(ctx, lists, 2, [smin] * n, rmin, share, mode)


@pytest.mark.parametrize("case", [(1, 400, 1.0, 0, 1, 1, 0), (2, 2500, 0.6, 700, 1, 1, 0), (50, 800, 0.9, 30, 4, 2, 3), (333, 300, 0.97, 7, 10, 50, 20),
                                  (700, 150, 0.95, 5, 1, 2, 0), (2048, 40, 0.9, 2, 3, 1, 100), (4, 20000, 0.3, 6000, 1, 1, 0)])
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
@pytest.mark.parametrize("kw,k", [(3, 65), (3, 96), (4, 97), (4, 127)])
def test_synthetic_wide_key_rows(ctx, merge_kernel, case, mode, kw, k):
    """keys of three and four words (k = 65 ... 96 and 97 ... 127, ceil(k / 32) words: kmer.hpp:215; the reference's default KMER_LIST
    "32 64 96 128", CMakeLists.txt:25-27): k_merge_rows whatever kernel is asked for; rows = the key's words low word first + counts / bits"""
    if merge_kernel != "rows" and k not in (65, 127):
        pytest.skip("wide keys run on k_merge_rows: one forced kernel suffices for the inner sizes")
    n, pool, pp, npriv, smin, rmin, share = case
    lists = synth_lists(31 * kw + n, n, pool, pp, npriv, kw=kw, key_bits=2 * k)
    assert all(l[0].shape[1] == kw for l in lists)
    check(ctx, lists, kw, [smin + (i % 2) for i in range(n)], rmin, share, mode)


@pytest.mark.parametrize("case", [(2500, 60, 0.9, 2, 2, 2, 3), (2049, 30, 0.97, 1, 1, 1, 0), (4096, 12, 0.9, 1, 3, 40, 100)])
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
@pytest.mark.parametrize("kw,k", [(3, 80), (3, 96), (4, 97), (4, 127)])
def test_wide_keys_beyond_2048_lists(ctx, case, mode, kw, k):
    """keys of three and four words (k = 65 ... 127: Kmer<96> / Kmer<128>, the reference's KMER_LIST "32 64 96 128") on cohorts of more than
    2048 samples -- configs[3]'s 2500 at k = 80: the builds of k_merge_rows<3|4> with 4096 / 3072 record slots a tile; rescue and
    recurrence as everywhere"""
    n, pool, pp, npriv, smin, rmin, share = case
    n = min(n, 4096 if kw == 3 else 3072)
    lists = synth_lists(17 * k + n, n, pool, pp, npriv, kw=kw, key_bits=2 * k)
    check(ctx, lists, kw, [smin + (i % 2) for i in range(n)], rmin, share, mode)
    mixed = lists[:300]      # the same context, a task below the limit afterwards: the half-tile build again
    check(ctx, mixed, kw, [smin] * 300, min(rmin, 2), 0, mode)


def test_wide_key_limits_fail_loudly(ctx):
    """4096 lists per task with keys of three words, 3072 with four (the staged keys of 4096 would be 128 of the 160 KB), refused beyond;
    Bloom modes take hash keys"""
    from kmtricks_amd import lib
    lists = [(np.zeros((0, 4), np.uint64), np.zeros(0, np.uint32)) for _ in range(3073)]
    with pytest.raises(lib.KmxError, match="3072 lists"):
        ctx.merge(lists, 4, [1] * 3073, 1, 0, orc.MODE_COUNT)
    lists = [(np.zeros((0, 3), np.uint64), np.zeros(0, np.uint32)) for _ in range(4097)]
    with pytest.raises(lib.KmxError, match="4096 lists"):
        ctx.merge(lists, 3, [1] * 4097, 1, 0, orc.MODE_COUNT)
    with pytest.raises(lib.KmxError, match="key_words"):
        ctx.merge(lists[:2], 5, [1, 1], 1, 0, orc.MODE_COUNT)


def test_first_batches_of_fresh_contexts_in_file_order(merge_kernel):
    """a context's first batches run with arenas it has to guess: tasks overflow them and are flagged WHILE other workgroups of the
    same launch read the flag.  Before round 5's fix (one decision per workgroup, k_merge_cols / k_cols_sparse / k_merge_pivot) threads
    of a workgroup that disagreed about the flag went different ways through its barriers and a first batch in ten took minutes
    (scripts/dev/stress_ord.py at configs[4]'s size); here: several fresh contexts, PA and count rows of 128- and 64-bit keys, rows_hint 1 so
    that the first attempt overflows, every result against the oracle (once per forced kernel: the fixture's environment reaches the contexts made here)"""
    from kmtricks_amd import lib
    lists2 = synth_lists(2024, 300, 4000, 0.9, 400, kw=2, key_bits=126)
    lists1 = synth_lists(2025, 520, 3000, 0.9, 200, kw=1)
    for it in range(16):      # (round 6: 16 fresh contexts -- the stress script needed up to 9 to hit the stall; a context costs ~0.1 s)
        c = lib.Context(0)
        try:
            check(c, lists2, 2, [1] * 300, 1, 0, orc.MODE_PA, rows_hint=1)
            check(c, lists1, 1, [1] * 520, 2, 0, orc.MODE_COUNT, rows_hint=1)
        finally:
            c.close()


def test_first_batches_of_fresh_contexts_bloom_rows(ctx):
    """the same for the Bloom kernels (round 6's audit of the stall class: k_merge_bf / k_merge_bft / k_merge_rows / k_merge_pivot read
    no cross-workgroup word per thread ahead of a barrier -- the error word is read by one thread and handed on through LDS, or through
    __syncthreads_or): a fresh context's first hash:bft batch with the single walk's put-aside scratch at its first size, twice per
    context, 16 contexts"""
    from kmtricks_amd import lib
    rng = np.random.default_rng(77)
    n, W, lower = 300, 40960, 7 * 40960
    lists = []
    for i in range(n):
        hs = np.unique(rng.integers(lower, lower + W, size=900, dtype=np.uint64))
        lists.append((hs, rng.integers(1, 4, size=len(hs)).astype(np.uint32)))
    soft = [2] * n
    exp = orc.merge_matrix(lists, 1, soft, 1, 1, orc.MODE_BFT, lower, lower + W - 1)
    for it in range(16):
        c = lib.Context(0)
        try:
            for _ in range(2):
                body, rows, st = c.merge(lists, 1, soft, 1, 1, orc.MODE_BFT, lower, lower + W - 1)
                assert rows == exp[1] and body == exp[0] and np.array_equal(st, exp[2])
        finally:
            c.close()


def test_sparse_many_distinct_per_tile(ctx):
    """few long lists: thousands of distinct keys per tile (slow ranking path)"""
    lists = synth_lists(5, 4, 20000, 0.3, 6000, kw=1)
    check(ctx, lists, 1, [1] * 4, 1, 0, orc.MODE_COUNT)
    check(ctx, lists, 1, [2] * 4, 2, 1, orc.MODE_PA)


def test_empty_and_ragged(ctx):
    lists = synth_lists(9, 40, 600, 0.8, 10, ragged=True)
    check(ctx, lists, 1, [1] * 40, 1, 0, orc.MODE_COUNT)
    empty = [(np.zeros((0, 1), np.uint64), np.zeros(0, np.uint32)) for _ in range(5)]
    check(ctx, empty, 1, [1] * 5, 1, 0, orc.MODE_COUNT)
    check(ctx, empty, 1, [1] * 5, 1, 0, orc.MODE_BF, 640, 640 + 6399)   # Appendix B-6: W zero rows
    one = [lists[0]] + empty
    check(ctx, one, 1, [1] * 6, 1, 0, orc.MODE_PA)


def test_rows_hint_too_small_retries(ctx):
    lists = synth_lists(11, 8, 100, 0.5, 5000, kw=1)
    check(ctx, lists, 1, [1] * 8, 1, 0, orc.MODE_COUNT, rows_hint=3)


@pytest.mark.parametrize("n,dens,smin,rmin,share", [(11, 0.3, 10, 3, 2), (100, 0.05, 1, 1, 0), (100, 0.05, 2, 2, 1), (257, 0.02, 3, 1, 5),
                                                     (2500, 0.004, 2, 1, 3)])   # BASELINE configs[3] shape: 2500 samples, low-abundance rescue
@pytest.mark.parametrize("mode", [orc.MODE_BF, orc.MODE_BFC])
def test_synthetic_bf(ctx, n, dens, smin, rmin, share, mode):
    lower, W = 3 * 19200, 19200
    lists = synth_hash_lists(42 + n, n, lower, W, dens)
    check(ctx, lists, 1, [smin + (i % 2) for i in range(n)], rmin, share, mode, lower, lower + W - 1, bitw=2)
    if mode == orc.MODE_BFC:
        check(ctx, lists, 1, [1] * n, 1, 0, mode, lower, lower + W - 1, bitw=3)


@pytest.mark.parametrize("n,dens,smin,rmin,share,W", [(2, 0.3, 1, 1, 0, 6400), (11, 0.3, 10, 3, 2, 19200), (100, 0.05, 1, 1, 0, 19264), (257, 0.02, 3, 1, 5, 1000),
                                                       (1001, 0.01, 1, 2, 0, 12345),
                                                       (2500, 0.004, 2, 1, 1, 19200),    # BASELINE configs[3]: 2500 samples, --soft-min 2 --share-min 1
                                                       (3000, 0.003, 2, 2, 1, 40000),    # beyond 2729 samples only the cursors sit in LDS; three tiles of 16384 rows
                                                       (40, 0.002, 1, 2, 1, 300 * 16384 + 777)])   # 301 tiles of 16384 rows: more tiles than workgroups
def test_synthetic_bft(ctx, merge_kernel, n, dens, smin, rmin, share, W):
    """hash:bft:bin -- HashMerger::write_as_bft (merge.hpp:631-644): the BF rows bit-transposed on the device
    (k_merge_bf -> k_bit_transpose without leaving HBM); windows that are no multiple of 8 / 64 / the tile"""
    if merge_kernel != "rows":
        pytest.skip("Bloom modes have one kernel")
    lower = 5 * W
    lists = synth_hash_lists(4242 + n, n, lower, W, dens)
    rows = check(ctx, lists, 1, [smin + (i % 2) for i in range(n)], rmin, share, orc.MODE_BFT, lower, lower + W - 1)
    assert rows == (n + 7) // 8 * 8


@pytest.mark.parametrize("which", ["configs1_bf", "configs3_bft"])
def test_full_window_bloom_configs(ctx, merge_kernel, which):
    """BASELINE configs[1] and configs[3] at their real window against the oracle, bit for bit: 100 samples x 156 k hashes,
    hash:bf:bin, W = 3 125 056 (bloom 1e8 / 32 partitions, hash.hpp:31-40) -- and 2500 samples x 19.5 k hashes, hash:bft:bin
    --soft-min 2 --share-min 1, W = 3 906 304 (bloom 1e9 / 256): 239 tiles of 16384 rows, every sample block, the rescue"""
    if merge_kernel != "rows":
        pytest.skip("Bloom modes have one kernel")
    if which == "configs1_bf":
        n, W, per, mode, smin, share = 100, 3125056, 156250, orc.MODE_BF, 1, 0
    else:
        n, W, per, mode, smin, share = 2500, 3906304, 19531, orc.MODE_BFT, 2, 1
    lower = 3 * W
    rng = np.random.default_rng(99 + n)
    pool = np.unique(rng.integers(lower, lower + W, per, dtype=np.uint64))
    lists = []
    for i in range(n):      # the bench's cohort: 96.9 % of a shared pool + private hashes, counts 1..11 (a third below soft-min 2)
        keep = pool[rng.random(len(pool)) < 0.969]
        priv = rng.integers(lower, lower + W, int(per * 0.031), dtype=np.uint64)
        hs = np.unique(np.concatenate([keep, priv]))
        lists.append((hs.reshape(-1, 1), rng.integers(1, 12, len(hs), dtype=np.uint32)))
    rows = check(ctx, lists, 1, [smin] * n, 1, share, mode, lower, lower + W - 1)
    assert rows == (W if mode == orc.MODE_BF else (n + 7) // 8 * 8)


@pytest.mark.parametrize("kind", ["cohort", "nothing_solid", "half_solid"])
def test_bft_single_walk_and_its_fallback(ctx, merge_kernel, kind, monkeypatch, capfd):
    """hash:bft:bin with --share-min 1 (one-bit recurrences): k_merge_bft walks a tile's records ONCE, puts aside the non-solid
    records whose row has no solid record yet and settles them when the tile's map is complete.  cohort: few are put aside.
    nothing_solid: every record is below its soft-min -- everything is put aside, the scratch overflows, the batch runs again with
    two walks.  half_solid: private hashes, half of them solid -- a lot is put aside and dropped.  Same bytes and statistics as the
    oracle every time, and as the two-walk kernel (KMX_BFT_TWO_WALKS)"""
    if merge_kernel != "rows":
        pytest.skip("Bloom modes have one kernel")
    n, W = 300, 70000
    lower = 2 * W
    rng = np.random.default_rng(17)
    lists = []
    if kind == "cohort":
        pool = np.unique(rng.integers(lower, lower + W, 9000, dtype=np.uint64))
        for i in range(n):
            hs = np.unique(np.concatenate([pool[rng.random(len(pool)) < 0.95], rng.integers(lower, lower + W, 300, dtype=np.uint64)]))
            lists.append((hs.reshape(-1, 1), rng.integers(1, 12, len(hs), dtype=np.uint32)))
        soft = [2] * n
    else:
        for i in range(n):
            hs = np.unique(rng.integers(lower, lower + W, 6000, dtype=np.uint64))
            lists.append((hs.reshape(-1, 1), rng.integers(1, 5, len(hs), dtype=np.uint32)))
        soft = [100] * n if kind == "nothing_solid" else [3] * n
    monkeypatch.setenv("KMX_TRACE", "1")
    capfd.readouterr()
    for rmin in (1, 0):
        check(ctx, lists, 1, soft, rmin, 1, orc.MODE_BFT, lower, lower + W - 1)
    err = capfd.readouterr().err
    assert ("runs again with two walks" in err) == (kind == "nothing_solid")
    monkeypatch.setenv("KMX_BFT_TWO_WALKS", "1")
    from kmtricks_amd import lib
    c2 = lib.Context(0)
    check(c2, lists, 1, soft, 1, 1, orc.MODE_BFT, lower, lower + W - 1)
    c2.close()


def test_limits_fail_loudly(ctx, merge_kernel):
    """what this build does not take is refused with a message, not computed wrongly: more samples per hash:bft task than cursors
    fit the LDS, more than 4096 lists per COUNT / PA task"""
    if merge_kernel != "rows":
        pytest.skip("one run is enough")
    empty = (np.zeros((0, 1), np.uint64), np.zeros(0, np.uint32))
    with pytest.raises(Exception, match="hash:bft"):
        ctx.merge([empty] * 19000, 1, [1] * 19000, 1, 0, orc.MODE_BFT, 0, 6399)      # (round 6: 18396 cursors fit beside the 768-thread build's smaller tile)
    one = (np.array([[7]], np.uint64), np.array([3], np.uint32))
    check(ctx, [one] + [empty] * 8999, 1, [1] * 9000, 1, 1, orc.MODE_BFT, 0, 6399)      # (9000 samples: the tables no longer fit the LDS, only the cursors; smaller tiles)
    with pytest.raises(Exception, match="4096 lists"):
        ctx.merge([empty] * 5000, 1, [1] * 5000, 1, 0, orc.MODE_COUNT)
    check(ctx, [empty] * 3, 1, [1] * 3, 1, 0, orc.MODE_COUNT)      # (the context is still usable)


def test_bft_rows_are_the_per_sample_filters(ctx, merge_kernel):
    """row s of the BFT body == column s of the BF body (what howde_utils.hpp:133-187 copies into sample s's filter)"""
    if merge_kernel != "rows":
        pytest.skip("Bloom modes have one kernel")
    n, W, lower = 37, 6400, 6400
    lists = synth_hash_lists(7, n, lower, W, 0.1)
    bf, _, _ = ctx.merge(lists, 1, [1] * n, 1, 0, orc.MODE_BF, lower, lower + W - 1)
    bft, rows, _ = ctx.merge(lists, 1, [1] * n, 1, 0, orc.MODE_BFT, lower, lower + W - 1)
    bits = np.unpackbits(np.frombuffer(bf, np.uint8).reshape(W, -1), axis=1, bitorder="little")
    tb = np.unpackbits(np.frombuffer(bft, np.uint8).reshape(rows, W // 8), axis=1, bitorder="little")
    for s in range(n):
        assert np.array_equal(tb[s], bits[:, s])
        hs = lists[s][0].reshape(-1) - np.uint64(lower)
        assert np.array_equal(np.nonzero(tb[s])[0], hs.astype(np.int64))      # (soft-min 1, recurrence-min 1: every hash is a bit)
    assert not tb[n:].any()


def test_bfc_full_width_counts(ctx, merge_kernel):
    """--bitw 32: the cap 2^w - 1 must not wrap (packc.hpp:26-35)"""
    if merge_kernel != "rows":
        pytest.skip("Bloom modes have one kernel")
    lower, W = 0, 6400
    lists = synth_hash_lists(99, 9, lower, W, 0.2, count_max=100000)
    for w in (31, 32):
        check(ctx, lists, 1, [1] * 9, 1, 0, orc.MODE_BFC, lower, lower + W - 1, bitw=w)


def test_bench_scale_partitions_properties(ctx):
    """BASELINE configs[2] at its real shape -- 1000 samples, one of the 256 partitions of a 5 Mbp genome
    (19.5 k shared k-mers, 3 % private per sample, recurrence-min 2) -- through the device-resident batch
    API used by bench.py.  Checked by size-independent properties on every partition (keys strictly
    ascending, row set == keys present in >= 2 samples, column sums == per-sample totals of the kept
    keys, statistics) and bit-exactly against the oracle on the first one."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    N, shared, npriv = 1000, 19531, 605
    dev = torch.device("cuda", 0)
    parts = []
    for seed in (7, 8):
        lists = synth_lists(seed, N, shared, 0.969, npriv, kw=1, count_max=12)
        recs = [lib.pack_records(k, c, 1) for k, c in lists]
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
        parts.append((lists, offs, dt))
    torch.cuda.synchronize()
    tasks = [dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                  soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT) for _, offs, dt in parts]
    res = ctx.merge_dev(tasks)
    res.wait()
    for t, (lists, offs, dt) in enumerate(parts):
        body = np.frombuffer(res.body(t), dtype=np.uint8)
        rows = res.rows(t)
        assert res.row_bytes(t) == 8 + 4 * N and len(body) == rows * (8 + 4 * N)
        m = body.reshape(rows, 8 + 4 * N)
        keys = m[:, :8].copy().view(np.uint64).ravel()
        counts = m[:, 8:].copy().view(np.uint32).reshape(rows, N)
        assert np.all(keys[1:] > keys[:-1])                                   # ascending, distinct
        allk = np.concatenate([k.ravel() for k, _ in lists])
        uk, mult = np.unique(allk, return_counts=True)
        assert np.array_equal(keys, uk[mult >= 2])                            # soft-min 1: recurrence == multiplicity
        kept = set(keys.tolist())
        for i in (0, 1, 499, 999):                                            # column sums of a few samples
            k_i, c_i = lists[i]
            sel = np.isin(k_i.ravel(), keys)
            assert int(counts[:, i].sum()) == int(c_i[sel].astype(np.uint64).sum())
            assert int((counts[:, i] != 0).sum()) == int(sel.sum())
        st = res.stats(t)
        assert [int(x) for x in st[0]] == [0] * N                             # nothing is non-solid at soft-min 1
        assert [int(x) for x in st[2]] == [len(c) for _, c in lists]
        assert [int(x) for x in st[4]] == [int(c.astype(np.uint64).sum()) for _, c in lists]
        if t == 0:
            eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT)
            assert rows == er and body.tobytes() == eb and np.array_equal(st, es)
    res.free()


@pytest.mark.parametrize("similar", [True, False])
def test_kernel_selection_and_handback(ctx, monkeypatch, similar):
    """Default kernel choice (KMX_MERGE_KERNEL unset): a batch of 600-list tasks with recurrence-min 2 goes to
    k_merge_cols; lists that do not resemble each other are flagged and
    re-run with k_merge_rows.  Either way the body and the statistics equal the oracle's."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    ctx = lib.Context(0)          # own context: a hand-back makes a context skip the pivot kernel for its next batches
    N = 600
    lists = synth_lists(4242, N, 8000, 0.97, 240, kw=1) if similar else synth_lists(4243, N, 6000, 0.25, 1500, kw=1)
    dev = torch.device("cuda", 0)
    recs = [lib.pack_records(k, c, 1) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    task = dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT)
    res = ctx.merge_dev([task])
    res.wait()
    assert res.kernel() == ("k_merge_cols" if similar else "k_merge_rows")
    exp_body, exp_rows, exp_stats = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT, 0, 0, 2)
    assert res.rows(0) == exp_rows
    assert res.body(0) == exp_body
    st = res.stats(0)
    assert np.array_equal(st, exp_stats)
    res.free()
    if not similar:               # back-off: the next eligible batch of this context goes straight to k_merge_rows
        res = ctx.merge_dev([task]); res.wait()
        assert res.kernel() == "k_merge_rows" and res.body(0) == exp_body
        res.free()
    ctx.close()


@pytest.mark.parametrize("n", [513, 1024])
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
def test_pivot_list_count_edges(ctx, n, mode):
    """the pivot kernel at the edges of its list range (one past the selection threshold, the maximum), odd
    soft-min / recurrence-min values, a list that is empty and one that is much denser than the pivot"""
    lists = synth_lists(900 + n, n, 400, 0.96, 9, kw=1)
    lists[7] = (lists[7][0][:0], lists[7][1][:0])                                   # empty list
    dense = synth_lists(901 + n, 1, 400, 1.0, 2500, kw=1)[0]                         # 7x the records of the others
    lists[11] = dense
    soft = [1 + (i % 4) for i in range(n)]
    check(ctx, lists, 1, soft, 3, 0, mode)


def test_partial_handback_in_a_batch(monkeypatch):
    """A batch of three tasks of 600 lists: two cohorts the pivot / column-blocked kernel suits and one set of unrelated
    lists.  The kernel hands the second task back, libkmx re-runs that task alone with the next kernel down (cols ->
    pivot -> rows), and all three bodies and statistics equal the oracle's."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    first = os.environ.get("KMX_MERGE_KERNEL")
    if first == "rows":
        pytest.skip("nothing to hand back")
    ctx = lib.Context(0)
    N = 600
    sets = [synth_lists(5100, N, 5000, 0.97, 150, kw=1), synth_lists(5101, N, 5000, 0.25, 1200, kw=1), synth_lists(5102, N, 4000, 0.96, 160, kw=1)]
    dev = torch.device("cuda", 0)
    keep, tasks = [], []
    for lists in sets:
        recs = [lib.pack_records(k, c, 1) for k, c in lists]
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
        keep.append(dt)
        tasks.append(dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                          soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT))
    torch.cuda.synchronize()
    res = ctx.merge_dev(tasks)
    res.wait()
    assert res.kernel() == "k_merge_" + first            # two of the three tasks were completed by it
    for t, lists in enumerate(sets):
        eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT, 0, 0, 2)
        assert res.rows(t) == er and res.body(t) == eb and np.array_equal(res.stats(t), es), t
    res.free(); ctx.close()


def test_cols_kept_key_outside_the_row_keys():
    """k_merge_cols takes its row keys from a merge of 8 of the lists.  A key none of those 8 has, but two other lists
    in DIFFERENT column blocks do, reaches recurrence-min 2: only k_cols_sparse (which brings the set-aside records
    of all blocks together) can see it, and the row must be in the result."""
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("column-blocked kernel only")
    ctx = lib.Context(0)
    N = 600
    lists = synth_lists(6100, N, 3000, 0.97, 40, kw=1)
    a, b = 5, 450                                    # two different column blocks (128 lists each)
    extra = np.array([[(1 << 61) + 12345]], dtype=np.uint64)
    for i in (a, b):
        k, c = lists[i]
        k2 = np.concatenate([k, extra]); c2 = np.concatenate([c, np.array([7], np.uint32)])
        o = np.argsort(k2[:, 0]); lists[i] = (np.ascontiguousarray(k2[o]), np.ascontiguousarray(c2[o]))
    # the 8 merged lists: the one of median length (ties: lower index) in each eighth of the task (kmx_api.hip)
    lens = [len(c) for _, c in lists]
    sampled = set()
    for i in range(8):
        w = sorted(range(i * N // 8, (i + 1) * N // 8), key=lambda j: (lens[j], j))
        sampled.add(w[len(w) // 2])
    assert a not in sampled and b not in sampled
    rows = check(ctx, lists, 1, [1] * N, 2, 0, orc.MODE_COUNT)
    assert rows > 0
    # and without the planted key the column-blocked kernel completes the task itself
    lists2 = synth_lists(6100, N, 3000, 0.97, 40, kw=1)
    recs = [(k, c) for k, c in lists2]
    check(ctx, recs, 1, [1] * N, 2, 0, orc.MODE_COUNT)
    ctx.close()


@pytest.mark.parametrize("n", [9, 257, 700, 2500])
def test_cols_long_private_runs_and_odd_blocks(ctx, n):
    """column-blocked kernel: list counts that do not fill the blocks, a list with a run of private keys longer
    than its 64-record window between two row keys (second round of a tile), an empty list, soft-min > 1"""
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("column-blocked kernel only")
    lists = synth_lists(7000 + n, n, 500, 0.95, 6, kw=1)
    lists[3] = (lists[3][0][:0], lists[3][1][:0])
    k, c = lists[4]
    lo = int(k[len(k) // 2, 0])
    run = (np.arange(1, 301, dtype=np.uint64) + np.uint64(lo)).reshape(-1, 1)       # 300 consecutive keys nobody else has
    run = run[~np.isin(run[:, 0], k[:, 0])]
    k2 = np.concatenate([k, run]); c2 = np.concatenate([c, np.full(len(run), 3, np.uint32)])
    o = np.argsort(k2[:, 0]); lists[4] = (np.ascontiguousarray(k2[o]), np.ascontiguousarray(c2[o]))
    soft = [1 + (i % 3) for i in range(n)]
    check(ctx, lists, 1, soft, 2, 0, orc.MODE_COUNT)
    check(ctx, lists, 1, soft, 3, 0, orc.MODE_COUNT)


def test_cols_many_tiles_per_work_item(monkeypatch):
    """column-blocked kernel with work items of several tiles (the row-key table is rebuilt, the image reused, the
    window refilled in place tile after tile): 600 lists x ~20k records, few work items"""
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("column-blocked kernel only")
    monkeypatch.setenv("KMX_ITEMS_PER_SLOT", "1")
    ctx = lib.Context(0)
    N = 600
    lists = synth_lists(8100, N, 20000, 0.97, 600, kw=1, count_max=9)
    soft = [1 + (i % 2) for i in range(N)]
    exp_body, exp_rows, exp_stats = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, soft, 2, 0, orc.MODE_COUNT)
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    recs = [lib.pack_records(k, c, 1) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    task = dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                soft_min=soft, rec_min=2, share_min=0, mode=lib.MODE_COUNT)
    res = ctx.merge_dev([task]); res.wait()
    assert res.kernel() == "k_merge_cols"
    assert res.rows(0) == exp_rows and res.body(0) == exp_body and np.array_equal(res.stats(0), exp_stats)
    res.free(); ctx.close()


def test_batches_in_flight_waited_out_of_order():
    """Three batches queued back to back on one context (the preparation of one runs beside the merge of the one
    before), waited for and freed out of order; the third after the caller asked for the stream (its lists count as
    produced on it).  Every body and every statistic equals the oracle's, whatever kernel took the batch."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    ctx = lib.Context(0)
    dev = torch.device("cuda", 0)
    N = 600
    sets = [synth_lists(9100 + i, N, 2500 + 700 * i, 0.97, 60, kw=1) for i in range(3)]
    keep, tasks, exp = [], [], []
    for lists in sets:
        recs = [lib.pack_records(k, c, 1) for k, c in lists]
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
        keep.append(dt)
        tasks.append(dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                          soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT))
        exp.append(orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT))
    torch.cuda.synchronize()
    r0 = ctx.merge_dev([tasks[0]]); r1 = ctx.merge_dev([tasks[1], tasks[0]])
    assert ctx.stream is not None
    r2 = ctx.merge_dev([tasks[2]])
    r1.wait()
    for t, e in ((0, exp[1]), (1, exp[0])):
        assert r1.rows(t) == e[1] and r1.body(t) == e[0] and np.array_equal(r1.stats(t), e[2])
    r2.wait(); r0.wait()
    assert r2.rows(0) == exp[2][1] and r2.body(0) == exp[2][0] and np.array_equal(r2.stats(0), exp[2][2])
    r1.free()
    assert r0.rows(0) == exp[0][1] and r0.body(0) == exp[0][0] and np.array_equal(r0.stats(0), exp[0][2])
    r2.free(); r0.free(); ctx.close()


@pytest.mark.parametrize("rec_min", [5, 12, 21, 22])
def test_default_kernel_for_larger_recurrence_min(monkeypatch, rec_min):
    """KMX_MERGE_KERNEL unset, 600 similar lists: recurrence-min up to 21 goes to k_merge_cols (row keys from up to 32
    lists), above that to k_merge_pivot; the result is the oracle's either way."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    ctx = lib.Context(0)
    N = 600
    lists = synth_lists(9300 + rec_min, N, 4000, 0.975, 100, kw=1)
    dev = torch.device("cuda", 0)
    recs = [lib.pack_records(k, c, 1) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    task = dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                soft_min=[1] * N, rec_min=rec_min, share_min=0, mode=lib.MODE_COUNT)
    res = ctx.merge_dev([task]); res.wait()
    assert res.kernel() == ("k_merge_cols" if rec_min <= 21 else "k_merge_pivot")
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, rec_min, 0, orc.MODE_COUNT)
    assert res.rows(0) == er and res.body(0) == eb and np.array_equal(res.stats(0), es)
    res.free(); ctx.close()


@pytest.mark.parametrize("n", [600, 1001])
def test_default_kernel_for_presence_absence_rows(monkeypatch, n):
    """KMX_MERGE_KERNEL unset: PA rows of a cohort of more than 512 lists go to k_merge_cols too (a bit per list in the
    block's image, a byte per 8 lists out); list counts that are not a multiple of 8, an empty list, soft-min 2."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    ctx = lib.Context(0)
    lists = synth_lists(9500 + n, n, 5000, 0.97, 130, kw=1, count_max=6)
    lists[17] = (lists[17][0][:0], lists[17][1][:0])
    soft = [1 + (i % 2) for i in range(n)]
    dev = torch.device("cuda", 0)
    recs = [lib.pack_records(k, c, 1) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    task = dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(n)], key_words=1,
                soft_min=soft, rec_min=3, share_min=0, mode=lib.MODE_PA)
    res = ctx.merge_dev([task]); res.wait()
    assert res.kernel() == "k_merge_cols"
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, soft, 3, 0, orc.MODE_PA)
    assert res.rows(0) == er and res.body(0) == eb and np.array_equal(res.stats(0), es)
    res.free(); ctx.close()


@pytest.mark.parametrize("similar,order,N", [(True, False, 200), (False, False, 200), (True, True, 200), (True, True, 300), (False, True, 300)])
def test_default_kernel_from_192_lists(monkeypatch, similar, order, N):
    """KMX_MERGE_KERNEL unset, N lists x 25k records.  Rows left where the kernels put them: a cohort goes to k_merge_cols from 192
    lists.  Rows in file order (the library's default): from 257 lists, where k_merge_rows' windows halve -- up to 256 lists
    k_merge_rows is the faster one (profiles/r04_crossover.txt).  Unrelated lists are handed straight down to k_merge_rows
    (k_merge_pivot is not in the chain below 513 lists)."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    monkeypatch.delenv("KMX_COLS_MIN_LISTS", raising=False); monkeypatch.delenv("KMX_COLS_MIN_LISTS_ORD", raising=False)
    ctx = lib.Context(0)
    ctx.set_file_order(order)
    lists = synth_lists(9700, N, 25000, 0.97, 700, kw=1) if similar else synth_lists(9701, N, 60000, 0.3, 7000, kw=1)
    dev = torch.device("cuda", 0)
    recs = [lib.pack_records(k, c, 1) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    task = dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT)
    res = ctx.merge_dev([task]); res.wait()
    assert res.kernel() == ("k_merge_cols" if similar and N >= (257 if order else 192) else "k_merge_rows")
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT)
    assert res.rows(0) == er and res.body(0) == eb and np.array_equal(res.stats(0), es)
    res.free(); ctx.close()


def test_cols_scratch_budget(monkeypatch):
    """k_merge_cols is not chosen for a batch whose set-aside slices would exceed the scratch budget (KMX_COLS_SCRATCH_GB / _MB):
    the batch goes to the next kernel, same result."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    N = 600
    lists = synth_lists(9900, N, 3000, 0.97, 60, kw=1)
    ctx = lib.Context(0)
    dev = torch.device("cuda", 0)
    recs = [lib.pack_records(k, c, 1) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    torch.cuda.synchronize()
    task = dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=1,
                soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT)      # (slices of ~100 MB: sized by the row keys, i.e. the lists' lengths)
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * N, 2, 0, orc.MODE_COUNT)
    for budget, kern in (("8", "k_merge_pivot"), ("8000", "k_merge_cols")):
        monkeypatch.setenv("KMX_COLS_SCRATCH_MB", budget)
        res = ctx.merge_dev([task]); res.wait()
        assert res.kernel() == kern
        assert res.rows(0) == er and res.body(0) == eb and np.array_equal(res.stats(0), es)
        res.free()
    ctx.close()


def test_cols_recurrence_min_1_key_outside_the_row_keys(ctx):
    """column-blocked kernel forced at recurrence-min 1: a key that only a list outside the row-key lists holds is a
    row (found by scripts/stress_cols.py: the check must hand the task back for ANY key it was given)"""
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("column-blocked kernel only")
    lists = synth_lists(9950, 9, 40, 0.97, 1, kw=1)
    check(ctx, lists, 1, [1] * 9, 1, 0, orc.MODE_COUNT)
    lists = synth_lists(9951, 200, 300, 0.9, 3, kw=1)
    check(ctx, lists, 1, [1] * 200, 1, 0, orc.MODE_PA)


def test_cols_randomised_stress():
    """scripts/stress_cols.py: 30 random cohorts (list counts, sizes, similarity, recurrence-min, soft-min, count / PA)
    through the forced column-blocked kernel and its hand-back chain, seed 23 (whose case 23 once overflowed the retry
    arena after a cols -> pivot -> rows chain), every body and statistic equal to the oracle's."""
    import subprocess, sys
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("KMX_MERGE_KERNEL", None); env.pop("KMX_ITEMS_PER_SLOT", None)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_cols.py"), "30", "23"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "all 30 cases equal the oracle" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    # (the row-key kernels on the second stream, as round 1 had them: still a supported setting)
    env["KMX_COLS_PREP_OVERLAP"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_cols.py"), "10", "7"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "all 10 cases equal the oracle" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_cols_128bit_keys_randomised_stress():
    """scripts/stress_cols.py ... kw2: 25 random cohorts of 128-bit keys (k >= 32; key widths 66, 72 and 126 bits, so that
    both the low-word tie-break and the high word decide) through merge_cols_k2.hip and its hand-back chain."""
    import subprocess, sys
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("KMX_MERGE_KERNEL", None); env.pop("KMX_ITEMS_PER_SLOT", None)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_cols.py"), "25", "5", "cols", "kw2"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "all 25 cases equal the oracle" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("mode,rec_min", [(orc.MODE_COUNT, 2), (orc.MODE_PA, 1), (orc.MODE_PA, 3)])
def test_cols_128bit_keys_cohort(ctx, mode, rec_min):
    """a cohort of 300 samples with k = 63 (BASELINE configs[4]'s shape): the column-blocked kernel runs it (no hand-back)
    and every row equals the oracle's"""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("cols only")
    dev = torch.device("cuda", 0)
    n = 300
    lists = synth_lists(4242, n, 12000, 0.97, 100, kw=2, key_bits=126)
    recs = [lib.pack_records(k, c, 2) for k, c in lists]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
    task = dict(lists=[(dt.data_ptr() + 20 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(n)], key_words=2,
                soft_min=[1] * n, rec_min=rec_min, share_min=0, mode=mode)
    torch.cuda.synchronize()
    res = ctx.merge_dev([task]); res.wait()
    assert res.kernel() == "k_merge_cols"
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 2, [1] * n, rec_min, 0, mode)
    assert res.rows(0) == er and res.body(0) == eb and np.array_equal(res.stats(0), es)
    res.free()


@pytest.mark.parametrize("kw", [1, 2])
def test_cols_outlier_samples(monkeypatch, kw):
    """Outlier samples in a cohort (KMX_MERGE_KERNEL unset).  A list of three times the cohort's size fills its wave's set-aside slices:
    libkmx sees the length and runs k_merge_cols with slice extensions.  Three unrelated lists of the cohort's size among the eight of
    one wave do the same without a length to see: that batch is handed back (k_merge_rows), the context's next batch runs the build
    with extensions -- not the back-off.  Results equal the oracle's every time."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    dev = torch.device("cuda", 0)
    N = 300
    rng = np.random.default_rng(5)
    def unrelated(n):
        k = np.unique(rng.integers(0, 1 << 62, n, dtype=np.uint64)).reshape(-1, 1)
        if kw == 2: k = np.concatenate([rng.integers(0, 1 << 62, (len(k), 1), dtype=np.uint64), k], axis=1)      # (low word first; the high one ascends)
        return (np.ascontiguousarray(k), rng.integers(1, 9, len(k), dtype=np.uint32))
    def run(ctx, lists):
        recs = [lib.pack_records(k, c, kw) for k, c in lists]
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
        torch.cuda.synchronize()
        task = dict(lists=[(dt.data_ptr() + (8 * kw + 4) * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(N)], key_words=kw,
                    soft_min=[1] * N, rec_min=2, share_min=0, mode=lib.MODE_COUNT)
        res = ctx.merge_dev([task]); res.wait()
        eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], kw, [1] * N, 2, 0, orc.MODE_COUNT)
        ok = res.rows(0) == er and res.body(0) == eb and np.array_equal(res.stats(0), es)
        kern = res.kernel(); res.free()
        assert ok
        return kern
    ctx = lib.Context(0)
    kb = dict(kw=kw, key_bits=62 if kw == 1 else 126)
    long_one = synth_lists(11, N, 20000, 0.97, 200, **kb); long_one[137] = unrelated(60000)
    assert run(ctx, long_one) == "k_merge_cols"
    ctx.close()
    ctx = lib.Context(0)
    three = synth_lists(12, N, 20000, 0.97, 200, **kb)
    for i in (136, 137, 138): three[i] = unrelated(20000)
    assert run(ctx, three) == "k_merge_rows"          # handed back: slices full, nothing to see beforehand
    assert run(ctx, three) == "k_merge_cols"          # ... and the next batch of the context takes the extensions
    assert run(ctx, synth_lists(13, N, 20000, 0.97, 200, **kb)) == "k_merge_cols"
    ctx.close()


def test_batch_of_tasks_with_different_list_counts(monkeypatch):
    """One kmx_merge_dev batch of four tasks with 200, 1000, 257 and 600 lists (different block counts, tile sizes and
    row widths side by side), count and PA, libkmx's own kernel choice: every task equals the oracle."""
    torch = pytest.importorskip("torch")
    from kmtricks_amd import lib
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("one run is enough")
    monkeypatch.delenv("KMX_MERGE_KERNEL", raising=False)
    monkeypatch.setenv("KMX_COLS_MIN_LISTS_ORD", "192")      # (rows in file order: the pair is libkmx's choice from 257 lists; the knob brings the 200-list task in)
    ctx = lib.Context(0)
    dev = torch.device("cuda", 0)
    for mode in (lib.MODE_COUNT, lib.MODE_PA):
        sets = [synth_lists(9960 + i, n, pool, 0.97, pool // 40, kw=1) for i, (n, pool) in enumerate(((200, 9000), (1000, 2500), (257, 4000), (600, 3000)))]
        keep, tasks = [], []
        for lists in sets:
            n = len(lists)
            recs = [lib.pack_records(k, c, 1) for k, c in lists]
            offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
            dt = torch.from_numpy(np.concatenate(recs).view(np.int32)).to(dev)
            keep.append(dt)
            tasks.append(dict(lists=[(dt.data_ptr() + 12 * int(offs[i]), int(offs[i + 1] - offs[i])) for i in range(n)], key_words=1,
                              soft_min=[1] * n, rec_min=2, share_min=0, mode=mode))
        torch.cuda.synchronize()
        res = ctx.merge_dev(tasks); res.wait()
        assert res.kernel() == "k_merge_cols"
        for t, lists in enumerate(sets):
            eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], 1, [1] * len(lists), 2, 0, mode)
            assert res.rows(t) == er and res.body(t) == eb and np.array_equal(res.stats(t), es), (mode, t)
        res.free()
    ctx.close()


@pytest.mark.parametrize("workload", ["count", "pa63"])
def test_bench_workload_full_size_parity(workload):
    """The workload the bench line is quoted on, at its full size, on the lists the bench times (the product's count stage, 32
    partitions of the 1000 x 5 Mbp cohort / the 500-sample k = 63 cohort): k_merge_cols + k_cols_sparse == k_merge_pivot ==
    k_merge_rows by sha256 of every partition's body and statistics, rows ascending, rows out of k_cols_sparse present, partition
    0 == the oracle (scripts/verify_bench_parity.py)"""
    import sys
    if os.environ.get("KMX_MERGE_KERNEL") != "rows":      # (the autouse fixture runs every test once per kernel; verify() forces the kernels itself)
        pytest.skip("one run is enough")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import verify_bench_parity
    rep = verify_bench_parity.verify(workload, "counted")
    assert rep["all_kernels_equal_sha256"] and rep["rows_from_k_cols_sparse"] > 0 and rep["rows_total"] > 1_000_000


@pytest.mark.parametrize("n,kw,smin,rmin,share", [(300, 1, 10, 2, 1), (300, 1, 10, 2, 2), (600, 1, 25, 1, 1), (257, 2, 10, 3, 2), (300, 1, 10, 0, 0), (300, 1, 10, 0, 1), (200, 2, 12, 0, 0),
                                                  (300, 1, 10, 2, 215), (300, 1, 10, 1, 3), (257, 2, 10, 3, 185), (300, 1, 10, 0, 220)])
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
def test_cols_share_min_and_recurrence_min_0(n, kw, smin, rmin, share, mode):
    """share-min (rescue, merge.hpp:210-247) and recurrence-min 0 in the column-blocked pair (the RESC builds): cohorts with many
    non-solid records (counts 1..49 against soft-mins of 10..30) -- a row key's non-solid records are rescued in k_merge_cols, the
    keys outside the row keys have their solid records counted in k_cols_sparse (rows from recurrence-min, rescue from share-min,
    statistics for every key); with recurrence-min 0 a key only non-solid records hold is a row of zeros.  The result must be the
    oracle's, and must have come from k_merge_cols.  Share-min ABOVE the recurrence-min (215 of ~213 solid records per key: about
    half of the row keys' rows lose their rescued records again): count rows through the pair + k_share_fix, PA rows through k_merge_rows."""
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("column-blocked kernel only")
    from kmtricks_amd import lib
    torch = pytest.importorskip("torch")
    lists = synth_lists(4000 + n + 7 * rmin + share, n, 3000, 0.95, 40, kw=kw, key_bits=62 if kw == 1 else 100)
    # a few keys that only two or three samples hold, solid in some, not in others (the rows k_cols_sparse decides)
    soft = [smin + (i % 5) for i in range(n)]
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], kw, soft, rmin, share, mode)
    ctx = lib.Context(0)
    try:
        recs = [torch.from_numpy(lib.pack_records(k, c, kw).view(np.int32)).cuda() for k, c in lists]
        task = dict(lists=[(r.data_ptr(), r.shape[0]) for r in recs], key_words=kw, soft_min=soft, rec_min=rmin, share_min=share, mode=mode)
        res = ctx.merge_dev([task]); res.wait()
        above = share > max(1, rmin)
        assert res.kernel() == ("k_merge_rows" if above and mode == orc.MODE_PA else "k_merge_cols"), res.kernel()
        assert res.rows() == er
        body = res.body()
        assert body == eb, "body differs"
        assert res.body_from_arena() == eb, "arena + order differ"
        assert np.array_equal(res.stats(), es)
        assert int(es[1].sum()) > 0 or share == 0          # (something was rescued)
        res.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
@pytest.mark.parametrize("n,kw", [(40, 1), (300, 1), (260, 2)])
def test_arena_and_row_order(n, kw, mode):
    """kmx_result_arena + kmx_result_copy_order (the rows where the kernels left them and the order of the body's rows: what the
    pipeline's file writer turns into pwrites) give the body kmx_result_copy_body gives, for every kernel (segments of
    k_merge_rows / k_merge_pivot, the two lists of k_merge_cols + k_cols_sparse) -- and that body is the oracle's"""
    from kmtricks_amd import lib
    torch = pytest.importorskip("torch")
    lists = synth_lists(7000 + n, n, 2500, 0.93, 60, kw=kw, key_bits=62 if kw == 1 else 100)
    soft = [1 + (i % 3) for i in range(n)]
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], kw, soft, 2, 0, mode)
    ctx = lib.Context(0)
    try:
        recs = [torch.from_numpy(lib.pack_records(k, c, kw).view(np.int32)).cuda() for k, c in lists]
        res = ctx.merge_dev([dict(lists=[(r.data_ptr(), r.shape[0]) for r in recs], key_words=kw, soft_min=soft, rec_min=2, share_min=0, mode=mode)])
        res.wait()
        assert res.rows() == er and res.body_from_arena() == eb and res.body() == eb
        res.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("n,kw", [(200, 1), (300, 1), (1022, 1), (262, 2), (301, 1)])
@pytest.mark.parametrize("heavy", [0.002, 0.3])
def test_file_order_count_rows_across_the_one_byte_boundary(n, kw, heavy, monkeypatch):
    """Count rows in file order leave k_merge_cols through a side store that holds a byte per count where a column block's slice of
    a row has none above 254, and the 4-byte counts otherwise (a flag byte per block says which): counts of 254, 255, 256 and
    2^32-1 sprinkled over the lists (a few per row: most blocks narrow; and a third of all: most blocks wide), list counts
    whose last group of eight lists is ragged, one (301) whose rows are not a multiple of 8 bytes (the 4-byte rows alone)."""
    from kmtricks_amd import lib
    torch = pytest.importorskip("torch")
    if os.environ.get("KMX_MERGE_KERNEL") != "cols":
        pytest.skip("the side store is k_merge_cols' own")
    monkeypatch.setenv("KMX_DENSE_NARROW", "1")      # (by default from 512 lists on)
    lists = synth_lists(9100 + n, n, 3000, 0.9, 40, kw=kw, key_bits=62 if kw == 1 else 100, count_max=254)
    rng = np.random.default_rng(n)
    edge = np.array([254, 255, 256, 65535, 65536, 0xFFFFFFFF], dtype=np.uint32)
    for _, c in lists:
        m = rng.random(len(c)) < heavy
        c[m] = edge[rng.integers(0, len(edge), int(m.sum()))]
    soft = [1 + (i % 2) for i in range(n)]
    eb, er, es = orc.merge_matrix([(k.reshape(-1), c) for k, c in lists], kw, soft, 2, 0, orc.MODE_COUNT)
    ctx = lib.Context(0)
    try:
        recs = [torch.from_numpy(lib.pack_records(k, c, kw).view(np.int32)).cuda() for k, c in lists]
        for order in (True, False):
            ctx.set_file_order(order)
            res = ctx.merge_dev([dict(lists=[(r.data_ptr(), r.shape[0]) for r in recs], key_words=kw, soft_min=soft, rec_min=2, share_min=0, mode=orc.MODE_COUNT)])
            res.wait()
            assert res.kernel() == "k_merge_cols"
            assert res.rows() == er and res.body() == eb and np.array_equal(res.stats(0), es)
            res.free()
    finally:
        ctx.close()


def test_hand_worked_merge_vectors_through_the_hip_path(ctx):
    """the vectors of tests/test_merge_independent.py -- every cell and statistic written down from merge.hpp:183-260, the deciding
    line cited there -- through the C ABI: the HIP merge is pinned to the reference's arithmetic without the oracle in between
    (rescue granted / denied at the share-min boundary, non-solid records with share-min 0, recurrence-min 0 rows of zeros, each of
    the six statistics; 64- to 256-bit keys)"""
    from test_merge_independent import HAND_CASES, HAND_SOFT, hand_arrays, body_of
    for kw, shift in ((1, 0), (1, 40), (2, 70), (3, 130), (4, 200)):
        lists = hand_arrays(kw, shift)
        for r, s, rows, stats in HAND_CASES:
            for mode in (orc.MODE_COUNT, orc.MODE_PA):
                body, n, st = ctx.merge(lists, kw, HAND_SOFT, r, s, mode)
                assert n == len(rows) and body == body_of(rows, kw, 3, mode, shift), (kw, r, s, mode)
                assert st.tolist() == stats, (kw, r, s, mode)


@pytest.mark.gpu
def test_hand_worked_bloom_vectors_through_the_hip_path(ctx):
    """round 6: the hand-worked write_as_bf / bfc / bft vectors of tests/test_merge_independent.py (gap-filling zero rows, a hash nobody
    holds solid, `current = m_current + 1`, MSB-first pack_v at widths 2 and 3, the closing fill to `upper`, the transposed rows with
    their padding) through kmx_merge on the device, bodies and statistics"""
    from test_merge_independent import bloom_cases, bloom_arrays, BLOOM_SOFT, BLOOM_LOWER, BLOOM_UPPER
    for mode, r, s, w, body, stats in bloom_cases():
        got, rows, st = ctx.merge(bloom_arrays(), 1, BLOOM_SOFT, r, s, mode, BLOOM_LOWER, BLOOM_UPPER, w)
        assert got == body, (mode, r, s, w, got.hex(), body.hex())
        assert rows == (8 if mode == orc.MODE_BFT else 15)
        assert st.tolist() == stats, (mode, r, s, w)


@pytest.mark.parametrize("seed", range(6))
def test_hip_merge_against_the_second_restatement(ctx, seed):
    """random cohorts (up to 300 lists, so that the column-blocked pair takes them when it is the forced kernel) judged by
    `dict_merge` -- a per-key dictionary over all samples, no cursors: another shape of merge.hpp:183-260 than the oracle's"""
    from test_merge_independent import dict_merge, random_cohort, body_of
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([2, 7, 40, 200, 300]))
    kw = int(rng.choice([1, 2]))
    lists = random_cohort(rng, n, 3000, kw)
    soft = [int(x) for x in rng.integers(1, 5, n)]
    arrays = []
    for l in lists:
        ks = sorted(l)
        arrays.append((np.array([[(key >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(kw)] for key in ks], np.uint64).reshape(len(ks), kw),
                       np.array([l[key] for key in ks], np.uint32)))
    for r, s in ((1, 0), (2, 1), (0, 0), (2, 3), (1, 2)):
        rows, stats = dict_merge(lists, soft, r, s)
        for mode in (orc.MODE_COUNT, orc.MODE_PA):
            body, nrows, st = ctx.merge(arrays, kw, soft, r, s, mode)
            assert nrows == len(rows) and body == body_of(rows, kw, n, mode), (n, kw, r, s, mode)
            assert st.tolist() == stats, (n, kw, r, s)
