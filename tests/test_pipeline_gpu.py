"""BASELINE configs[0]: the two tiny FASTA samples of the reference's tests through `kmx pipeline`
(the kmtricks-pipeline-compatible driver over libkmx): run-directory layout, file headers, matrix
bodies, merge_info, plugin.  Expected values: the reference's own goldens (57/67/70/82 rows with the
fixture repartition) and the oracle for the bodies."""
import gzip, json, os, struct, subprocess
import numpy as np
import pytest

import orc
import kmfiles
from test_oracle_goldens import repart_table, read_fasta, GD, G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMX = os.path.join(ROOT, "kmtricks_amd", "kmx")
K, M, P = 31, 10, 4


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("kmxin")
    t = repart_table()
    with open(d / "fixture.minimRepart", "wb") as f:   # layout of tests/data/repart_gatb/repartition.minimRepart
        f.write(struct.pack("<HQH", 4, len(t), 1)); f.write(t.tobytes()); f.write(struct.pack("<BI", 0, 0x12345678))
    assert os.path.getsize(d / "fixture.minimRepart") == G["repartition_table"]["file_size"]
    # second sample as FASTQ.gz to exercise the reader
    with gzip.open(d / "2.fastq.gz", "wt") as f:
        for i, s in enumerate(read_fasta(os.path.join(GD, "2.fasta"))):
            f.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
    with open(d / "in.fof", "w") as f:
        f.write(f"D1 : {GD}/1.fasta\nD2 : {d}/2.fastq.gz\n")
    return d


def run(inputs, out, *args):
    cmd = [KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(out), "--kmer-size", "31", "--hard-min", "1",
           "--nb-partitions", "4", "--repart-file", str(inputs / "fixture.minimRepart")] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def oracle_lists(hash_mode, W=None):
    lut = orc.minimizer_lut(M); rep = repart_table()
    sk = [orc.superk_partition(read_fasta(os.path.join(GD, f)), K, M, lut, rep, P) for f in ("1.fasta", "2.fasta")]
    out = []
    for p in range(P):
        if hash_mode:
            out.append([orc.count_hash(s[p][0], K, W, p, 1) for s in sk])
        else:
            out.append([tuple(x if i else x.reshape(-1) for i, x in enumerate(orc.count_kmer(s[p][0], K, 1))) for s in sk])
    return out


def test_kmer_count_pipeline(inputs, tmp_path):
    out = run(inputs, tmp_path / "run", "--mode", "kmer:count:bin", "--keep-tmp")
    for d in ("superkmers", "counts", "matrices", "merge_infos", "partition_infos", "fpr", "repartition_gatb"):
        assert (out / d).is_dir()
    for f in ("kmtricks.fof", "options.txt", "build_infos.txt", "run_infos.txt", "hash.info"):
        assert (out / f).is_file()
    assert open(out / "repartition_gatb" / "repartition.minimRepart", "rb").read() == open(inputs / "fixture.minimRepart", "rb").read()
    # minimizers/minimizers.<p>: every 10-mer once, in the partition the table gives it, ascending inside a file
    rep_tab = np.frombuffer(open(inputs / "fixture.minimRepart", "rb").read()[12:12 + 2 * 4 ** 10], np.uint16)
    total = 0
    for p in range(P):
        mm = open(out / "minimizers" / f"minimizers.{p}").read().split()
        vals = [sum("ACTG".index(c) << (2 * (9 - j)) for j, c in enumerate(m)) for m in (mm if len(mm) <= 100 else mm[:50] + mm[-50:])]
        assert all(len(m) == 10 for m in mm) and all(rep_tab[v] == p for v in vals) and vals == sorted(vals)
        assert len(mm) == int((rep_tab == p).sum())
        total += len(mm)
    assert total == 4 ** 10
    lists = oracle_lists(False)
    rows_exp = G["merge_test"]["kmer_rows"]
    for p in range(P):
        # count files (kept by --keep-tmp) hold the reference's committed k-mers
        for i, s in enumerate(("D1", "D2")):
            f = kmfiles.read_kmer_file(str(out / "counts" / f"partition_{p}" / f"{s}.kmer"))
            ref = kmfiles.read_kmer_file(f"{GD}/partitions/kmers/partition_{p}/{s}.kmer")
            assert (f["k"], f["id"], f["partition"], f["count_slots"]) == (31, i, p, 4)
            assert np.array_equal(f["keys"], ref["keys"]) and np.array_equal(f["counts"], ref["counts"])
        raw = open(out / "matrices" / f"matrix_{p}.count", "rb").read()
        hdr = struct.unpack("<QIBQIIIIII", raw[:45])
        assert hdr == (kmfiles.KM_MAGIC, 0, 0, 0x6b5f78697274616d, 31, 1, 1, 2, 0, p)   # count_slots literal 1 (Appendix B-2)
        body, rows, stats = orc.merge_matrix(lists[p], 1, [1, 1], 1, 0, orc.MODE_COUNT)
        assert rows == rows_exp[p] and raw[45:] == body
        mi = open(out / "merge_infos" / f"partition{p}.merge_info").read().splitlines()
        assert mi[0].split("\t")[0] == "NON_SOLID" and len(mi) == 6
        assert [int(x) for x in mi[2].split("\t")[1:3]] == [int(x) for x in stats[2]]
    # super-k-mer files: block framing + the reference's k-mer counts per file (task_main.cpp:85-114)
    for s in ("D1", "D2"):
        info = open(out / "superkmers" / s / "SuperKmerBinInfoFile").read().split("\n")
        assert info[0] == "skp" and info[2] == "4"
        assert [int(x) for x in info[3:11]] == G["task_main"]["superk_info_" + s][1:]
    pin = [int(x) for x in open(out / "partition_infos" / "D1.pinfo").read().split()]
    assert pin == G["task_main"]["superk_info_D1"][1::2]


def test_soft_min_file(inputs, tmp_path):
    """--soft-min <file>: one threshold per sample in fof order (src/cli.cpp:228-248, cmd/all.hpp:150-162)"""
    (tmp_path / "soft.txt").write_text("1\n2\n")
    out = run(inputs, tmp_path / "run", "--mode", "kmer:count:bin", "--soft-min", str(tmp_path / "soft.txt"), "--recurrence-min", "1", "--share-min", "1")
    lists = oracle_lists(False)
    for p in range(P):
        raw = open(out / "matrices" / f"matrix_{p}.count", "rb").read()
        body, rows, _ = orc.merge_matrix(lists[p], 1, [1, 2], 1, 1, orc.MODE_COUNT)
        assert raw[45:] == body
    (tmp_path / "soft1.txt").write_text("1\n")
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / "r2"), "--soft-min", str(tmp_path / "soft1.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "different from the number of samples" in r.stderr


@pytest.mark.parametrize("frac", ["0.2", "0.95", "1.0"])
def test_soft_min_fraction_is_the_reference_s(inputs, tmp_path, frac):
    """--soft-min <fraction> (src/cli.cpp:228-248): kmtricks 1.6.0 derives thresholds from the abundance histograms but appends
    them to a vector that already holds one zero per sample (histogram.hpp:218-243), so the merge reads zeros: the run is the
    `--soft-min 0 --hist` run, and merge_amin.txt holds N zeros followed by the values it computed -- per sample the first index of
    the unique bins at which the running sum exceeds unique() * p, none when it never does"""
    out = run(inputs, tmp_path / "f", "--mode", "kmer:count:bin", "--soft-min", frac, "--recurrence-min", "1")
    ref = run(inputs, tmp_path / "z", "--mode", "kmer:count:bin", "--soft-min", "0", "--recurrence-min", "1", "--hist")
    for sub in ("matrices", "merge_infos", "histograms"):
        names = sorted(os.listdir(ref / sub))
        assert names and names == sorted(os.listdir(out / sub))
        for f in names:
            assert open(out / sub / f, "rb").read() == open(ref / sub / f, "rb").read(), (sub, f)
    opts = open(out / "options.txt").read()
    assert "m_ab_float=1" in opts and "hist=1" in opts
    lists = oracle_lists(False)
    exp = [0, 0]
    for si in range(2):
        h = None
        for p in range(P): h = orc.khist(lists[p][si][1], 1, 255, acc=h)
        n = int(np.uint32(int(float(int(h["sums"][0])) * float(frac))))
        acc = 0
        for i, v in enumerate(h["unique"]):
            if acc > n:
                exp.append(i); break
            acc = (acc + int(v)) & 0xFFFFFFFF
    assert open(out / "merge_amin.txt").read() == "".join(f"{v}\n" for v in exp)
    assert not os.path.exists(ref / "merge_amin.txt")


def test_pa_and_recurrence_pipeline(inputs, tmp_path):
    out = run(inputs, tmp_path / "run", "--mode", "kmer:pa:bin", "--recurrence-min", "1")
    lists = oracle_lists(False)
    for p in range(P):
        raw = open(out / "matrices" / f"matrix_{p}.pa", "rb").read()
        assert struct.unpack("<QIBQIIIIII", raw[:45]) == (kmfiles.KM_MAGIC, 0, 0, 0x6b5f74616d6170, 31, 1, 2, 1, 0, p)
        body, rows, _ = orc.merge_matrix(lists[p], 1, [1, 1], 1, 0, orc.MODE_PA)
        assert raw[45:] == body
        assert not (out / "counts" / f"partition_{p}" / "D1.kmer").exists()   # tmp files removed without --keep-tmp


@pytest.mark.parametrize("mode,omode,ext,hdrlen", [("hash:count:bin", orc.MODE_COUNT, "count_hash", 37), ("hash:bf:bin", orc.MODE_BF, "cmbf", 49),
                                                   ("hash:bfc:bin", orc.MODE_BFC, "cmbf", 49)])
def test_hash_pipeline(inputs, tmp_path, mode, omode, ext, hdrlen):
    out = run(inputs, tmp_path / "run", "--mode", mode, "--bloom-size", "1000000", "--bitw", "2")
    hi = struct.unpack("<QQQQI", open(out / "hash.info", "rb").read())
    assert hi == (250048 * 4, 4, 250048, 31256, 10)        # the window of the committed .hash fixtures
    W = hi[2]
    lists = oracle_lists(True, W)
    for p in range(P):
        raw = open(out / "matrices" / f"matrix_{p}.{ext}", "rb").read()
        body, rows, stats = orc.merge_matrix([(h, c) for h, c in lists[p]], 1, [1, 1], 1, 0, omode, W * p, W * (p + 1) - 1, 2)
        assert raw[hdrlen:] == body
        if ext == "cmbf":
            bits = 2 if omode == orc.MODE_BF else 4
            assert struct.unpack("<QIBQIQQII", raw[:49]) == (kmfiles.KM_MAGIC, 0, 0, 0x74616d746962, bits, W * p, W, 0, p)
        if omode == orc.MODE_BF:
            fpr = [float(x) for x in open(out / "fpr" / f"partition_{p}.txt").read().split()]
            exp = [1.0 - np.exp(-float(n) / W) for n in stats[3]]
            assert np.allclose(fpr, exp, atol=1e-6)


def test_plugin_pipeline(inputs, tmp_path):
    plug = os.path.join(ROOT, "kmtricks_amd", "libkmx_test_plugin.so")
    out = run(inputs, tmp_path / "run", "--mode", "kmer:count:bin", "--plugin", plug, "--plugin-config", "0")
    lists = oracle_lists(False)
    for p in range(P):
        raw = open(out / "matrices" / f"matrix_{p}.count", "rb").read()[45:]
        body, rows, _ = orc.merge_matrix(lists[p], 1, [1, 1], 0, 0, orc.MODE_COUNT)
        exp = np.frombuffer(body, np.uint8).reshape(rows, 16).copy()
        c0 = exp[:, 8:12].copy().view(np.uint32); c0 *= 2; exp[:, 8:12] = c0.view(np.uint8)   # the plugin doubles sample 0
        assert raw == exp.tobytes()
    out2 = run(inputs, tmp_path / "run2", "--mode", "kmer:count:bin", "--plugin", plug, "--plugin-config", "1")
    # threshold 1: only rows present in BOTH samples survive; the two fixtures share no 31-mer
    assert all(os.path.getsize(out2 / "matrices" / f"matrix_{p}.count") == 45 for p in range(P))


@pytest.mark.parametrize("what", ["bf", "bfc", "bft"])
@pytest.mark.parametrize("threshold", [0, 1])
def test_plugin_in_bloom_modes(inputs, tmp_path, what, threshold):
    """--plugin with hash:bf|bfc|bft:bin (merge.hpp:509-514, 575-644): process_hash sees every hash of the window's lists in ascending
    order with the counts the soft-min rule left, its answer replaces the recurrence test, and the Bloom rows are packed from the
    counts it left (the test plugin keeps a row when every sample's count reaches the threshold and doubles sample 0's count)."""
    plug = os.path.join(ROOT, "kmtricks_amd", "libkmx_test_plugin.so")
    out = run(inputs, tmp_path / "run", "--mode", f"hash:{what}:bin", "--bloom-size", "1000000", "--bitw", "3", "--plugin", plug, "--plugin-config", str(threshold))
    W = struct.unpack("<QQQQI", open(out / "hash.info", "rb").read())[2]
    lists = oracle_lists(True, W)
    any_kept = False
    for p in range(P):
        body, rows, _ = orc.merge_matrix([(h, c) for h, c in lists[p]], 1, [1, 1], 0, 0, orc.MODE_COUNT)
        m = np.frombuffer(body, np.uint8).reshape(rows, 16)
        hs = m[:, :8].copy().view(np.uint64).ravel(); cs = m[:, 8:].copy().view(np.uint32).reshape(rows, 2).astype(np.uint64)
        keep = (cs >= threshold).all(axis=1)
        cs[:, 0] *= 2
        any_kept |= bool(keep.any())
        rb = 1 if what != "bfc" else (2 * 3 + 7) // 8
        exp = np.zeros((W, rb), np.uint8)
        for h, c, k_ in zip(hs, cs, keep):
            if not k_: continue
            r = int(h) - W * p
            if what == "bfc":
                for i in range(2):
                    v = min(int(c[i]).bit_length(), 7)
                    for b in range(3):
                        if (v >> (2 - b)) & 1:
                            bit = i * 3 + b; exp[r, bit >> 3] |= 0x80 >> (bit & 7)
            else:
                for i in range(2):
                    if c[i]: exp[r, 0] |= 1 << i
        raw = open(out / "matrices" / f"matrix_{p}.cmbf", "rb").read()
        assert struct.unpack("<QIBQIQQII", raw[:49])[4:7] == (2 * 3 if what == "bfc" else 2, W * p, W)
        if what == "bft":
            exp = orc.transpose_bits(exp.reshape(-1), W, 8).reshape(8, W // 8)
        assert raw[49:] == exp.tobytes(), (what, threshold, p)
    assert any_kept == (threshold == 0)      # (the two fixtures share no 31-mer: with threshold 1 nothing survives)


def test_cli_errors(inputs, tmp_path):
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "already exists" in r.stderr          # src/cli.cpp:101-104
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / "x"), "--mode", "hash:bfx:bin"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "not supported" in r.stderr
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / "y"), "--mode", "hash:bft:bin",
                        "--restrict-to-list", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "requires all partitions" in r.stderr  # cmd/all.hpp:137-143
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / "w"), "--soft-min", "1.5"], capture_output=True, text=True)
    assert r.returncode == 1 and "Not in range [0.0, 1.0]" in r.stderr      # src/cli.cpp:237
    (tmp_path / "bad.fof").write_text(f"D1 : {tmp_path}/missing.fasta\n")
    r = subprocess.run([KMX, "pipeline", "--file", str(tmp_path / "bad.fof"), "--run-dir", str(tmp_path / "z"), "--static-repart"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "[error]" in r.stderr and "missing.fasta" in r.stderr   # an unreadable input is an error line, not an abort


def test_bft_pipeline_and_filters(inputs, tmp_path):
    """hash:bft:bin end to end (BASELINE configs[3] in small): matrix_<p>.cmbf = HashMerger::write_as_bft (merge.hpp:631-644) and
    filters/<id>.bf = header + u64 bits + the sample's row of every partition (howde_utils.hpp:133-187)"""
    out = run(inputs, tmp_path / "run", "--mode", "hash:bft:bin", "--bloom-size", "1000000", "--soft-min", "1", "--share-min", "1")
    W = 250048
    lists = oracle_lists(True, W)
    rows_of = {0: [], 1: []}
    for p in range(P):
        raw = open(out / "matrices" / f"matrix_{p}.cmbf", "rb").read()
        assert struct.unpack("<QIBQIQQII", raw[:49]) == (kmfiles.KM_MAGIC, 0, 0, 0x74616d746962, 2, W * p, W, 0, p)
        body, rows, stats = orc.merge_matrix([(h, c) for h, c in lists[p]], 1, [1, 1], 1, 1, orc.MODE_BFT, W * p, W * (p + 1) - 1)
        assert rows == 8 and raw[49:] == body
        for s in (0, 1):
            rows_of[s].append(body[s * (W // 8):(s + 1) * (W // 8)])
        fpr = [float(x) for x in open(out / "fpr" / f"partition_{p}.txt").read().split()]
        assert np.allclose(fpr, [1.0 - np.exp(-float(n) / W) for n in stats[3]], atol=1e-6)
    for s, name in enumerate(("D1", "D2")):
        bf = open(out / "filters" / f"{name}.bf", "rb").read()
        assert len(bf) == 112 + 8 + 4 * W // 8
        magic, hsize, version, kind, _pad, ksz, nh = struct.unpack("<QIIIIII", bf[:32])
        assert (hsize, kind, ksz, nh) == (112, 1, 31, 1)
        assert struct.unpack("<QQ", bf[48:64]) == (4 * W, 4 * W)             # hashModulus, numBits = bloom size
        assert struct.unpack("<IIQQQ", bf[80:112]) == (1, 0, 112, 4 * W // 8 + 8, 0)
        assert struct.unpack("<Q", bf[112:120]) == (4 * W,)
        assert bf[120:] == b"".join(rows_of[s])
        # the filter holds exactly the sample's window hashes (soft-min 1)
        bits = np.unpackbits(np.frombuffer(bf[120:], np.uint8), bitorder="little")
        hs = np.concatenate([lists[p][s][0] for p in range(P)]).astype(np.int64)
        assert np.array_equal(np.nonzero(bits)[0], np.sort(hs))


def test_sampled_repartition(tmp_path):
    """the default `kmtricks pipeline` (no --static-repart): gatb's sampled repartition (RepartitionAlgorithm.cpp:395-496,
    PartiInfo.cpp:48-103) on the reference's two test samples reproduces its committed tests/data/repart_gatb table"""
    (tmp_path / "t.fof").write_text(f"D1 : {GD}/1.fasta\nD2 : {GD}/2.fasta\n")
    out = tmp_path / "run"
    r = subprocess.run([KMX, "pipeline", "--file", str(tmp_path / "t.fof"), "--run-dir", str(out), "--kmer-size", "31", "--hard-min", "1",
                        "--nb-partitions", "4", "--until", "repart"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out / "repartition_gatb" / "repartition.minimRepart", "rb").read()
    assert len(raw) == G["repartition_table"]["file_size"]
    got = np.frombuffer(raw[12:12 + 2 * 4 ** 10], np.uint16)
    assert np.array_equal(got, repart_table())
    # the table of a cohort balances the kx-mers of the sampled reads: no partition is empty, loads within a few percent
    reads = _synthetic_samples(tmp_path, 3, 60_000, 5)
    out2 = tmp_path / "run2"
    r = subprocess.run([KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--run-dir", str(out2), "--kmer-size", "31", "--nb-partitions", "8",
                        "--until", "repart"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tab = np.frombuffer(open(out2 / "repartition_gatb" / "repartition.minimRepart", "rb").read()[12:12 + 2 * 4 ** 10], np.uint16)
    lut = orc.minimizer_lut(10)
    _, _, _, mx = orc.superk_stats([x for rs in reads for x in rs], 31, 10, lut, np.zeros(4 ** 10, np.uint16), 1)
    exp = orc.repart_sampled(mx, 8)
    assert np.array_equal(tab, exp)
    load = np.bincount(tab, weights=mx.astype(np.float64), minlength=8)
    assert load.min() > 0.9 * load.mean()


def test_cpr_and_restrict(inputs, tmp_path):
    """--cpr: lz4-framed count files and matrix bodies decompress to the uncompressed run's bytes; --restrict-to-list"""
    import ctypes
    lz4 = ctypes.CDLL("liblz4.so.1")
    def unlz4(b):
        dctx = ctypes.c_void_p()
        assert lz4.LZ4F_createDecompressionContext(ctypes.byref(dctx), 100) == 0
        lz4.LZ4F_decompress.restype = ctypes.c_size_t
        out = bytearray(); src = ctypes.create_string_buffer(b, len(b)); pos = 0
        while pos < len(b):
            dst = ctypes.create_string_buffer(1 << 16); dn = ctypes.c_size_t(1 << 16); sn = ctypes.c_size_t(len(b) - pos)
            r = lz4.LZ4F_decompress(dctx, dst, ctypes.byref(dn), ctypes.byref(src, pos), ctypes.byref(sn), None)
            assert not lz4.LZ4F_isError(ctypes.c_size_t(r))
            out += dst.raw[:dn.value]; pos += sn.value
            if r == 0 and dn.value == 0 and sn.value == 0: break
        return bytes(out)
    plain = run(inputs, tmp_path / "plain", "--mode", "kmer:count:bin", "--keep-tmp")
    cpr = run(inputs, tmp_path / "cpr", "--mode", "kmer:count:bin", "--keep-tmp", "--cpr")
    for p in range(P):
        a = open(plain / "matrices" / f"matrix_{p}.count", "rb").read()
        b = open(cpr / "matrices" / f"matrix_{p}.count.lz4", "rb").read()
        assert b[12] == 1 and a[:12] == b[:12] and a[13:45] == b[13:45] and unlz4(b[45:]) == a[45:]
        for s in ("D1", "D2"):
            a = open(plain / "counts" / f"partition_{p}" / f"{s}.kmer", "rb").read()
            b = open(cpr / "counts" / f"partition_{p}" / f"{s}.kmer.lz4", "rb").read()
            assert b[12] == 1 and unlz4(b[41:]) == a[41:]
            a = open(plain / "superkmers" / s / f"skp.{p}", "rb").read()
            b = open(cpr / "superkmers" / s / f"skp.{p}", "rb").read()
            assert b[12] == 1 and unlz4(b[25:]) == a[25:]
    hc = run(inputs, tmp_path / "hcpr", "--mode", "hash:pa:bin", "--cpr", "--bloom-size", "1000000")
    hp = run(inputs, tmp_path / "hplain", "--mode", "hash:pa:bin", "--bloom-size", "1000000")
    for p in range(P):   # hash-mode matrices keep their name, the body is an lz4 frame (task.hpp:794-795, 817)
        a = open(hp / "matrices" / f"matrix_{p}.pa_hash", "rb").read(); b = open(hc / "matrices" / f"matrix_{p}.pa_hash", "rb").read()
        assert b[12] == 1 and unlz4(b[37:]) == a[37:]
    sub = run(inputs, tmp_path / "sub", "--mode", "kmer:count:bin", "--restrict-to-list", "1,3")
    assert sorted(os.listdir(sub / "matrices")) == ["matrix_1.count", "matrix_3.count"]
    for p in (1, 3):
        assert open(sub / "matrices" / f"matrix_{p}.count", "rb").read() == open(plain / "matrices" / f"matrix_{p}.count", "rb").read()


def _hist_bytes(k, idx, counts_per_partition):
    """a .hist file as HistWriter leaves it (io/hist_file.hpp:30-116) for KHist(idx, k, 1, 255) fed the given counts"""
    h = None
    for c in counts_per_partition: h = orc.khist(c, 1, 255, acc=h)
    hdr = struct.pack("<QIB", 0x736b636972746d6b, 0, 0) + struct.pack("<QIIQQQQQQQQ", 0x747369686b, k, idx, 1, 255, int(h["sums"][0]), int(h["sums"][1]),
                                                                int(h["oob"][2]), int(h["oob"][0]), int(h["oob"][3]), int(h["oob"][1]))
    return hdr + h["unique"].tobytes() + h["total"].tobytes(), h


@pytest.mark.parametrize("mode", ["kmer:count:bin", "hash:count:bin"])
def test_hist_pipeline(inputs, tmp_path, mode):
    """--hist: histograms/<id>.hist of every sample = KHist(i, k, 1, 255) over ALL its distinct k-mers / hashes (before --hard-min
    3), as a file and through `kmx dump`; with --restrict-to-list only the selected partitions are counted"""
    hashed = mode.startswith("hash")
    extra = ["--bloom-size", "1000000"] if hashed else []
    out = run(inputs, tmp_path / "run", "--mode", mode, "--hist", "--hard-min", "3", *extra)
    W = None
    if hashed:
        W = struct.unpack_from("<QQQ", open(out / "hash.info", "rb").read(), 0)[2]
    lists = oracle_lists(hashed, W)      # hard-min 1: every distinct key
    assert "hist=1" in open(out / "options.txt").read()
    for si, sid in enumerate(("D1", "D2")):
        exp, h = _hist_bytes(K, si, [lists[p][si][1] for p in range(P)])
        assert open(out / "histograms" / f"{sid}.hist", "rb").read() == exp
        r = subprocess.run([KMX, "dump", "--input", str(out / "histograms" / f"{sid}.hist")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == f"@LOWER=1\n@UPPER=255\n@OOB_L=0\n@OOB_U={int(h['oob'][1])}\n" + "".join(f"{c} {int(v)}\n" for c, v in zip(range(1, 256), h["unique"]))
        assert int(h["sums"][0]) > 100
    if not hashed:
        sub = run(inputs, tmp_path / "sub", "--mode", mode, "--hist", "--restrict-to-list", "1,3")
        for si, sid in enumerate(("D1", "D2")):
            exp, _ = _hist_bytes(K, si, [lists[p][si][1] for p in (1, 3)])
            assert open(sub / "histograms" / f"{sid}.hist", "rb").read() == exp
    plain = run(inputs, tmp_path / "nohist", "--mode", mode, *extra)
    assert os.listdir(plain / "histograms") == []


@pytest.mark.parametrize("mode", ["kmer:count:bin", "hash:count:bin"])
def test_sample_in_several_read_batches(inputs, tmp_path, mode):
    """a sample whose reads reach the GPU in several batches (256 MB of bases each in production; a few kb here): the streams,
    k-mer totals, PartiInfo<5> counters and the histogram add up over the batches -- the run directory is the one-batch run's"""
    extra = ["--bloom-size", "1000000"] if mode.startswith("hash") else []
    one = run(inputs, tmp_path / "one", "--mode", mode, "--keep-tmp", "--hist", *extra)
    # (and the count of such a sample in groups of partitions: libkmx takes < 2^32 k-mers and bytes per call, the limit is lowered to
    #  a few hundred here so that every partition goes alone or in pairs)
    env = dict(os.environ, KMX_READ_BATCH_BYTES="3000", KMX_COUNT_GROUP_LIMIT="700")
    cmd = [KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / "many"), "--kmer-size", "31", "--hard-min", "1",
           "--nb-partitions", "4", "--repart-file", str(inputs / "fixture.minimRepart"), "--mode", mode, "--keep-tmp", "--hist", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    many = tmp_path / "many"
    checked = 0
    for sub in ("matrices", "merge_infos", "partition_infos", "histograms", "counts", "superkmers"):
        for root, _, files in os.walk(one / sub):
            for f in files:
                a = os.path.join(root, f); b = os.path.join(many, os.path.relpath(a, one))
                da, db = open(a, "rb").read(), open(b, "rb").read()
                if f == "SuperKmerBinInfoFile":      # (its second line is the directory)
                    da, db = da.split(b"\n", 2)[2], db.split(b"\n", 2)[2]
                assert da == db, os.path.relpath(a, one)
                checked += 1
    assert checked > 30


def test_restricted_run_statistics_cover_every_partition(inputs, tmp_path):
    """--restrict-to-list: the split counts EVERY partition (KmFillPartitions feeds PartiInfo before the writer drops the
    super-k-mers of unselected partitions, fill_partitions.hpp:59-105, io/superk_storage.hpp:301), so <id>.pinfo and PartiInfoFile
    are the full run's -- whichever way the sample reached the GPU: in one batch with its counts left in HBM, in one batch through
    count files, or in several read batches."""
    full = run(inputs, tmp_path / "full", "--mode", "kmer:count:bin", "--keep-tmp")
    variants = {"resident": ([], {}), "files": (["--keep-tmp"], {}), "batches": (["--keep-tmp"], {"KMX_READ_BATCH_BYTES": "3000"})}
    for name, (flags, env) in variants.items():
        cmd = [KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / name), "--kmer-size", "31", "--hard-min", "1",
               "--nb-partitions", "4", "--repart-file", str(inputs / "fixture.minimRepart"), "--mode", "kmer:count:bin", "--restrict-to-list", "1,3"] + flags
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        out = tmp_path / name
        assert sorted(os.listdir(out / "matrices")) == ["matrix_1.count", "matrix_3.count"]
        for s_ in ("D1", "D2"):
            assert open(out / "partition_infos" / f"{s_}.pinfo").read() == open(full / "partition_infos" / f"{s_}.pinfo").read(), (name, s_)
            assert open(out / "superkmers" / s_ / "PartiInfoFile").read() == open(full / "superkmers" / s_ / "PartiInfoFile").read(), (name, s_)
        for p in (1, 3):
            assert open(out / "matrices" / f"matrix_{p}.count", "rb").read() == open(full / "matrices" / f"matrix_{p}.count", "rb").read()


def test_parti_info_file(inputs, tmp_path):
    """superkmers/<id>/PartiInfoFile (gatb PartiInfo.hpp:266-287) from the HIP split's statistics == the oracle's PartiInfo<5>"""
    out = run(inputs, tmp_path / "run", "--mode", "kmer:count:bin", "--until", "superk")
    lut = orc.minimizer_lut(M); rep = repart_table()
    for name, f in (("D1", "1.fasta"), ("D2", "2.fasta")):
        pin, ms, mk, _ = orc.superk_stats(read_fasta(os.path.join(GD, f)), K, M, lut, rep, P)
        lines = [int(x) for x in open(out / "superkmers" / name / "PartiInfoFile").read().split()]
        assert len(lines) == 4 + P * 1282 + 3 * 4 ** M
        assert lines[:4] == [P, 4 ** M, int(ms.sum()), int(mk.sum())]
        assert lines[4:4 + P * 1282] == [int(x) for x in pin.reshape(-1)]
        tail = np.array(lines[4 + P * 1282:], dtype=np.uint64).reshape(-1, 3)
        assert np.array_equal(tail[:, 0], ms) and np.array_equal(tail[:, 1], mk) and not tail[:, 2].any()
        assert (out / "superkmers" / name / "skp.0").is_file()
    cfg = open(out / "config_gatb" / "gatb.config", "rb").read()
    assert len(cfg) == 140 and struct.unpack("<QQ", cfg[:16]) == (31, 10) and struct.unpack("<I", cfg[128:132]) == (4,)


def test_two_gpu_workers_same_output(tmp_path):
    """--gpus G shards x --gpu-workers W count workers (a host thread and a context each; on a one-GPU box all on device 0):
    samples go round-robin over the workers, partitions over the shards, workers are in flight at once (KMX_TRACE time stamps
    overlap), and the run directory is the one --gpus 1 writes -- with the count lists going through count files (--keep-tmp)
    or staying in HBM (a store per shard, kmx_count_reads_dev), with 8 shards (the shape of an 8-GPU node), and with stores so
    small that most samples are turned away and the merge takes resident lists and count files in one batch, and with several
    samples per count call (KMX_COUNT_SAMPLES_PER_CALL)"""
    reads = _synthetic_samples(tmp_path, 8, 200_000, 3)
    base_cmd = [KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--kmer-size", "31", "--nb-partitions", "16",
                "--static-repart", "--recurrence-min", "2", "--merge-batch-mb", "16"]
    runs = {"g1_files": (["--gpus", "1", "--gpu-workers", "1", "--keep-tmp"], {}),
            "g2_files": (["--gpus", "2", "--gpu-workers", "1", "--keep-tmp"], {}),
            "g1_resident": (["--gpus", "1", "--gpu-workers", "2"], {}),
            "g2_resident": (["--gpus", "2", "--gpu-workers", "2"], {"KMX_OUT_PIECE_KB": "64"}),
            "g2_resident_order": (["--gpus", "2", "--gpu-workers", "2"], {"KMX_OUT_PIECE_KB": "64", "KMX_OUT_ORDER": "1"}),
            "g8_resident": (["--gpus", "8", "--gpu-workers", "1"], {}),
            "g2_resident_multi": (["--gpus", "2", "--gpu-workers", "1"], {"KMX_COUNT_SAMPLES_PER_CALL": "4"}),      # several samples per count call (kmx_count_reads_dev_multi)
            "g1_resident_multi": (["--gpus", "1", "--gpu-workers", "2", "--mode", "kmer:count:bin"], {"KMX_COUNT_SAMPLES_PER_CALL": "3"}),
            "g3_mixed": (["--gpus", "3", "--gpu-workers", "1"], {"KMX_STORE_LIMIT_MB": "8"}),
            # round 4's defaults off: statistics by atomics, a sample's bases uploaded inside its call, ring pieces pinned when first needed
            "g1_resident_r3": (["--gpus", "1", "--gpu-workers", "2"], {"KMX_STATS_ATOMICS": "1", "KMX_READS_AHEAD": "0", "KMX_RING_PREFILL": "0"})}
    outs = {}
    for name, (flags, env) in runs.items():
        out = tmp_path / name
        r = subprocess.run(base_cmd + ["--run-dir", str(out)] + flags, capture_output=True, text=True, env=dict(os.environ, KMX_TRACE="1", **env))
        assert r.returncode == 0, r.stderr
        outs[name] = (out, r.stderr)
    ref = outs["g1_files"][0]
    for name, (out, err) in outs.items():
        for sub in ("matrices", "merge_infos", "partition_infos"):
            names = sorted(os.listdir(ref / sub))
            assert names == sorted(os.listdir(out / sub)) and names, (name, sub)
            for n in names:
                assert open(ref / sub / n, "rb").read() == open(out / sub / n, "rb").read(), (name, sub, n)
        rep = json.loads([l for l in err.splitlines() if l.startswith("[kmx pipeline]")][-1][len("[kmx pipeline] "):])
        if name.endswith("_multi"): assert rep["resident_count_calls"] < 8, rep      # (samples did share calls)
        if name.endswith("_files"): assert rep["resident_samples"] == 0
        elif "_resident" in name: assert rep["resident_samples"] == 8
        else: assert 0 < rep["resident_samples"] < 8, rep      # (some samples fitted the 8 MB stores, the others left count files)
        if not name.endswith("_files"):   # no count file survives a run without --keep-tmp
            assert all(not os.listdir(out / "counts" / f"partition_{p}") for p in range(16)), name
        if name == "g8_resident":
            # the 8-GPU shape: a sample's lists reach the stores in ONE copy per destination shard (the partitions bound for a GPU lie
            # back to back; over xGMI that is one hipMemcpyPeerAsync per destination GPU and sample) -- 8 samples x 8 stores, every
            # store by every sample; on this one-GPU box all of them device copies
            import re
            copies = re.findall(r"\[kmx copy\] lists from device (\d+) to store (\d+) on device (\d+): (\d+) bytes, (device copy|hipMemcpyPeerAsync)", err)
            assert len(copies) == 8 * 8, len(copies)
            assert sorted(int(c[1]) for c in copies) == sorted(list(range(8)) * 8)
            assert all(int(c[3]) > 0 for c in copies)
            assert rep["gpus"] == 8 and rep["peer_pairs"] >= 0
    for s_ in range(8):      # the PartiInfo<5> statistics: per partition from the sorted descriptors (default) == by atomics
        n = f"S{s_:04d}/PartiInfoFile"
        assert open(outs["g1_resident"][0] / "superkmers" / n, "rb").read() == open(outs["g1_resident_r3"][0] / "superkmers" / n, "rb").read(), n
    for p in range(16):
        for s_ in range(8):
            n = f"partition_{p}/S{s_:04d}.kmer"
            assert open(ref / "counts" / n, "rb").read() == open(outs["g2_files"][0] / "counts" / n, "rb").read()
    # overlap: some split/count interval of one worker intersects one of another
    for name in ("g2_files", "g2_resident"):
        ev, open_at = {}, {}
        for line in outs[name][1].splitlines():
            if not line.startswith("[kmx trace]"): continue
            t, g, what_, ident = line.split()[2:6]
            t = float(t); g = int(g.split("=")[1])
            base = what_.rsplit("_", 1)[0]
            if what_.endswith("_begin"): open_at[(g, base, ident)] = t
            elif what_.endswith("_end"): ev.setdefault(g, []).append((open_at.pop((g, base, ident)), t))
        assert len(ev) >= 2, name
        ws = sorted(ev)
        assert any(a0 < b1 and a1 < b0 for i in ws for j in ws if i < j for a0, b0 in ev[i] for a1, b1 in ev[j]), name


def _synthetic_samples(d, n_samples, genome_len, seed):
    """SURVEY 8d generator in small: one ancestor, per-sample substitutions, 150-bp error-free reads at 6x,
    random strand, FASTQ (half of the samples gzipped)."""
    rng = np.random.default_rng(seed)
    anc = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=genome_len)
    comp = np.zeros(256, np.uint8); comp[list(b"ACGTN")] = list(b"TGCAN")
    reads_all, fof = [], []
    for s in range(n_samples):
        g = anc.copy()
        mut = rng.random(genome_len) < 0.002
        g[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(mut.sum()))
        n = genome_len * 6 // 150
        starts = rng.integers(0, genome_len - 150, n)
        reads = []
        for i, st in enumerate(starts):
            r = g[st:st + 150]
            if rng.random() < 0.5: r = comp[r][::-1]
            r = r.copy()
            if i % 97 == 0: r[rng.integers(0, 150)] = ord("N")        # a few invalid k-mers
            reads.append(r.tobytes().decode())
        path = d / (f"S{s:04d}.fastq" + (".gz" if s % 2 else ""))
        op = gzip.open if s % 2 else open
        with op(path, "wt") as f:
            for i, r in enumerate(reads): f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
        reads_all.append(reads); fof.append(f"S{s:04d} : {path}")
    (d / "syn.fof").write_text("\n".join(fof) + "\n")
    return reads_all


@pytest.mark.parametrize("mode", ["kmer:count:bin", "hash:bf:bin"])
def test_synthetic_multi_sample_pipeline(tmp_path, mode):
    """BASELINE configs[1]/[2] in small (12 samples x 120 kbp, 8 static partitions, hard-min 2,
    recurrence-min 2): FASTQ(.gz) -> `kmx pipeline` -> every matrix body and merge_info bit-exact
    against the oracle run stage by stage on the same reads."""
    NS, GL, PP = 12, 120_000, 8
    reads = _synthetic_samples(tmp_path, NS, GL, 11)
    out = tmp_path / "run"
    args = [KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--run-dir", str(out), "--kmer-size", "31", "--hard-min", "2",
            "--nb-partitions", str(PP), "--static-repart", "--mode", mode, "--recurrence-min", "2", "--bloom-size", "2000000"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lut = orc.minimizer_lut(10); rep = orc.repart_static(10, PP)
    assert np.frombuffer(open(out / "repartition_gatb" / "repartition.minimRepart", "rb").read()[12:12 + 2 * 4 ** 10], np.uint16).tolist() == rep.tolist()
    sk = [orc.superk_partition(rs, 31, 10, lut, rep, PP) for rs in reads]
    W = ((2000000 + PP - 1) // PP + 63) // 64 * 64
    total_rows = 0
    for p in range(PP):
        if mode.startswith("kmer"):
            lists = [tuple(x if i else x.reshape(-1) for i, x in enumerate(orc.count_kmer(s[p][0], 31, 2))) for s in sk]
            body, rows, stats = orc.merge_matrix(lists, 1, [1] * NS, 2, 0, orc.MODE_COUNT)
            raw = open(out / "matrices" / f"matrix_{p}.count", "rb").read()
            assert raw[45:] == body
            total_rows += rows
        else:
            lists = [orc.count_hash(s[p][0], 31, W, p, 2) for s in sk]
            body, rows, stats = orc.merge_matrix(lists, 1, [1] * NS, 2, 0, orc.MODE_BF, W * p, W * (p + 1) - 1)
            raw = open(out / "matrices" / f"matrix_{p}.cmbf", "rb").read()
            assert raw[49:] == body
            total_rows += rows
        mi = [l.split("\t") for l in open(out / "merge_infos" / f"partition{p}.merge_info").read().splitlines()]
        for rix in range(6):
            assert [int(x) for x in mi[rix][1:1 + NS]] == [int(x) for x in stats[rix]]
    assert total_rows > 50_000


@pytest.mark.parametrize("KK,mode", [(64, "kmer:count:bin"), (80, "kmer:pa:bin"), (96, "hash:count:bin"), (97, "kmer:count:bin"), (127, "kmer:pa:bin"), (127, "hash:bf:bin")])
def test_wide_kmer_pipeline(tmp_path, KK, mode):
    """k = 64 ... 127 (the reference's default KMER_LIST "32 64 96 128", CMakeLists.txt:25-27: Kmer<96> / Kmer<128>, keys of ceil(k / 32)
    words) end to end: FASTQ(.gz) -> `kmx pipeline` (split by k_superk_wide, counts through the count files, merged by k_merge_rows) ->
    every matrix body, header and merge_info against the oracle run stage by stage on the same reads"""
    NS, GL, PP = 9, 30_000, 4
    reads = _synthetic_samples(tmp_path, NS, GL, 100 + KK)
    out = tmp_path / "run"
    args = [KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--run-dir", str(out), "--kmer-size", str(KK), "--hard-min", "2",
            "--nb-partitions", str(PP), "--static-repart", "--mode", mode, "--recurrence-min", "2", "--bloom-size", "400000"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    kw = orc.kw_of_k(KK)
    lut = orc.minimizer_lut(10); rep = orc.repart_static(10, PP)
    sk = [orc.superk_partition(rs, KK, 10, lut, rep, PP) for rs in reads]
    W = struct.unpack("<QQQQI", open(out / "hash.info", "rb").read())[2] if mode.startswith("hash") else 0
    total_rows = 0
    for p in range(PP):
        if mode.startswith("kmer"):
            lists = [tuple(x if i else x.reshape(-1) for i, x in enumerate(orc.count_kmer(s[p][0], KK, 2))) for s in sk]
            omode, ext = (orc.MODE_COUNT, "count") if mode == "kmer:count:bin" else (orc.MODE_PA, "pa")
            body, rows, stats = orc.merge_matrix(lists, kw, [1] * NS, 2, 0, omode)
            raw = open(out / "matrices" / f"matrix_{p}.{ext}", "rb").read()
            assert struct.unpack_from("<II", raw, 21) == (KK, kw) and raw[45:] == body
        elif mode == "hash:count:bin":
            lists = [orc.count_hash(s[p][0], KK, W, p, 2) for s in sk]
            body, rows, stats = orc.merge_matrix(lists, 1, [1] * NS, 2, 0, orc.MODE_COUNT)
            raw = open(out / "matrices" / f"matrix_{p}.count_hash", "rb").read()
            assert raw[37:] == body
        else:
            lists = [orc.count_hash(s[p][0], KK, W, p, 2) for s in sk]
            body, rows, stats = orc.merge_matrix(lists, 1, [1] * NS, 2, 0, orc.MODE_BF, W * p, W * (p + 1) - 1)
            raw = open(out / "matrices" / f"matrix_{p}.cmbf", "rb").read()
            assert raw[49:] == body
        total_rows += rows
        mi = [l.split("\t") for l in open(out / "merge_infos" / f"partition{p}.merge_info").read().splitlines()]
        for rix in range(6):
            assert [int(x) for x in mi[rix][1:1 + NS]] == [int(x) for x in stats[rix]]
    assert total_rows > 10_000
    if mode == "kmer:count:bin":      # the text dump of a matrix: the k-mers as strings (kmer.hpp:797-810), ascending
        d = subprocess.run([KMX, "dump", "--input", str(out / "matrices" / "matrix_0.count")], capture_output=True, text=True)
        if d.returncode == 0:
            first = d.stdout.splitlines()[0].split()
            assert len(first[0]) == KK and set(first[0]) <= set("ACGT") and len(first) == 1 + NS


@pytest.mark.parametrize("KK", [80, 127])
def test_wide_kmer_pipeline_through_files(tmp_path, KK):
    """k >= 64 with --keep-tmp --hist (super-k-mer and count files written and read back; the abundance histograms) and with --cpr:
    the matrices of the plain run; the count files hold the oracle's lists with ceil(k / 32)-word keys; histograms as KHist gives them"""
    NS, GL, PP = 4, 30_000, 4
    reads = _synthetic_samples(tmp_path, NS, GL, 300 + KK)
    base = [KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--kmer-size", str(KK), "--hard-min", "2", "--nb-partitions", str(PP), "--static-repart",
            "--mode", "kmer:count:bin"]
    for name, extra in (("plain", []), ("files", ["--keep-tmp", "--hist"]), ("cpr", ["--cpr"])):
        r = subprocess.run(base + ["--run-dir", str(tmp_path / name)] + extra, capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr)
    kw = orc.kw_of_k(KK)
    lut = orc.minimizer_lut(10); rep = orc.repart_static(10, PP)
    sk = [orc.superk_partition(rs, KK, 10, lut, rep, PP) for rs in reads]
    for p in range(PP):
        plain = open(tmp_path / "plain" / "matrices" / f"matrix_{p}.count", "rb").read()
        assert plain == open(tmp_path / "files" / "matrices" / f"matrix_{p}.count", "rb").read() and len(plain) > 45 + 1000 * (kw * 8 + 4 * NS)
        d = subprocess.run([KMX, "dump", "--input", str(tmp_path / "cpr" / "matrices" / f"matrix_{p}.count.lz4")], capture_output=True, text=True)
        d0 = subprocess.run([KMX, "dump", "--input", str(tmp_path / "plain" / "matrices" / f"matrix_{p}.count")], capture_output=True, text=True)
        assert d.returncode == 0 and d0.returncode == 0 and d.stdout == d0.stdout, (d.stderr, d0.stderr)
        first = d0.stdout.splitlines()[0].split()
        assert len(first[0]) == KK and set(first[0]) <= set("ACGT") and len(first) == 1 + NS
        for si in range(NS):
            f = kmfiles.read_kmer_file(tmp_path / "files" / "counts" / f"partition_{p}" / f"S{si:04d}.kmer")
            ek, ec = orc.count_kmer(sk[si][p][0], KK, 2)
            assert (f["k"], f["slots"]) == (KK, kw) and np.array_equal(f["keys"], ek) and np.array_equal(f["counts"], ec)
    for si in range(NS):
        exp, _ = _hist_bytes(KK, si, [orc.count_kmer(sk[si][p][0], KK, 1)[1] for p in range(PP)])
        assert open(tmp_path / "files" / "histograms" / f"S{si:04d}.hist", "rb").read() == exp


def test_wide_kmer_plugin_pipeline(tmp_path):
    """--plugin at k = 97 (rows of four-word keys reach process_kmer as the reference's Kmer<128>::get_data64() would hand them):
    the test plugin doubles sample 0's count of every row it keeps"""
    NS, GL, PP, KK = 3, 20_000, 4, 97
    reads = _synthetic_samples(tmp_path, NS, GL, 5)
    out = tmp_path / "run"
    plug = os.path.join(ROOT, "kmtricks_amd", "libkmx_test_plugin.so")
    r = subprocess.run([KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--run-dir", str(out), "--kmer-size", str(KK), "--hard-min", "1",
                        "--nb-partitions", str(PP), "--static-repart", "--mode", "kmer:count:bin", "--plugin", plug, "--plugin-config", "0"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    kw = orc.kw_of_k(KK)
    lut = orc.minimizer_lut(10); rep = orc.repart_static(10, PP)
    sk = [orc.superk_partition(rs, KK, 10, lut, rep, PP) for rs in reads]
    for p in range(PP):
        lists = [tuple(x if i else x.reshape(-1) for i, x in enumerate(orc.count_kmer(s[p][0], KK, 1))) for s in sk]
        body, rows, _ = orc.merge_matrix(lists, kw, [1] * NS, 0, 0, orc.MODE_COUNT)
        exp = np.frombuffer(body, np.uint8).reshape(rows, kw * 8 + 4 * NS).copy()
        c0 = exp[:, kw * 8:kw * 8 + 4].copy().view(np.uint32); c0 *= 2; exp[:, kw * 8:kw * 8 + 4] = c0.view(np.uint8)
        assert rows > 1000 and open(out / "matrices" / f"matrix_{p}.count", "rb").read()[45:] == exp.tobytes()


def test_k63_pa_cohort_pipeline(tmp_path):
    """BASELINE configs[4] in small (200 samples x 40 kbp, k = 63, `kmer:pa:bin`, recurrence-min 1, 4 partitions): the 128-bit-key
    build of the column-blocked merge runs the batches (KMX_TRACE names the kernel), every PA matrix and merge_info equals the oracle's"""
    NS, GL, PP, KK = 200, 40_000, 4, 63
    reads = _synthetic_samples(tmp_path, NS, GL, 23)
    out = tmp_path / "run"
    args = [KMX, "pipeline", "--file", str(tmp_path / "syn.fof"), "--run-dir", str(out), "--kmer-size", str(KK), "--hard-min", "2",
            "--nb-partitions", str(PP), "--static-repart", "--mode", "kmer:pa:bin", "--recurrence-min", "1"]
    r = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, KMX_TRACE="1"))
    assert r.returncode == 0, r.stderr
    ran = [l for l in r.stderr.splitlines() if l.startswith("[kmx merge] batch of")]
    assert ran and all("128-bit keys" in l and l.endswith("k_merge_cols") for l in ran), ran
    lut = orc.minimizer_lut(10); rep = orc.repart_static(10, PP)
    sk = [orc.superk_partition(rs, KK, 10, lut, rep, PP) for rs in reads]
    total_rows = 0
    for p in range(PP):
        lists = [tuple(x if i else x.reshape(-1) for i, x in enumerate(orc.count_kmer(s[p][0], KK, 2))) for s in sk]
        body, rows, stats = orc.merge_matrix(lists, 2, [1] * NS, 1, 0, orc.MODE_PA)
        raw = open(out / "matrices" / f"matrix_{p}.pa", "rb").read()
        assert struct.unpack_from("<II", raw, 21) == (KK, 2) and raw[45:] == body
        total_rows += rows
        mi = [l.split("\t") for l in open(out / "merge_infos" / f"partition{p}.merge_info").read().splitlines()]
        for rix in range(6):
            assert [int(x) for x in mi[rix][1:1 + NS]] == [int(x) for x in stats[rix]]
    assert total_rows > 100_000


def test_cohort_pipeline_where_the_limits_interact(tmp_path):
    """a cohort large enough for the column-blocked pair (256 samples x 300 kbp, k = 31, count rows, 32 partitions, rows in file
    order out of the kernels), run three ways whose run directories must be identical: (a) plain; (b) stores of 48 MB shared by two
    shards, a 4 MB output ring in 256 KB pieces, 40 000 k-mers per count call, three samples per call -- some samples resident, the
    others through count files, every batch a mix, bodies leaving in many pieces; (c) no resident lists at all (count files, as
    the reference).  Partition 0 of (a) equals the oracle's matrix."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    NS, GL, PP = 256, 300_000, 32
    paths = [bench._pipeline_make_sample((s_, GL, 0.002, 6, str(tmp_path))) for s_ in range(NS)]
    (tmp_path / "c.fof").write_text("".join(f"S{i:04d} : {p}\n" for i, p in enumerate(paths)))
    base = [KMX, "pipeline", "--file", str(tmp_path / "c.fof"), "--kmer-size", "31", "--hard-min", "2", "--nb-partitions", str(PP), "--static-repart",
            "--recurrence-min", "2", "--mode", "kmer:count:bin", "-t", "16"]
    runs = {"plain": ([], {"KMX_COLS_MIN_LISTS_ORD": "192"}),      # (libkmx's own choice for count rows in file order starts at 257 lists)
            "tight": (["--gpus", "2", "--samples-per-call", "3", "--merge-batch-mb", "64"], {"KMX_STORE_LIMIT_MB": "48", "KMX_OUT_RING_MB": "4", "KMX_OUT_PIECE_KB": "256", "KMX_COUNT_GROUP_LIMIT": "40000"}),
            "files": (["--no-resident"], {})}
    outs = {}
    for name, (flags, env) in runs.items():
        r = subprocess.run(base + ["--run-dir", str(tmp_path / name)] + flags, capture_output=True, text=True, env=dict(os.environ, KMX_TRACE="1", **env))
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = json.loads([l for l in r.stderr.splitlines() if l.startswith("[kmx pipeline]")][-1][len("[kmx pipeline] "):])
        if name == "plain":
            ran = [l for l in r.stderr.splitlines() if l.startswith("[kmx merge] batch of")]
            assert ran and any(l.endswith("k_merge_cols") for l in ran), ran      # (the first, one-partition batch may be too small for it)
    assert outs["plain"]["resident_samples"] == NS and outs["files"]["resident_samples"] == 0
    assert 0 < outs["tight"]["resident_samples"] < NS, outs["tight"]
    for sub in ("matrices", "merge_infos", "partition_infos"):
        names = sorted(os.listdir(tmp_path / "plain" / sub))
        assert names
        for other in ("tight", "files"):
            assert names == sorted(os.listdir(tmp_path / other / sub)), (other, sub)
            for n in names:
                assert open(tmp_path / "plain" / sub / n, "rb").read() == open(tmp_path / other / sub / n, "rb").read(), (other, sub, n)
    reads = [[bytes(r) for r in np.fromfile(p, np.uint8).reshape(-1, 154)[:, 3:153]] for p in paths]
    lut = orc.minimizer_lut(10); rep = orc.repart_static(10, PP)
    lists = []
    for rs in reads:
        sk = orc.superk_partition(rs, 31, 10, lut, rep, PP)
        k_, c_ = orc.count_kmer(sk[0][0], 31, 2)
        lists.append((k_.reshape(-1), c_))
    body, rows, _ = orc.merge_matrix(lists, 1, [1] * NS, 2, 0, orc.MODE_COUNT)
    assert rows > 5000 and open(tmp_path / "plain" / "matrices" / "matrix_0.count", "rb").read()[45:] == body


def test_combine_two_runs_equals_one_run(inputs, tmp_path):
    """`kmx combine` over two `kmx pipeline` runs that share a repartition (one sample each) == the matrix of one run over both
    samples (recurrence-min 1) -- with --reference-compat up to MatrixMerger's dropped last key (matrix.hpp:534-583); also with
    a run that kept its count files (taken as one-column matrices, matrix.hpp:756-771)."""
    fofs = []
    for i, line in enumerate((f"D1 : {GD}/1.fasta", f"D2 : {inputs}/2.fastq.gz")):
        f = tmp_path / f"s{i}.fof"; f.write_text(line + "\n"); fofs.append(f)
    def pipe(fof, out, *extra):
        r = subprocess.run([KMX, "pipeline", "--file", str(fof), "--run-dir", str(out), "--kmer-size", "31", "--hard-min", "1", "--nb-partitions", "4",
                            "--repart-file", str(inputs / "fixture.minimRepart"), "--mode", "kmer:count:bin", "--recurrence-min", "1", *extra], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return out
    a = pipe(fofs[0], tmp_path / "runA"); b = pipe(fofs[1], tmp_path / "runB"); bk = pipe(fofs[1], tmp_path / "runBk", "--keep-tmp")
    both = run(inputs, tmp_path / "both", "--mode", "kmer:count:bin", "--recurrence-min", "1")
    for name, second, compat in (("c1", b, False), ("c2", bk, False), ("c3", b, True)):
        lst = tmp_path / f"{name}.fof"; lst.write_text(f"{a}\n{second}\n")
        out = tmp_path / name
        r = subprocess.run([KMX, "combine", "--fof", str(lst), "--output", str(out)] + (["--reference-compat"] if compat else []), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out / "kmtricks.fof").read().split() == open(both / "kmtricks.fof").read().split()
        for p in range(P):
            full = np.frombuffer(open(both / "matrices" / f"matrix_{p}.count", "rb").read()[45:], np.uint8).reshape(-1, 16)
            raw = open(out / "matrices" / f"matrix_{p}.count", "rb").read()
            got = np.frombuffer(raw[45:], np.uint8).reshape(-1, 16)
            assert struct.unpack_from("<IIII", raw, 21)[3] == 2                      # two columns
            last_in_one_run_only = int((full[-1, 8:12].view(np.uint32)[0] == 0) or (full[-1, 12:16].view(np.uint32)[0] == 0))
            assert len(got) == len(full) - (last_in_one_run_only if compat else 0)
            assert np.array_equal(got, full[:len(got)]), p


def test_dump_and_aggregate(inputs, tmp_path):
    """`kmx dump` / `kmx aggregate` (cmd.hpp:275-369, 441-607) on a run directory: matrix rows as text, partitions concatenated or merged
    into one ascending stream, aggregated binary matrix"""
    out = run(inputs, tmp_path / "run", "--mode", "kmer:count:bin", "--keep-tmp")
    lists = oracle_lists(False)
    rows_all = []
    for p in range(P):
        body, rows, _ = orc.merge_matrix(lists[p], 1, [1, 1], 1, 0, orc.MODE_COUNT)
        a = np.frombuffer(body, np.uint8).reshape(rows, 16)
        part = [(int(a[i, :8].view(np.uint64)[0]), int(a[i, 8:12].view(np.uint32)[0]), int(a[i, 12:16].view(np.uint32)[0])) for i in range(rows)]
        r = subprocess.run([KMX, "dump", "--input", str(out / "matrices" / f"matrix_{p}.count")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == "".join(f"{orc.kmer_to_string(np.array([k], np.uint64), 31)} {c0} {c1}\n" for k, c0, c1 in part)
        rows_all.append(part)
    flat = [x for part in rows_all for x in part]
    r = subprocess.run([KMX, "aggregate", "--run-dir", str(out), "--matrix", "kmer"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "".join(f"{orc.kmer_to_string(np.array([k], np.uint64), 31)} {c0} {c1}\n" for k, c0, c1 in flat)
    r = subprocess.run([KMX, "aggregate", "--run-dir", str(out), "--matrix", "kmer", "--sorted", "--no-count"], capture_output=True, text=True)
    assert r.stdout == "".join(f"{orc.kmer_to_string(np.array([k], np.uint64), 31)}\n" for k, _, _ in sorted(flat))
    agg = tmp_path / "all.count"
    r = subprocess.run([KMX, "aggregate", "--run-dir", str(out), "--matrix", "kmer", "--sorted", "--format", "bin", "--output", str(agg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(agg, "rb").read()
    assert len(raw) == 45 + 16 * len(flat)
    got = np.frombuffer(raw[45:], np.uint8).reshape(-1, 16)
    assert [int(x) for x in got[:, :8].copy().view(np.uint64).reshape(-1)] == [k for k, _, _ in sorted(flat)]
    r = subprocess.run([KMX, "aggregate", "--run-dir", str(out), "--count", "D1:kmer", "--sorted"], capture_output=True, text=True)
    assert r.returncode == 0 and len(r.stdout.splitlines()) == sum(G["task_main"]["superk_info_D1"][1::2])


def _text_rows(body, kw, n, hash_mode, k=K):
    """the oracle's count rows (key + n x u32) as write_as_text / write_as_pa_text print them (merge.hpp:288-316, 531-572): the k-mer as
    letters (Kmer::to_string, kmer.hpp:541-550) or the hash in decimal, then the counts"""
    rb = 8 * kw + 4 * n
    out = []
    for r in range(len(body) // rb):
        row = body[r * rb:(r + 1) * rb]
        key = int.from_bytes(row[:8 * kw], "little")
        name = str(key) if hash_mode else "".join("ACTG"[(key >> (2 * d)) & 3] for d in range(k - 1, -1, -1))
        out.append((name, list(struct.unpack(f"<{n}I", row[8 * kw:]))))
    return out


@pytest.mark.parametrize("mode", ["kmer:count:text", "kmer:pa:text", "hash:count:text", "hash:pa:text"])
def test_text_modes(inputs, tmp_path, mode):
    """the four :text modes (src/cli.cpp:151-157): matrix_<p>.<ext>.txt, no header, a line per kept row -- key, then " <count>" or
    " 0" / " 1" per sample (merge.hpp:288-316, 531-572); the rows are the :bin run's rows"""
    hash_mode, what = mode.startswith("hash"), mode.split(":")[1]
    out = run(inputs, tmp_path / "run", "--mode", mode, "--recurrence-min", "1", "--soft-min", "1", "--bloom-size", "1000000")
    opts = open(out / "options.txt").read()
    assert f"mode={what}," in opts and "format=text," in opts
    W = ((1000000 + P - 1) // P + 63) // 64 * 64
    lists = oracle_lists(hash_mode, W)
    ext = {"count": "count", "pa": "pa"}[what] + ("_hash" if hash_mode else "") + ".txt"
    assert sorted(os.listdir(out / "matrices")) == sorted(f"matrix_{p}.{ext}" for p in range(P))
    for p in range(P):
        body, rows, stats = orc.merge_matrix(lists[p], 1, [1, 1], 1, 0, orc.MODE_COUNT)
        exp = "".join(name + "".join(" " + (str(c) if what == "count" else ("1" if c else "0")) for c in cnt) + "\n" for name, cnt in _text_rows(body, 1, 2, hash_mode))
        got = open(out / "matrices" / f"matrix_{p}.{ext}").read()
        assert got == exp and got.count("\n") == rows
        if not hash_mode:
            assert rows == G["merge_test"]["kmer_rows"][p]
        mi = open(out / "merge_infos" / f"partition{p}.merge_info").read().splitlines()
        assert [int(x) for x in mi[2].split("\t")[1:3]] == [int(x) for x in stats[2]]


def _fixture_run_dir(d, kind):
    """a run directory as `kmtricks pipeline --until count --keep-tmp` leaves it, made of the REFERENCE's committed count files
    (tests/data/partitions/{kmers,hashes}: files of a MAX_C = 255 build, 1-byte counts)"""
    import shutil
    os.makedirs(d)
    shutil.copy(os.path.join(GD, "kmtricks.fof"), d / "kmtricks.fof")
    shutil.copy(os.path.join(GD, "hash.info"), d / "hash.info")
    for p in range(P):
        os.makedirs(d / "counts" / f"partition_{p}")
        for s in ("D1", "D2"):
            shutil.copy(os.path.join(GD, "partitions", kind + "s" if kind == "kmer" else "hashes", f"partition_{p}", f"{s}.{kind}"), d / "counts" / f"partition_{p}" / f"{s}.{kind}")
    (d / "options.txt").write_text(f"Options: dir={d}, verbosity=info, nb_threads=8, fof=x, kmer_size=31, c_ab_min=1, m_ab_min=1, r_min=1, save_if=0, minim_size=10, "
                                   f"nb_parts=4, bloom_size=1000192, keep_tmp=1, lz4=0, mode=count, format=bin, count_format={kind}, until=count\n")
    return d


@pytest.mark.parametrize("kind", ["kmer", "hash"])
def test_merge_module_over_the_reference_s_count_files(tmp_path, kind):
    """`kmx merge --run-dir` (the reference's `kmtricks merge`, src/cli.cpp:526-646) over the count files the REFERENCE committed with its
    tests -- 1-byte counts (count_slots 1: a MAX_C = 255 build, CMakeLists.txt:25-41): the 57 / 67 / 70 / 82 rows of
    tests/merge_test.cpp:21-39, bodies = the oracle's over the same files, and `kmx dump` reads them back"""
    d = _fixture_run_dir(tmp_path / "run", kind)
    hash_mode = kind == "hash"
    sub = "kmers" if kind == "kmer" else "hashes"
    one = (kmfiles.read_kmer_file if kind == "kmer" else kmfiles.read_hash_file)(f"{GD}/partitions/{sub}/partition_0/D1.{kind}")
    assert one["count_slots"] == 1
    r = subprocess.run([KMX, "merge", "--run-dir", str(d), "--mode", f"{kind}:count:bin", "--recurrence-min", "1", "--soft-min", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows_exp = G["merge_test"]["kmer_rows" if kind == "kmer" else "hash_rows"]
    for p in range(P):
        fs = [(kmfiles.read_kmer_file if kind == "kmer" else kmfiles.read_hash_file)(f"{GD}/partitions/{sub}/partition_{p}/{s}.{kind}") for s in ("D1", "D2")]
        lists = [(np.ascontiguousarray(f["keys"]).reshape(-1), f["counts"].astype(np.uint32)) for f in fs]
        body, rows, stats = orc.merge_matrix(lists, 1, [1, 1], 1, 0, orc.MODE_COUNT)
        raw = open(d / "matrices" / f"matrix_{p}.{'count_hash' if hash_mode else 'count'}", "rb").read()
        assert rows == rows_exp[p] and raw[(37 if hash_mode else 45):] == body
        assert (d / "counts" / f"partition_{p}" / f"D1.{kind}").exists()      # (no --clear: the partition files stay)
        mi = open(d / "merge_infos" / f"partition{p}.merge_info").read().splitlines()
        assert [int(x) for x in mi[2].split("\t")[1:3]] == [int(x) for x in stats[2]]
    dump = subprocess.run([KMX, "dump", "--input", str(d / "matrices" / f"matrix_0.{'count_hash' if hash_mode else 'count'}")], capture_output=True, text=True)
    assert dump.returncode == 0 and dump.stdout.count("\n") == rows_exp[0]
    # --partition-id: one partition; --clear: its count files go; a directory that is no run directory is refused
    d2 = _fixture_run_dir(tmp_path / "run2", kind)
    r = subprocess.run([KMX, "merge", "--run-dir", str(d2), "--mode", f"{kind}:pa:text", "--partition-id", "2", "--clear"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.listdir(d2 / "matrices") == [f"matrix_2.{'pa_hash' if hash_mode else 'pa'}.txt"]
    assert not (d2 / "counts" / "partition_2" / f"D1.{kind}").exists() and (d2 / "counts" / "partition_1" / f"D1.{kind}").exists()
    assert open(d2 / "matrices" / os.listdir(d2 / "matrices")[0]).read().count("\n") == rows_exp[2]
    r = subprocess.run([KMX, "merge", "--run-dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "not a kmtricks runtime directory" in r.stderr
