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


def test_cli_errors(inputs, tmp_path):
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "already exists" in r.stderr          # src/cli.cpp:101-104
    r = subprocess.run([KMX, "pipeline", "--file", str(inputs / "in.fof"), "--run-dir", str(tmp_path / "x"), "--mode", "hash:bft:bin"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "not supported" in r.stderr
