"""CPU-only: `kmx dump` on the reference's committed count files (text forms of io/kmer_file.hpp:140-148 and
io/hash_file.hpp:211-219: `ACGT... count` / `hash count` per line), and the example plugins of the reference compiled UNCHANGED
against the shipped include/kmtricks headers (only in the build container, where /root/reference exists; nothing of it ships)."""
import glob, os, subprocess
import numpy as np
import pytest

import orc
import kmfiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMX = os.path.join(ROOT, "kmtricks_amd", "kmx")
GD = os.path.join(ROOT, "tests", "golden")


def test_dump_kmer_and_hash_files():
    for p in range(4):
        path = f"{GD}/partitions/kmers/partition_{p}/D1.kmer"
        f = kmfiles.read_kmer_file(path)
        r = subprocess.run([KMX, "dump", "--input", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        exp = "".join(f"{orc.kmer_to_string(k, 31)} {int(c)}\n" for k, c in zip(f["keys"].reshape(-1, 1), f["counts"]))
        assert r.stdout == exp
        path = f"{GD}/partitions/hashes/partition_{p}/D2.hash"
        f = kmfiles.read_hash_file(path)
        r = subprocess.run([KMX, "dump", "--input", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == "".join(f"{int(h)} {int(c)}\n" for h, c in zip(f["keys"], f["counts"]))
    r = subprocess.run([KMX, "dump", "--input", f"{GD}/hash.info"], capture_output=True, text=True)
    assert r.returncode == 1 and "[error]" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/plugins/example"), reason="the reference tree is only present in the build container")
def test_reference_example_plugins_compile_unchanged(tmp_path):
    """plugins/example/*.cpp of the reference against include/kmtricks/{plugin,kmer,utils}.hpp (compile + the exported symbols the
    plugin manager looks up, plugin_manager.hpp:38-113)"""
    for src in sorted(glob.glob("/root/reference/plugins/example/*.cpp")):
        so = tmp_path / (os.path.basename(src)[:-4] + ".so")
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-DDMAX_C=4294967295", f"-I{ROOT}/include", src, "-o", str(so)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
        assert " plugin_name" in syms and " use_template" in syms
        assert (" create0" in syms) or (" create32" in syms and " create64" in syms)


def test_shipped_kmer_header_against_the_oracle(tmp_path):
    """include/kmtricks/kmer.hpp (the value type plugins use): strings, comparisons, reverse complement, canonical form and
    minimizer == the oracle's restatements (gatb Model.hpp:857-884, 1220-1287; kmer.hpp:520-632), k = 21, 31, 32, 47, 63"""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <kmtricks/kmer.hpp>
#include <iostream>
template <size_t MK> void run(const std::string& s, unsigned m) {
  km::Kmer<MK> a(s);
  km::Kmer<MK> rc = a.rev_comp(), ca = a.canonical();
  km::Kmer<MK> b(a.k()); b.set64_p(a.get_data64());
  std::cout << a.to_string() << " " << rc.to_string() << " " << ca.to_string() << " " << (b == a) << (a < rc) << (a <= a) << (rc > a) << " "
            << a.minimizer(m).value() << " " << a.at(0) << a.at(a.k() - 1) << " " << ((a << 2) >> 2 == a ? 1 : 0) << " " << (a + 5 - 5 == a) << " " << ((a * 3) / 3 == a) << "\n";
}
int main(int argc, char** argv) {
  for (int i = 1; i + 1 < argc; i += 2) { std::string s = argv[i]; unsigned m = std::stoul(argv[i + 1]); if (s.size() < 32) run<32>(s, m); else run<64>(s, m); }
}
''')
    exe = tmp_path / "t"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", f"-I{ROOT}/include", str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(5)
    args, cases = [], []
    for k in (21, 31, 32, 47, 63):
        for _ in range(6):
            s = "".join(rng.choice(list("ACGT"), size=k))
            m = 10 if k > 21 else 8
            args += [s, str(m)]; cases.append((s, m))
    cases.append(("A" * 31, 10)); args += ["A" * 31, "10"]
    cases.append(("ACGT" * 8, 10)[:2]); args += ["ACGT" * 8, "10"]
    out = subprocess.run([str(exe)] + args, capture_output=True, text=True).stdout.splitlines()
    assert len(out) == len(cases)
    for line, (s, m) in zip(out, cases):
        f = line.split()
        k = len(s)
        w = orc.kmer_from_string(s)
        rc = orc.revcomp(w, k)
        rcs = orc.kmer_to_string(rc, k)
        less = tuple(int(x) for x in w[::-1]) < tuple(int(x) for x in rc[::-1])
        canon = s if less else rcs
        assert f[0] == s and f[1] == rcs and f[2] == canon
        assert f[3] == "1" + ("1" if less else "0") + "1" + ("1" if less else "0")
        lut = orc.minimizer_lut(m)
        assert int(f[4]) == orc.minimizer_of(w, k, m, lut)      # (gatb's rescan rule and the plain minimum agree on the value)
        assert int(f[4]) == min(int(lut[int(x)]) for x in _mmers(s, m))
        assert f[5] == s[0] + s[-1]
        assert f[6] == "1" and f[7] == "1" and f[8] == "1"


def test_shipped_kmer_header_against_the_reference_vectors(tmp_path):
    """the values tests/kmer_test.cpp:9-152 asserts (tests/golden/reference_goldens.json "kmer_test"), on the shipped
    include/kmtricks/kmer.hpp: type names, to_string / at / rev_comp identities on random strings of 20, 40 and 90 nucleotides
    (Kmer<32>, <64>, <92>), canonical forms, comparisons for one, two and three words, the m-mers and the minimizer"""
    import json
    Gk = json.load(open(os.path.join(GD, "reference_goldens.json")))["kmer_test"]
    src = tmp_path / "v.cpp"
    src.write_text(r'''
#include <kmtricks/kmer.hpp>
#include <iostream>
template <size_t MK> void ident(const std::string& s) {
  km::Kmer<MK> a(s); std::string at; for (size_t i = 0; i < s.size(); i++) at.push_back(a.at(i));
  std::cout << "ident " << km::Kmer<MK>::name() << "|" << a.to_string() << "|" << at << "|" << a.rev_comp().to_string() << "|" << km::str_rev_comp(s) << "\n";
}
template <size_t MK> void less(const std::string& x, const std::string& y) {
  km::Kmer<MK> a(x), b(y);
  std::cout << "less " << (a < b) << (a > b) << (a == b) << (a != b) << (a == a) << "\n";
}
int main(int argc, char** argv) {
  for (int i = 1; i < argc; i++) {
    const std::string c = argv[i];
    if (c == "ident") { const size_t mk = std::stoul(argv[i + 1]); const std::string s = argv[i + 2]; i += 2; if (mk == 32) ident<32>(s); else if (mk == 64) ident<64>(s); else ident<92>(s); }
    else if (c == "canon") { km::Kmer<32> a(argv[++i]); std::cout << "canon " << a.canonical().to_string() << "\n"; }
    else if (c == "less") { const size_t mk = std::stoul(argv[i + 1]); const std::string x = argv[i + 2], y = argv[i + 3]; i += 3; if (mk == 32) less<32>(x, y); else if (mk == 64) less<64>(x, y); else less<96>(x, y); }
    else if (c == "mmers") { km::Kmer<32> a(argv[i + 1]); const unsigned m = std::stoul(argv[i + 2]); i += 2; std::cout << "mmers"; for (auto& v : a.mmers(m)) std::cout << " " << v.to_string(); std::cout << " | " << a.minimizer(m).to_string() << "\n"; }
  }
}
''')
    exe = tmp_path / "v"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", f"-I{ROOT}/include", str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(11)
    rnd = {mk: "".join(rng.choice(list("ACGT"), size=n)) for mk, n in ((32, 20), (64, 40), (92, 90))}      # (kmer_test.cpp:11-14: random_dna_seq(20 / 40 / 90))
    args = []
    for mk, _ in Gk["names"]:
        args += ["ident", str(mk), rnd[mk]]
    for a, _ in Gk["canonical"]:
        args += ["canon", a]
    for mk, x, y in Gk["less"]:
        args += ["less", str(mk), x, y]
    mi = Gk["minimizer"]
    args += ["mmers", mi["kmer"], str(mi["m"])]
    out = subprocess.run([str(exe)] + args, capture_output=True, text=True).stdout.splitlines()
    it = iter(out)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for mk, name in Gk["names"]:
        f = next(it)[len("ident "):].split("|")
        s = rnd[mk]
        assert f[0] == name and f[1] == s and f[2] == s
        assert f[3] == "".join(comp[c] for c in reversed(s)) == f[4]
    for _, c in Gk["canonical"]:
        assert next(it) == "canon " + c
    for _ in Gk["less"]:
        assert next(it) == "less 10011"      # a < b, !(a > b), !(a == b), a != b, a == a (kmer_test.cpp:97-115)
    assert next(it) == "mmers " + " ".join(mi["mmers"]) + " | " + mi["minimizer"]


def _mmers(s, m):
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    for i in range(len(s) - m + 1):
        v = 0
        for c in s[i:i + m]:
            v = (v << 2) | code[c]
        yield v


# ---- kmx combine (matrix.hpp:396-886) -------------------------------------------------------------------------------------------
import struct

KM_BASE = struct.pack("<QIB", 0x736b636972746d6b, 0, 0)


def _magics():
    """the type magics as the shipped headers' library writes them: read back from a file `kmx` wrote would be circular, so they
    are taken from kmx_io.hpp's constants (which the pipeline tests pin against the reference's committed files)"""
    import re
    src = open(os.path.join(ROOT, "kmtricks_amd", "host", "kmx_io.hpp")).read()
    return {n.lower(): int(v, 16) for n, v in re.findall(r"MAGIC_(\w+) = (0x[0-9a-fA-F]+)ULL", src)}


def _write_matrix(path, kind, k, ncols, sid, part, keys, data):
    """keys: uint64[n, slots] (hash: slots = 1); data: uint32[n, ncols] counts or uint8[n, ceil(ncols / 8)] PA bytes"""
    M = _magics()
    slots = (k + 31) // 32
    if kind == "count": hdr = struct.pack("<QIIIIII", M["matrix"], k, slots, 1, ncols, sid, part)
    elif kind == "count_hash": hdr = struct.pack("<QIIII", M["matrix_hash"], 4, ncols, sid, part)
    elif kind == "pa": hdr = struct.pack("<QIIIIII", M["pa"], k, slots, ncols, (ncols + 7) // 8, sid, part)
    else: hdr = struct.pack("<QIIII", M["pa_hash"], ncols, (ncols + 7) // 8, sid, part)
    with open(path, "wb") as f:
        f.write(KM_BASE + hdr)
        for kk, d in zip(keys, data):
            f.write(np.ascontiguousarray(kk, dtype=np.uint64).tobytes() + np.ascontiguousarray(d).tobytes())


def _partition_merger(files, pa, compat=True):
    """PartitionMerger::next / write loop restated (matrix.hpp:534-583, 632-680): files = [(keys as tuples most significant word
    first, rows of column values or bits, ncols)].  -> [(key, merged columns)].  The row whose first file empties the queue is
    not handed out."""
    import heapq
    pos, total = [], 0
    for _, _, n in files: pos.append(total); total += n
    cur = [0] * len(files)
    heap = [(f[0][0], i) for i, f in enumerate(files) if len(f[0])]
    heapq.heapify(heap)
    out = []
    def adv(i):
        cur[i] += 1
        if cur[i] < len(files[i][0]): heapq.heappush(heap, (files[i][0][cur[i]], i))
    while heap:
        row = [0] * total
        key, i = heapq.heappop(heap)
        for c, v in enumerate(files[i][1][cur[i]]): row[pos[i] + c] = v
        adv(i)
        if not heap and compat: break
        while heap and heap[0][0] == key:
            _, j = heapq.heappop(heap)
            for c, v in enumerate(files[j][1][cur[j]]): row[pos[j] + c] = v
            adv(j)
        out.append((key, row))
    return out


def _make_run(root, kind, k, P, ncols, seed, ids, hashed, shared_pool):
    rng = np.random.default_rng(seed)
    os.makedirs(f"{root}/matrices"); os.makedirs(f"{root}/repartition_gatb"); os.makedirs(f"{root}/config_gatb"); os.makedirs(f"{root}/counts/partition_0")
    open(f"{root}/repartition_gatb/repartition.minimRepart", "wb").write(b"same table" * 10)
    open(f"{root}/config_gatb/gatb.config", "wb").write(b"cfg")
    open(f"{root}/hash.info", "wb").write(struct.pack("<QQQQQ", 1000, P, 10, 2, 10))
    mode = "pa" if kind.startswith("pa") else "count"
    open(f"{root}/options.txt", "w").write(f"Options: dir={root}, verbosity=info, nb_threads=1, mode={mode}, format=bin, bf_format=howdesbt, count_format={'hash' if hashed else 'kmer'}, until=all\n")
    open(f"{root}/kmtricks.fof", "w").write("".join(f"{i}: /x/{i}.fa\n" for i in ids))
    slots = 1 if hashed else (k + 31) // 32
    parts = []
    for p in range(P):
        pool = shared_pool[p]
        take = np.sort(rng.choice(len(pool), size=rng.integers(0, len(pool) + 1), replace=False))
        keys = [pool[i] for i in take]
        if mode == "count": data = [rng.integers(0, 1000, ncols, dtype=np.uint32) for _ in keys]
        else: data = [np.packbits(rng.integers(0, 2, ncols, dtype=np.uint8), bitorder="little") for _ in keys]
        ext = {"count": "count", "count_hash": "count_hash", "pa": "pa", "pa_hash": "pa_hash"}[kind]
        _write_matrix(f"{root}/matrices/matrix_{p}.{ext}", kind, k, ncols, 7, p, [np.array(kk[::-1], dtype=np.uint64) for kk in keys], data)
        parts.append((keys, data))
    return parts


@pytest.mark.parametrize("kind,k,compat", [("count", 31, True), ("pa", 40, True), ("count_hash", 31, True), ("pa_hash", 31, True), ("count", 31, False), ("pa", 40, False)])
def test_combine_matrices_of_three_runs(tmp_path, kind, k, compat):
    """`kmx combine`: per partition, the rows of the runs' matrices joined by key, a run's columns behind the previous run's,
    zeros where a run lacks the key -- against a restatement of PartitionMerger, with its quirks: matrices/ entries are
    taken in NAME order (12 partitions: matrix_10 comes third), header fields come from the last run's file, and -- with
    --reference-compat only -- the last key of a partition is lost unless two runs hold it (kmx keeps it by default)."""
    P, hashed, pa = 12, kind.endswith("hash"), kind.startswith("pa")
    slots = 1 if hashed else (k + 31) // 32
    rng = np.random.default_rng(17)
    # per partition: a pool of ascending keys (tuples, most significant word first)
    pools = []
    for p in range(P):
        # (two-word keys: few distinct high words, so that the low word decides often)
        ks = {tuple(int(rng.integers(0, 1 << (4 if (slots == 2 and w == 0) else 40))) for w in range(slots)) for _ in range(25)}
        pools.append(sorted(ks))
    ncols = [3, 11, 6]
    runs, parts = [], []
    for r in range(3):
        root = str(tmp_path / f"run{r}"); runs.append(root)
        parts.append(_make_run(root, kind, k, P, ncols[r], 100 + r, [f"S{r}a", f"S{r}b"], hashed, pools))
    fof = tmp_path / "runs.fof"; fof.write_text("\n".join(runs) + "\n\n")
    out = str(tmp_path / "combined")
    r = subprocess.run([KMX, "combine", "--fof", str(fof), "--output", out] + (["--reference-compat"] if compat else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(f"{out}/kmtricks.fof").read() == "".join(f"S{r}{x}: /x/S{r}{x}.fa\n" for r in range(3) for x in "ab")
    assert open(f"{out}/options.txt").read() == open(f"{runs[0]}/options.txt").read() and os.path.exists(f"{out}/hash.info")
    assert open(f"{out}/repartition_gatb/repartition.minimRepart", "rb").read() == b"same table" * 10
    names = sorted(os.listdir(f"{runs[0]}/matrices"))
    total = sum(ncols)
    M = _magics()
    for p in range(P):
        src_p = int(names[p].split("_")[1].split(".")[0])                 # the partition the p-th name holds
        files = []
        for rr in range(3):
            keys, data = parts[rr][src_p]
            rows = [([int(x) for x in d] if not pa else [int(b) for b in np.unpackbits(d, bitorder="little")[:ncols[rr]]]) for d in data]
            files.append((keys, rows, ncols[rr]))
        exp = _partition_merger(files, pa, compat)
        raw = open(f"{out}/matrices/matrix_{p}.{kind}", "rb").read()
        assert raw[:13] == KM_BASE
        if kind == "count": hdr = struct.pack("<QIIIIII", M["matrix"], k, slots, 1, total, 7, src_p); hl = 45
        elif kind == "count_hash": hdr = struct.pack("<QIIII", M["matrix_hash"], 4, total, 7, src_p); hl = 37
        elif kind == "pa": hdr = struct.pack("<QIIIIII", M["pa"], k, slots, total, (total + 7) // 8, 7, src_p); hl = 45
        else: hdr = struct.pack("<QIIII", M["pa_hash"], total, (total + 7) // 8, 7, src_p); hl = 37
        assert raw[13:hl] == hdr
        body = b""
        for key, row in exp:
            body += np.array(key[::-1], dtype=np.uint64).tobytes()
            body += (np.array(row, dtype=np.uint32).tobytes() if not pa else np.packbits(np.array(row, dtype=np.uint8), bitorder="little").tobytes())
        assert raw[hl:] == body, (p, len(exp))
        n_all = len({kk for f in files for kk in f[0]})
        assert len(exp) in ((n_all, n_all - 1) if compat else (n_all,))
    # the text form of a combined matrix through `kmx dump`
    r = subprocess.run([KMX, "dump", "--input", f"{out}/matrices/matrix_0.{kind}"], capture_output=True, text=True)
    assert r.returncode == 0 and len(r.stdout.splitlines()) == len(_partition_merger(
        [(parts[rr][int(names[0].split('_')[1].split('.')[0])][0], [[0] * ncols[rr]] * len(parts[rr][int(names[0].split('_')[1].split('.')[0])][0]), ncols[rr]) for rr in range(3)], pa, compat))


def test_combine_renames_duplicate_ids_and_checks_the_repartition(tmp_path):
    pools = [sorted({(int(x),) for x in np.random.default_rng(3).integers(0, 1 << 40, 20)}) for _ in range(2)]
    runs = []
    for r in range(2):
        root = str(tmp_path / f"run{r}"); runs.append(root)
        _make_run(root, "count", 31, 2, 2, 5 + r, ["A", "B"], False, pools)
    fof = tmp_path / "runs.fof"; fof.write_text("\n".join(runs) + "\n")
    out = str(tmp_path / "c1")
    r = subprocess.run([KMX, "combine", "--fof", str(fof), "--output", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(f"{out}/kmtricks.fof").read() == "A_0:  /x/A.fa\nB_0:  /x/B.fa\nA_1:  /x/A.fa\nB_1:  /x/B.fa\n"      # (cat_fof_and_rename, matrix.hpp:841-864)
    open(f"{runs[1]}/repartition_gatb/repartition.minimRepart", "wb").write(b"another table")
    r = subprocess.run([KMX, "combine", "--fof", str(fof), "--output", str(tmp_path / "c2")], capture_output=True, text=True)
    assert r.returncode == 1 and "are not mergeable" in r.stderr
    open(f"{runs[0]}/options.txt", "w").write("Options: mode=bf, count_format=hash\n")
    r = subprocess.run([KMX, "combine", "--fof", str(fof), "--output", str(tmp_path / "c3")], capture_output=True, text=True)
    assert r.returncode == 1 and "not supported by 'kmtricks combine'" in r.stderr


def test_fasta_reader_views_equal_records(tmp_path):
    """round 6: the pipeline's readers take a short read as a VIEW into the read block (SeqReader::next_view, kmx_io.hpp: one copy, to
    the page-locked batch); it must cut every file into the records SeqReader::next does -- one-line and multi-line FASTA, CRLF, blanks
    at a line's end, no final newline, gzip, FASTQ with quality lines that start with '>'"""
    import gzip, random
    exe = tmp_path / "reader_equiv"
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", f"-I{ROOT}/kmtricks_amd/host", f"{ROOT}/tests/helpers/reader_equiv.cpp", "-lz", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rnd = random.Random(5)
    out = ""
    for i in range(20000):
        L = rnd.choice([0, 1, 5, 150, 150, 150, 150, 151, 400])
        s = "".join(rnd.choice("ACGTN") for _ in range(L))
        kind = rnd.random()
        if kind < 0.8: out += f">r{i} some text\n{s}\n"
        elif kind < 0.9: out += f">r{i}\n{s[:L // 2]}\n{s[L // 2:]}\n"
        elif kind < 0.95: out += f">r{i}\n{s} \n\n"
        else: out += f">r{i}\r\n{s}\r\n"
    files = {"a.fa": out, "b.fa": out.rstrip("\n"), "d.fq": "".join(f"@q{i}\n{'ACGT' * 10}\n+\n{'>' * 40}\n" for i in range(3000)),
             "e.fa": ">only\n" + "ACGT" * 700000 + "\n"}      # (a record longer than the reader's 1 MB block)
    for name, text in files.items():
        (tmp_path / name).write_text(text)
    with gzip.open(tmp_path / "c.fa.gz", "wt") as f:
        f.write(out)
    for name in list(files) + ["c.fa.gz"]:
        r = subprocess.run([str(exe), str(tmp_path / name)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.split()[-1] == "same", (name, r.stdout, r.stderr)
