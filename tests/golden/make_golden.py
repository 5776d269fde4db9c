#!/usr/bin/env python3
"""Extract the reference's golden vectors into tests/golden/ (run in the build
container only, where /root/reference exists; the outputs are committed).

What is copied is DATA: the fixture files the reference's own tests read
(tests/data/**) and the expected values its tests assert (parsed out of the
EXPECT_/ASSERT_ lines of tests/task_main.cpp, tests/merge_test.cpp,
tests/repartition_test.cpp, tests/packc_test.cpp, tests/histogram_test.cpp, tests/kmer_test.cpp,
tests/processor_test.cpp).  No reference source text
is stored.
"""
import json, os, re, shutil, struct, sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def copy_data():
    for rel in ["1.fasta", "2.fasta", "hash.info", "kmtricks.fof"]:
        shutil.copyfile(f"{REF}/tests/data/{rel}", f"{OUT}/{rel}")
    for kind, ext in (("kmers", "kmer"), ("hashes", "hash")):
        for p in range(4):
            d = f"{OUT}/partitions/{kind}/partition_{p}"
            os.makedirs(d, exist_ok=True)
            for s in ("D1", "D2"):
                shutil.copyfile(f"{REF}/tests/data/partitions/{kind}/partition_{p}/{s}.{ext}", f"{d}/{s}.{ext}")


def repart_sparse():
    """tests/data/repart_gatb/repartition.minimRepart (2 MB, almost all zero)
    -> sparse JSON {header fields, non-zero entries}."""
    raw = open(f"{REF}/tests/data/repart_gatb/repartition.minimRepart", "rb").read()
    nb_part, nb_minims, nb_pass = struct.unpack_from("<HQH", raw, 0)
    table = struct.unpack_from(f"<{nb_minims}H", raw, 12)
    tail = raw[12 + 2 * nb_minims:]
    has_freq = tail[0]
    magic = struct.unpack_from("<I", tail, 1)[0]
    nz = {str(i): v for i, v in enumerate(table) if v}
    return {"nb_part": nb_part, "nb_minims": nb_minims, "nb_pass": nb_pass,
            "has_freq": has_freq, "magic": magic, "file_size": len(raw), "nonzero": nz}


def task_main_goldens():
    src = open(f"{REF}/tests/task_main.cpp").read()
    out = {}
    # SuperKmerBinInfoFile expected lines (tests/task_main.cpp:85-114)
    sk = re.findall(r'ASSERT_EQ\(line, "(\d+)"\)', src)
    out["superk_info_D1"] = [int(x) for x in sk[0:9]]
    out["superk_info_D2"] = [int(x) for x in sk[9:18]]
    # blocks of expected k-mers / hashes, in file order
    blocks = re.split(r'km::KmerReader<8192> kr\("([^"]+)"\)|km::HashReader<4294967296> kr\("([^"]+)"\)', src)
    # re.split with 2 groups yields [pre, g1, g2, body, g1, g2, body, ...]
    res = {}
    for i in range(1, len(blocks), 3):
        path = blocks[i] or blocks[i + 1]
        body = blocks[i + 2]
        body = body.split("km::Kmer<MK> kmer;")[0] if False else body
        # stop at next reader (already split), keep only this block's asserts
        kmers = re.findall(r'EXPECT_EQ\(kmer\.to_string\(\), "([ACGT]+)"\); EXPECT_EQ\((\d+), count\)', body)
        hashes = re.findall(r'EXPECT_EQ\(kmer, (\d+)\); EXPECT_EQ\(count, (\d+)\)', body)
        nrec = re.search(r'EXPECT_EQ\(n, (\d+)\)', body)
        key = path.split("km_dir_test/")[1]
        if kmers:
            res[key] = {"kmers": [[k, int(c)] for k, c in kmers]}
        elif hashes:
            res[key] = {"hashes": [[int(h), int(c)] for h, c in hashes]}
        elif nrec:
            res[key] = {"n": int(nrec.group(1))}
    out["count_files"] = res
    return out


def merge_goldens():
    src = open(f"{REF}/tests/merge_test.cpp").read()
    vals = [int(x) for x in re.findall(r'EXPECT_EQ\(count, (\d+)\)', src)]
    return {"hash_rows": vals[0:4], "kmer_rows": vals[4:8], "soft_min": [1, 1], "rec_min": 1, "share_min": 1}


def repartition_goldens():
    src = open(f"{REF}/tests/repartition_test.cpp").read()
    ks = re.findall(r'std::string k(\d) = "([ACGT]+)"', src)
    return {"minimizer_size": 10, "cases": [[k, int(i)] for i, k in ks]}


def packc_goldens():
    src = open(f"{REF}/tests/packc_test.cpp").read()
    bcp = re.findall(r'byte_count_pack\((\d+), (\d+)\), (\d+)\)', src)
    tnb = re.findall(r'to_n_b\((\d+), (\d+)\), (\d+)\)', src)
    return {"byte_count_pack": [[int(a), int(b), int(c)] for a, b, c in bcp],
            "to_n_b": [[int(a), int(b), int(c)] for a, b, c in tnb]}


def histogram_goldens():
    """tests/histogram_test.cpp: the counts fed to KHist(0, 20, 1, 10) and the unique / total bins it asserts"""
    src = open(f"{REF}/tests/histogram_test.cpp").read().split("TEST(histogram, clones)")[0]
    vec = lambda name: [int(x) for x in re.search(r'std::vector<uint64_t> %s \{([^}]*)\}' % name, src).group(1).split(",")]
    lo, hi = re.search(r'KHist hist\(0, 20, (\d+), (\d+)\)', src).groups()
    return {"counts": vec("v"), "unique": vec("r"), "total": vec("rn"), "lower": int(lo), "upper": int(hi), "kmer_size": 20}


def kmer_goldens():
    """tests/kmer_test.cpp:9-152: the asserted names, the canonical / comparison / m-mer / minimizer strings (the random-sequence
    cases assert identities -- to_string(Kmer(s)) == s, rev_comp == str_rev_comp -- which the tests re-run on their own strings)"""
    src = open(f"{REF}/tests/kmer_test.cpp").read()
    names = re.findall(r'km::Kmer<(\d+)> \w+\(\w\);\s*EXPECT_EQ\(\w+\.name\(\), "([^"]+)"\)', src)
    can = src.split("TEST(kmer, canonical)")[1].split("TEST(")[0]
    strs = dict(re.findall(r'std::string (\w) = "([ACGT]+)"', can))
    canon = [[strs["a"], strs["a"]], [strs["b"], strs["c"]]]      # canonical(a) == a; canonical(b) == c != b
    op = src.split("TEST(kmer, operator)")[1].split("TEST(")[0]
    ostr = dict(re.findall(r'std::string (\w) = "([ACGT]+)"', op))
    less = [[int(mk), ostr[x], ostr[y]] for mk, x, y in re.findall(r'km::Kmer<(\d+)> kmer(\w)\(\w\); km::Kmer<\d+> kmer(\w)\(\w\);', op)]
    mi = src.split("TEST(kmer, minimizer)")[1]
    seq = re.search(r'std::string a = "([ACGT]+)"', mi).group(1)
    msize = int(re.search(r'kmer\.mmers\((\d+)\)', mi).group(1))
    blk = mi.split("Mmer m = kmer.minimizer")[0]
    mmers = [m for _, m in sorted(set((int(i), m) for m, i in re.findall(r'EXPECT_EQ\("([ACGT]+)", v\[(\d+)\]\.to_string\(\)\)', blk)))]
    mini = re.search(r'EXPECT_EQ\(m\.to_string\(\), "([ACGT]+)"\)', mi).group(1)
    return {"names": [[int(a), b] for a, b in names], "canonical": canon, "less": less,
            "minimizer": {"kmer": seq, "m": msize, "mmers": mmers, "minimizer": mini}}


def processor_goldens():
    """tests/processor_test.cpp:10-75: records fed to the count processors with abundance-min 3 and what the files hold afterwards"""
    src = open(f"{REF}/tests/processor_test.cpp").read()
    hc = src.split("TEST(processor, hash_count_processor)")[1].split("TEST(")[0]
    ksz, amin = re.search(r'HashCountProcessor<32, 255> p\((\d+), (\d+),', hc).groups()
    fed = [[int(h), int(c)] for h, c in re.findall(r'p\.process\(0, (\d+), (\d+)\)', hc)]
    kept = [[int(re.search(r'EXPECT_EQ\(hash, (\d+)\)', hc).group(1)), int(re.search(r'EXPECT_EQ\(c, (\d+)\)', hc).group(1))]]
    kc = src.split("TEST(processor, kmer_count_processor)")[1]
    k2, a2 = re.search(r'KmerCountProcessor<32, 255> p\((\d+), (\d+),', kc).groups()
    kfed = [int(c) for c in re.findall(r'p\.process\(0, gk\d, (\d+)\)', kc)]
    kkept = [int(re.search(r'EXPECT_EQ\(c, (\d+)\)', kc).group(1))]
    return {"hash": {"kmer_size": int(ksz), "abundance_min": int(amin), "fed": fed, "kept": kept},
            "kmer": {"kmer_size": int(k2), "abundance_min": int(a2), "fed_counts": kfed, "kept_counts": kkept}}


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; goldens are already committed")
    copy_data()
    g = {"repartition_table": repart_sparse(), "task_main": task_main_goldens(),
         "merge_test": merge_goldens(), "repartition_test": repartition_goldens(),
         "packc_test": packc_goldens(), "histogram_test": histogram_goldens(),
         "kmer_test": kmer_goldens(), "processor_test": processor_goldens()}
    json.dump(g, open(f"{OUT}/reference_goldens.json", "w"), indent=1, sort_keys=True)
    print("wrote", f"{OUT}/reference_goldens.json")
