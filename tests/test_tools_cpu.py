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


def _mmers(s, m):
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    for i in range(len(s) - m + 1):
        v = 0
        for c in s[i:i + m]:
            v = (v << 2) | code[c]
        yield v
