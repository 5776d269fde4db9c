"""An independent pin for the merge arithmetic and for the PartiInfo<5> counters (CPU; the GPU twins are in test_merge_gpu.py /
test_count_gpu.py and import the vectors from here).

The reference's own tests pin the merge by row counts only (tests/merge_test.cpp:5-78).  What a cell of a row holds, which records are
rescued and what the six statistics count rests on merge.hpp:183-260 alone, so it is pinned twice here, by means that share nothing
with oracle/kmx_oracle.c's cursor-walking restatement:
  * hand-worked vectors -- every expected cell and counter below was written down by reading merge.hpp, the line that decides it cited;
  * `dict_merge`: a second restatement of another shape (a dictionary key -> entries over ALL samples, no cursors, no streams),
    checked against the hand-worked vectors first and then used as the judge of the oracle on random cohorts.
Same idea for `fill_partitions.hpp:59-105`: `pinfo_from_strings` cuts super-k-mers and kx-mers out of the reads as STRINGS."""
import itertools
import numpy as np
import pytest

import orc
from synth import kmer_value


# ---------------------------------------------------------------------------------------------------------------- second restatement
def dict_merge(lists, soft, rec_min, share_min):
    """lists[i] = {key: count}.  -> (rows [(key, [counts])] ascending, stats[6][N]) -- merge.hpp:183-260 per KEY instead of per stream"""
    N = len(lists)
    table = {}
    for i, l in enumerate(lists):
        for key, c in l.items():
            table.setdefault(key, []).append((i, c))
    stats = [[0] * N for _ in range(6)]          # NON_SOLID, RESCUED, UNIQUE_WO, UNIQUE_W, TOTAL_WO, TOTAL_W (merge.hpp:72-83)
    rows = []
    for key in sorted(table):
        solid = [(i, c) for i, c in table[key] if c >= soft[i]]                     # merge.hpp:199
        weak = [(i, c) for i, c in table[key] if c < soft[i]]
        saved = weak if (share_min and len(solid) >= share_min) else []             # merge.hpp:214-217, 234-247
        out = [0] * N
        for i, c in solid + saved:
            out[i] = c
        for i, c in solid:
            stats[2][i] += 1; stats[3][i] += 1; stats[4][i] += c; stats[5][i] += c  # inc_uwo, inc_two
        for i, c in weak:
            stats[0][i] += 1                                                        # inc_ns
        for i, c in saved:
            stats[1][i] += 1; stats[3][i] += 1; stats[5][i] += c                    # inc_rd, inc_uw, inc_tw
        if len(solid) >= rec_min:                                                   # merge.hpp:249-250
            rows.append((key, out))
    return rows, stats


# ---------------------------------------------------------------------------------------------------------------- hand-worked vectors
# three samples, soft-min a = [2, 2, 3]
HAND_LISTS = [{10: 5, 11: 1, 12: 1, 13: 1, 14: 1, 16: 1, 17: 2},
              {12: 4, 13: 4, 14: 4, 17: 7},
              {13: 2, 14: 3, 16: 2, 18: 3}]
HAND_SOFT = [2, 2, 3]
# key 10: one solid record (5 >= 2, merge.hpp:199-209)               key 11: one non-solid record (1 < 2, :210-217)
# key 12: non-solid in S0, solid in S1                                 key 13: non-solid in S0 and S2 (2 < 3), solid in S1
# key 14: non-solid in S0, solid in S1 and S2 (3 >= 3: `>=`, :199)     key 16: non-solid in S0 and S2, solid nowhere
# key 17: solid in S0 (2 >= 2: the boundary) and S1                    key 18: solid in S2 (3 >= 3)
_NS = [5, 0, 2]              # NON_SOLID: S0 keys 11 12 13 14 16; S2 keys 13 16 -- counted whatever share-min is (:212-213)
_UWO = [2, 4, 2]             # UNIQUE_WO_RESCUE: S0 keys 10 17; S1 keys 12 13 14 17; S2 keys 14 18 (:205-207)
_TWO = [7, 19, 6]            # TOTAL_WO_RESCUE: 5 + 2; 4 + 4 + 4 + 7; 3 + 3
HAND_CASES = [
    # (recurrence-min, share-min, rows, statistics)
    # share-min 0: a non-solid record is zeroed at once (:216-217); recurrence counts solid records only (:201)
    (1, 0, [(10, [5, 0, 0]), (12, [0, 4, 0]), (13, [0, 4, 0]), (14, [0, 4, 3]), (17, [2, 7, 0]), (18, [0, 0, 3])],
     [_NS, [0, 0, 0], _UWO, _UWO, _TWO, _TWO]),
    # share-min 1: one solid record anywhere rescues every non-solid record of the key (:234-247): key 12 S0, key 13 S0 + S2, key 14 S0;
    # keys 11 and 16 have no solid record: solid_in 0 < 1, zeroed (:236-237), recurrence 0: no row
    (1, 1, [(10, [5, 0, 0]), (12, [1, 4, 0]), (13, [1, 4, 2]), (14, [1, 4, 3]), (17, [2, 7, 0]), (18, [0, 0, 3])],
     [_NS, [3, 0, 1], _UWO, [5, 4, 3], _TWO, [10, 19, 8]]),
    # share-min 2, recurrence-min 2: key 13 has ONE solid record: rescue denied at the boundary (1 >= 2 is false, :236); key 14 has two:
    # granted -- and only keys 14 and 17 reach recurrence 2 (the rescued record does not count: recurrence is set at :201 only)
    (2, 2, [(14, [1, 4, 3]), (17, [2, 7, 0])],
     [_NS, [1, 0, 0], _UWO, [3, 4, 2], _TWO, [8, 19, 6]]),
    # share-min 1, recurrence-min 2: keys 12 and 13 are rescued but have recurrence 1 -> no row, yet their rescue IS in the statistics
    # (they are accumulated for every key, kept or not: :239-245 run before :249)
    (2, 1, [(14, [1, 4, 3]), (17, [2, 7, 0])],
     [_NS, [3, 0, 1], _UWO, [5, 4, 3], _TWO, [10, 19, 8]]),
    # recurrence-min 0: every key is a row (0 >= 0, :249), the keys without a solid record as rows of zeros
    (0, 0, [(10, [5, 0, 0]), (11, [0, 0, 0]), (12, [0, 4, 0]), (13, [0, 4, 0]), (14, [0, 4, 3]), (16, [0, 0, 0]), (17, [2, 7, 0]), (18, [0, 0, 3])],
     [_NS, [0, 0, 0], _UWO, _UWO, _TWO, _TWO]),
    (0, 1, [(10, [5, 0, 0]), (11, [0, 0, 0]), (12, [1, 4, 0]), (13, [1, 4, 2]), (14, [1, 4, 3]), (16, [0, 0, 0]), (17, [2, 7, 0]), (18, [0, 0, 3])],
     [_NS, [3, 0, 1], _UWO, [5, 4, 3], _TWO, [10, 19, 8]]),
    # share-min 3 (more than any key's solid records): nothing is ever rescued, the rows are share-min 0's
    (1, 3, [(10, [5, 0, 0]), (12, [0, 4, 0]), (13, [0, 4, 0]), (14, [0, 4, 3]), (17, [2, 7, 0]), (18, [0, 0, 3])],
     [_NS, [0, 0, 0], _UWO, _UWO, _TWO, _TWO]),
    # recurrence-min 3: no key is solid in all three samples
    (3, 0, [], [_NS, [0, 0, 0], _UWO, _UWO, _TWO, _TWO]),
]


def hand_arrays(kw=1, key_shift=0):
    """HAND_LISTS as the (keys, counts) arrays the C ABIs take; key_shift moves the keys into the upper word(s) (wide keys compare most
    significant word first, kmer.hpp:262-268)"""
    out = []
    for l in HAND_LISTS:
        ks = sorted(l)
        keys = np.zeros((len(ks), kw), np.uint64)
        for j, key in enumerate(ks):
            v = key << key_shift
            for w in range(kw):
                keys[j, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
        out.append((keys, np.array([l[key] for key in ks], np.uint32)))
    return out


def body_of(rows, kw, n, mode, key_shift=0):
    """the rows as the matrix body: key words (low first) + n x u32, or + ceil(n / 8) presence/absence bytes, bit i%8 of byte i/8
    (utils.hpp:104-116, io/matrix_file.hpp:120-127, io/pa_matrix_file.hpp:98-105)"""
    out = bytearray()
    for key, cnt in rows:
        out += (key << key_shift).to_bytes(8 * kw, "little")
        if mode == orc.MODE_COUNT:
            out += np.array(cnt, np.uint32).tobytes()
        else:
            bits = bytearray((n + 7) // 8)
            for i, c in enumerate(cnt):
                if c:
                    bits[i >> 3] |= 1 << (i & 7)
            out += bits
    return bytes(out)


@pytest.mark.parametrize("case", range(len(HAND_CASES)))
def test_dict_merge_reproduces_the_hand_worked_vectors(case):
    r, s, rows, stats = HAND_CASES[case]
    got_rows, got_stats = dict_merge(HAND_LISTS, HAND_SOFT, r, s)
    assert got_rows == rows and got_stats == stats


@pytest.mark.parametrize("kw,shift", [(1, 0), (1, 40), (2, 0), (2, 70), (3, 130), (4, 200)])
@pytest.mark.parametrize("mode", [orc.MODE_COUNT, orc.MODE_PA])
@pytest.mark.parametrize("case", range(len(HAND_CASES)))
def test_oracle_reproduces_the_hand_worked_vectors(case, mode, kw, shift):
    r, s, rows, stats = HAND_CASES[case]
    body, n, st = orc.merge_matrix(hand_arrays(kw, shift), kw, HAND_SOFT, r, s, mode)
    assert n == len(rows) and body == body_of(rows, kw, 3, mode, shift)
    assert st.tolist() == stats


# ---------------------------------------------------------------------------------------------------------------- the Bloom writers
# Round 6: write_as_bf / write_as_bfc / write_as_bft (merge.hpp:575-644) pinned the same way -- the reference's own vectors give these
# modes row counts and packc_test.cpp only.  One window [100, 114] (15 hash rows: not a multiple of 8), three samples, soft-min
# [2, 2, 3]; every byte below was written down from the lines cited.
#   hash 100: S0 5 (solid)                         -> counts [5, 0, 0]
#   hash 101: S0 1 (not solid, nobody solid)       -> recurrence 0: no row of its own -- the NEXT kept hash's `while (m_current > current)`
#             fills it with the empty vector (:581-585); with recurrence-min 0 it IS kept (0 >= 0, merge.hpp:503-504) and written as
#             the zero vector, `current = m_current + 1` (:590): the same bytes
#   hash 102: nobody                               -> empty (gap filling)
#   hash 103: S0 1 (not solid), S1 4 (solid)       -> [0, 4, 0]; share-min 1: S0 rescued (solid_in 1 >= 1, :489-499) -> [1, 4, 0]
#   hash 104: S0 1, S1 4, S2 3 (3 >= 3: solid)     -> [0, 4, 3]; rescued: [1, 4, 3]; recurrence 2
#   hash 107: S0 2 (2 >= 2: solid), S1 7           -> [2, 7, 0]; recurrence 2
#   hash 109: S2 1 (not solid, nobody solid)       -> no row
#   hash 112: S0 9 (solid), S2 1 (not solid)       -> [9, 0, 0]; rescued: [9, 0, 1]
#   hashes 113, 114: the closing `while (current <= upper)` (:593-597)
BLOOM_LISTS = [{100: 5, 101: 1, 103: 1, 104: 1, 107: 2, 112: 9}, {103: 4, 104: 4, 107: 7}, {104: 3, 109: 1, 112: 1}]
BLOOM_SOFT = [2, 2, 3]
BLOOM_LOWER, BLOOM_UPPER = 100, 114
_B_NS = [3, 0, 2]            # NON_SOLID: S0 hashes 101 103 104; S2 hashes 109 112 (inc_ns, merge.hpp:475-476)
_B_UWO = [3, 3, 1]           # S0 100 107 112; S1 103 104 107; S2 104 (inc_uwo, :468-472)
_B_TWO = [16, 15, 3]         # 5 + 2 + 9; 4 + 4 + 7; 3
_B_STATS0 = [_B_NS, [0, 0, 0], _B_UWO, _B_UWO, _B_TWO, _B_TWO]
_B_STATS1 = [_B_NS, [2, 0, 1], _B_UWO, [5, 3, 2], _B_TWO, [18, 15, 4]]      # rescued: S0 at 103 and 104 (1 + 1), S2 at 112 (1): inc_rd, inc_uw, inc_tw (:494-498)


def _bf_rows(d):
    """{hash: byte(s)} -> the window's rows, the others empty"""
    w = len(next(iter(d.values())))
    return b"".join(bytes(d.get(h, [0] * w)) for h in range(BLOOM_LOWER, BLOOM_UPPER + 1))


# write_as_bf: a row is NBYTES(3) = 1 byte, sample i is bit i % 8 of byte i / 8 (set_bit_vector, utils.hpp:104-116: BITSET = 1 << (b % 8))
BLOOM_BF = {
    (1, 0): _bf_rows({100: [0x01], 103: [0x02], 104: [0x06], 107: [0x03], 112: [0x01]}),
    (1, 1): _bf_rows({100: [0x01], 103: [0x03], 104: [0x07], 107: [0x03], 112: [0x05]}),
    (2, 0): _bf_rows({104: [0x06], 107: [0x03]}),                     # recurrence-min 2: only 104 (S1, S2) and 107 (S0, S1) are kept
    (2, 1): _bf_rows({104: [0x07], 107: [0x03]}),                     # ... the rescued S0 at 104 rides along; 103 and 112 are rescued but not kept
    (0, 0): _bf_rows({100: [0x01], 103: [0x02], 104: [0x06], 107: [0x03], 112: [0x01]}),      # recurrence-min 0: 101 and 109 are kept as zero vectors
}
# write_as_bfc: pack_v (packc.hpp:26-43) -- field i holds to_n_b(count, w) = min(bit length, 2^w - 1) in w bits from bit offset i * w,
# counted from the MOST significant bit of the first byte (bitpacker::insert).  to_n_b: 1 -> 1, 2 -> 2, 3 -> 2, 4 -> 3, 5 -> 3, 7 -> 3,
# 9 -> 4 (w = 2 caps at 3).  share-min 1 rows: [5,0,0] [1,4,0] [1,4,3] [2,7,0] [9,0,1].
#   w = 2, a row is byte_count_pack(3, 2) = 1 byte: 11 00 00 00 | 01 11 00 00 | 01 11 10 00 | 10 11 00 00 | 11 00 01 00
#   w = 3, a row is 2 bytes: 011 000 000 -> 0110 0000 0 | 001 011 000 -> 0010 1100 0 | 001 011 010 -> 0010 1101 0 | 010 011 000 -> 0100 1100 0
#          | 100 000 001 -> 1000 0000 1
BLOOM_BFC = {
    (1, 1, 2): _bf_rows({100: [0xC0], 103: [0x70], 104: [0x78], 107: [0xB0], 112: [0xC4]}),
    (1, 0, 2): _bf_rows({100: [0xC0], 103: [0x30], 104: [0x38], 107: [0xB0], 112: [0xC0]}),      # not rescued: [0,4,0] [0,4,3] [9,0,0]
    (1, 1, 3): _bf_rows({100: [0x60, 0x00], 103: [0x2C, 0x00], 104: [0x2D, 0x00], 107: [0x4C, 0x00], 112: [0x80, 0x80]}),
    (2, 1, 3): _bf_rows({104: [0x2D, 0x00], 107: [0x4C, 0x00]}),
}
# write_as_bft: the bf rows in a BitMatrix(ROUND_UP(15, 8) = 16 rows, ROUND_UP(3, 8) / 8 = 1 byte), transposed (bitmatrix.hpp:209-289:
# out row y, bit x % 8 of byte x / 8 = in row x, bit y) and dumped whole: 8 rows (samples 3..7 are padding) of 2 bytes, bit r of a
# sample's row = hash 100 + r.  share-min 1: S0 holds hashes 100 103 104 107 112 -> bits 0 3 4 7 | 12 -> 99 10; S1 103 104 107 -> 98 00;
# S2 104 112 -> 10 10.  share-min 0: S0 100 107 112 -> 81 10; S1 103 104 107 -> 98 00; S2 104 -> 10 00.
BLOOM_BFT = {
    (1, 1): bytes([0x99, 0x10, 0x98, 0x00, 0x10, 0x10] + [0] * 10),
    (1, 0): bytes([0x81, 0x10, 0x98, 0x00, 0x10, 0x00] + [0] * 10),
    (2, 1): bytes([0x90, 0x00, 0x90, 0x00, 0x10, 0x00] + [0] * 10),      # kept: 104 (all three, S0 rescued) and 107 (S0, S1)
}
BLOOM_STATS = {0: _B_STATS0, 1: _B_STATS1}


def bloom_arrays():
    return [(np.array(sorted(l), np.uint64), np.array([l[h] for h in sorted(l)], np.uint32)) for l in BLOOM_LISTS]


def bloom_cases():
    """(mode, recurrence-min, share-min, bitw, body, statistics)"""
    out = []
    for (r, s), body in BLOOM_BF.items():
        out.append((orc.MODE_BF, r, s, 2, body, BLOOM_STATS[s]))
    for (r, s, w), body in BLOOM_BFC.items():
        out.append((orc.MODE_BFC, r, s, w, body, BLOOM_STATS[s]))
    for (r, s), body in BLOOM_BFT.items():
        out.append((orc.MODE_BFT, r, s, 2, body, BLOOM_STATS[s]))
    return out


@pytest.mark.parametrize("case", range(12))
def test_oracle_reproduces_the_hand_worked_bloom_vectors(case):
    mode, r, s, w, body, stats = bloom_cases()[case]
    got, rows, st = orc.merge_matrix(bloom_arrays(), 1, BLOOM_SOFT, r, s, mode, BLOOM_LOWER, BLOOM_UPPER, w)
    assert got == body
    assert rows == (8 if mode == orc.MODE_BFT else 15)
    assert st.tolist() == stats


def test_the_bloom_vectors_agree_with_each_other():
    """what the three writers share: a bfc field is non-zero exactly where the bf bit is set, a bft row is the bf column"""
    assert len(bloom_cases()) == 12
    for (r, s), bf in BLOOM_BF.items():
        if (r, s, 2) in BLOOM_BFC:
            for row, (a, b) in enumerate(zip(bf, BLOOM_BFC[(r, s, 2)])):
                for i in range(3):
                    assert ((a >> i) & 1) == (1 if (b >> (6 - 2 * i)) & 3 else 0), (r, s, row, i)
        if (r, s) in BLOOM_BFT:
            t = BLOOM_BFT[(r, s)]
            for row in range(15):
                for i in range(3):
                    assert ((bf[row] >> i) & 1) == ((t[2 * i + (row >> 3)] >> (row & 7)) & 1), (r, s, row, i)


def random_cohort(rng, n, nkeys, kw):
    """a cohort with every kind of key: in most samples, in a few, in one; counts around the soft-mins"""
    top = 1 << (64 * kw - 2)
    pool = sorted({int(rng.integers(0, 1 << 62)) * (top >> 62) + int(rng.integers(0, 1 << 30)) for _ in range(nkeys)})
    lists = [dict() for _ in range(n)]
    for key in pool:
        kind = rng.random()
        who = range(n) if kind < 0.3 else rng.choice(n, size=min(n, int(rng.integers(1, 4))), replace=False)
        for i in who:
            if kind >= 0.3 or rng.random() < 0.9:
                lists[i][key] = int(rng.integers(1, 7))
    if n > 2:
        lists[int(rng.integers(0, n))] = {}       # an empty list
    return lists


@pytest.mark.parametrize("seed", range(24))
def test_oracle_against_the_second_restatement_on_random_cohorts(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 9, 17, 40]))
    kw = int(rng.choice([1, 1, 2, 3, 4]))
    lists = random_cohort(rng, n, int(rng.integers(20, 400)), kw)
    soft = [int(x) for x in rng.integers(1, 5, n)]
    arrays = []
    for l in lists:
        ks = sorted(l)
        keys = np.array([[(key >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(kw)] for key in ks], np.uint64).reshape(len(ks), kw)
        arrays.append((keys, np.array([l[key] for key in ks], np.uint32)))
    for r, s in itertools.product([0, 1, 2, n], [0, 1, 2, n + 1]):
        rows, stats = dict_merge(lists, soft, r, s)
        for mode in (orc.MODE_COUNT, orc.MODE_PA):
            body, nrows, st = orc.merge_matrix(arrays, kw, soft, r, s, mode)
            assert nrows == len(rows) and body == body_of(rows, kw, n, mode), (n, kw, r, s, mode)
            assert st.tolist() == stats, (n, kw, r, s)


# ---------------------------------------------------------------------------------------------------------------- PartiInfo<5>
_COMP = str.maketrans("ACGT", "TGCA")


def pinfo_from_strings(reads, k, m, repart, nb_parts):
    """fill_partitions.hpp:59-105 + Sequence2SuperKmer.hpp:80-133 + Model.hpp:1010-1070, 1220-1251 over STRINGS: the k-mers of a read,
    their minimizers (smallest allowed canonical m-mer value), super-k-mers = maximal runs of valid k-mers with one minimizer, at most
    maxs long; a super-k-mer's strand word cut into runs of one strand, those into pieces of at most 5: a kx-mer per piece, counted under
    (partition, pieces' k-mers - 1, top four nucleotides of the piece's first canonical k-mer if forward, of its last one if reverse).
    -> (pinfo[nb_parts][2 + 5 * 256], {minimizer: [super-k-mers, k-mers]})"""
    bits = 64 * ((k + 31) // 32) if k % 32 else 64 * (k // 32 + 1)      # the instantiated type: the first KMER_LIST entry ABOVE k (loop_executor.hpp:47-52)
    maxs = min((bits - 8) // 2, 255)                                    # Sequence2SuperKmer.hpp:146
    dflt = 4 ** m - 1

    def mmer_value(s):                                                  # Model.hpp:1040-1064
        v = min(kmer_value(s), kmer_value(s[::-1].translate(_COMP)))
        name = "".join("ACTG"[(v >> (2 * d)) & 3] for d in range(m - 1, -1, -1))
        return dflt if "AA" in name[1:] else v                          # "AA" anywhere but at the two leading positions (:1220-1251)

    pinfo = [[0] * (2 + 5 * 256) for _ in range(nb_parts)]
    minim = {}

    def flush(sk):      # sk: [(canonical value, forward?)] of one super-k-mer, its minimizer
        if not sk[0]:
            return
        kms, mini = sk
        p = int(repart[mini])
        e = minim.setdefault(mini, [0, 0]); e[0] += 1; e[1] += len(kms)
        for fwd, grp in itertools.groupby(kms, key=lambda t: t[1]):
            grp = list(grp)
            for a in range(0, len(grp), 5):
                piece = grp[a:a + 5]
                val = piece[0][0] if fwd else piece[-1][0]
                radix = (val >> (2 * (k - 4))) & 255
                pinfo[p][0] += len(piece); pinfo[p][1] += 1; pinfo[p][2 + (len(piece) - 1) * 256 + radix] += 1

    for read in reads:
        read = read.upper()
        cur, cur_min = [], None
        for j in range(len(read) - k + 1):
            s = read[j:j + k]
            if any(c not in "ACGT" for c in s):
                flush((cur, cur_min)); cur, cur_min = [], None
                continue
            f, r = kmer_value(s), kmer_value(s[::-1].translate(_COMP))
            mini = min(mmer_value(s[i:i + m]) for i in range(k - m + 1))
            if cur and (mini != cur_min or len(cur) >= maxs):
                flush((cur, cur_min)); cur = []
            cur.append((min(f, r), f < r)); cur_min = mini
        flush((cur, cur_min))
    return pinfo, minim


@pytest.mark.parametrize("k,m,P", [(31, 10, 4), (21, 8, 3), (32, 10, 5), (47, 9, 4), (63, 10, 4), (64, 10, 3), (96, 11, 4), (127, 10, 2)])
def test_parti_info_oracle_against_the_string_restatement(k, m, P):
    rng = np.random.default_rng(k * 131 + m)
    g = "".join("ACGT"[i] for i in rng.integers(0, 4, 1500))
    reads = []
    for _ in range(40):
        a = int(rng.integers(0, len(g) - 300)); s = g[a:a + int(rng.integers(k - 3, 300))]
        if rng.random() < 0.5:
            s = s[::-1].translate(_COMP)
        if rng.random() < 0.3:
            i = int(rng.integers(0, max(1, len(s)))); s = s[:i] + "N" + s[i + 1:]
        if rng.random() < 0.2:
            i = int(rng.integers(0, max(1, len(s) - 20))); s = s[:i] + "A" * 18 + s[i + 18:]      # poly-A: the default minimizer
        reads.append(s)
    reads += reads[:7]
    lut, rep = orc.minimizer_lut(m), orc.repart_static(m, P)
    exp, minim = pinfo_from_strings(reads, k, m, rep, P)
    got, ms, mk, _ = orc.superk_stats(reads, k, m, lut, rep, P)
    assert got.tolist() == exp
    assert {int(v): [int(ms[v]), int(mk[v])] for v in np.nonzero(ms)[0]} == minim
