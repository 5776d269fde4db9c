"""ctypes binding of oracle/libkmx_oracle.so -- the CPU checker (tests only)."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "libkmx_oracle.so"))

_lib2 = C.CDLL(os.path.join(ROOT, "oracle", "libkmx_oracle_repart.so"))
_lib2.orc_repart_sampled.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
u8p, u16p, u32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64))


class OrcBuf(C.Structure):
    _fields_ = [("data", u8p), ("len", C.c_size_t), ("cap", C.c_size_t),
                ("nb_kmers", C.c_uint64), ("nb_superk", C.c_uint64)]


class OrcList(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("counts", C.c_void_p), ("n", C.c_uint64)]


_lib.orc_xxh64.restype = C.c_uint64
_lib.orc_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
_lib.orc_minimizer_lut.argtypes = [C.c_int, C.c_void_p]
_lib.orc_minimizer_of.restype = C.c_uint32
_lib.orc_minimizer_of.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
_lib.orc_repart_static.argtypes = [C.c_int, C.c_uint32, C.c_void_p]
_lib.orc_superk_partition.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.POINTER(OrcBuf), C.c_void_p]
_lib.orc_superk_partition_stats.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.POINTER(OrcBuf), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.orc_buf_free.argtypes = [C.POINTER(OrcBuf)]
_lib.orc_superk_decode.restype = C.c_uint64
_lib.orc_superk_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
_lib.orc_count_kmer.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
_lib.orc_count_hash.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32,
                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
_lib.orc_merge_matrix.argtypes = [C.POINTER(OrcList), C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                  C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
_lib.orc_to_n_b.restype = C.c_uint32
_lib.orc_to_n_b.argtypes = [C.c_uint32, C.c_uint32]
_lib.orc_byte_count_pack.restype = C.c_uint64
_lib.orc_byte_count_pack.argtypes = [C.c_uint64, C.c_uint64]
_lib.orc_transpose_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
_lib.orc_free.argtypes = [C.c_void_p]
_lib.orc_kmer_to_string.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
_lib.orc_kmer_from_string.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
_lib.orc_revcomp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]

MODE_COUNT, MODE_PA, MODE_BF, MODE_BFC, MODE_BFT = range(5)


def xxh64(data: bytes, seed=0):
    return _lib.orc_xxh64(data, len(data), seed)


def minimizer_lut(m):
    lut = np.empty(4 ** m, dtype=np.uint32)
    _lib.orc_minimizer_lut(m, lut.ctypes.data)
    return lut


def repart_static(m, nb_parts):
    t = np.empty(4 ** m, dtype=np.uint16)
    _lib.orc_repart_static(m, nb_parts, t.ctypes.data)
    return t


def kmer_from_string(s, kw=None):
    k = len(s)
    kw = kw or (k + 31) // 32
    w = np.zeros(kw, dtype=np.uint64)
    _lib.orc_kmer_from_string(s.encode(), k, w.ctypes.data, kw)
    return w


def kmer_to_string(words, k):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    out = C.create_string_buffer(k + 1)
    _lib.orc_kmer_to_string(words.ctypes.data, k, out)
    return out.value.decode()


def revcomp(words, k):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    out = np.zeros_like(words)
    _lib.orc_revcomp(words.ctypes.data, out.ctypes.data, k, len(words))
    return out


def minimizer_of(words, k, m, lut):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    return _lib.orc_minimizer_of(words.ctypes.data, k, len(words), m, lut.ctypes.data)


def superk_partition(seqs, k, m, lut, repart, nb_parts, with_pinfo=False):
    """-> list of (record bytes, nb_kmers, nb_superk) per partition [, pinfo]"""
    bufs = (OrcBuf * nb_parts)()
    pinfo = np.zeros(nb_parts * (2 + 5 * 256), dtype=np.uint64) if with_pinfo else None
    for s in seqs:
        b = s if isinstance(s, bytes) else s.encode()
        rc = _lib.orc_superk_partition(b, len(b), k, m, lut.ctypes.data, repart.ctypes.data, nb_parts, bufs,
                                       pinfo.ctypes.data if with_pinfo else None)
        assert rc == 0, rc
    out = []
    for p in range(nb_parts):
        out.append((C.string_at(bufs[p].data, bufs[p].len) if bufs[p].len else b"", bufs[p].nb_kmers, bufs[p].nb_superk))
        _lib.orc_buf_free(C.byref(bufs[p]))
    return (out, pinfo.reshape(nb_parts, -1)) if with_pinfo else out


def superk_stats(seqs, k, m, lut, repart, nb_parts):
    """PartiInfo<5> of a read set: (pinfo[nb_parts, 2 + 5*256], minim_superks, minim_kmers, minim_kxmers) -- the
    per-partition counters of fill_partitions.hpp:67-102 and the per-minimizer records (kx-mers counted as the
    sampling pass of the repartition does, RepartitionAlgorithm.cpp:182-215)"""
    bufs = (OrcBuf * nb_parts)()
    pinfo = np.zeros(nb_parts * (2 + 5 * 256), dtype=np.uint64)
    ms, mk, mx = (np.zeros(4 ** m, dtype=np.uint64) for _ in range(3))
    for s in seqs:
        b = s if isinstance(s, bytes) else s.encode()
        rc = _lib.orc_superk_partition_stats(b, len(b), k, m, lut.ctypes.data, repart.ctypes.data, nb_parts, bufs,
                                             pinfo.ctypes.data, ms.ctypes.data, mk.ctypes.data, mx.ctypes.data)
        assert rc == 0, rc
    for p in range(nb_parts):
        _lib.orc_buf_free(C.byref(bufs[p]))
    return pinfo.reshape(nb_parts, -1), ms, mk, mx


def repart_sampled(minim_kxmers, nb_parts):
    """Repartitor::computeDistrib (gatb PartiInfo.cpp:48-103) on kx-mers per minimizer"""
    mx = np.ascontiguousarray(minim_kxmers, dtype=np.uint64)
    out = np.zeros(len(mx), dtype=np.uint16)
    rc = _lib2.orc_repart_sampled(mx.ctypes.data, len(mx), nb_parts, out.ctypes.data)
    assert rc == 0
    return out


def kw_of_k(k):
    """words of a k-mer in files, hashes and comparisons: ceil(k / 32) (kmer.hpp:215), whatever Kmer<MAX_K> holds it"""
    return (k + 31) // 32


def superk_decode(recs: bytes, k):
    kw = kw_of_k(k)
    n = _lib.orc_superk_decode(recs, len(recs), k, kw, None)
    out = np.zeros((n, kw), dtype=np.uint64)
    _lib.orc_superk_decode(recs, len(recs), k, kw, out.ctypes.data)
    return out


def _take(ptr, n, dtype, width=1):
    if n == 0:
        _lib.orc_free(ptr)
        return np.zeros((0, width) if width > 1 else (0,), dtype=dtype)
    nbytes = n * width * np.dtype(dtype).itemsize
    a = np.frombuffer(C.string_at(ptr.value, nbytes), dtype=dtype).copy()
    _lib.orc_free(ptr)
    return a.reshape(n, width) if width > 1 else a


def count_kmer(recs: bytes, k, hard_min):
    kw = kw_of_k(k)
    kp, cp, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
    rc = _lib.orc_count_kmer(recs, len(recs), k, hard_min, C.byref(kp), C.byref(cp), C.byref(n))
    assert rc == 0
    keys = _take(kp, n.value, np.uint64, kw)
    return keys.reshape(n.value, kw), _take(cp, n.value, np.uint32)


def count_hash(recs: bytes, k, win, part, hard_min):
    kp, cp, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
    rc = _lib.orc_count_hash(recs, len(recs), k, win, part, hard_min, C.byref(kp), C.byref(cp), C.byref(n))
    assert rc == 0
    return _take(kp, n.value, np.uint64), _take(cp, n.value, np.uint32)


_lib.orc_khist.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.orc_khist.restype = None


def khist(counts, lower=1, upper=255, acc=None):
    """KHist::inc over `counts` -> dict(unique, total (bins lower..upper), oob [lu, uu, ln, un], sums [unique, total]); `acc` adds to a previous result"""
    h = acc or dict(unique=np.zeros(upper - lower + 1, np.uint64), total=np.zeros(upper - lower + 1, np.uint64),
                    oob=np.zeros(4, np.uint64), sums=np.zeros(2, np.uint64))
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    _lib.orc_khist(c.ctypes.data, len(c), lower, upper, h["unique"].ctypes.data, h["total"].ctypes.data, h["oob"].ctypes.data, h["sums"].ctypes.data)
    return h


def merge_matrix(lists, kw, soft_min, rec_min, share_min, mode, lower=0, upper=0, bitw=2):
    """lists: [(keys uint64[n*kw] or [n,kw], counts uint32[n])].  -> (body bytes, rows, stats[6,N])"""
    N = len(lists)
    arr = (OrcList * max(N, 1))()
    keep = []
    for i, (k_, c_) in enumerate(lists):
        k_ = np.ascontiguousarray(k_, dtype=np.uint64)
        c_ = np.ascontiguousarray(c_, dtype=np.uint32)
        keep.append((k_, c_))
        arr[i].keys = k_.ctypes.data
        arr[i].counts = c_.ctypes.data
        arr[i].n = len(c_)
    sm = np.ascontiguousarray(soft_min, dtype=np.uint32)
    stats = np.zeros((6, N), dtype=np.uint64)
    body, blen, rows = C.c_void_p(), C.c_uint64(), C.c_uint64()
    rc = _lib.orc_merge_matrix(arr, N, kw, sm.ctypes.data, rec_min, share_min, mode, lower, upper, bitw,
                               C.byref(body), C.byref(blen), C.byref(rows), stats.ctypes.data)
    assert rc == 0, rc
    # (string_at takes an int: bodies of 2 GiB and more are copied through a ctypes array)
    data = (C.string_at(body.value, blen.value) if blen.value < (1 << 31) else bytes((C.c_ubyte * blen.value).from_address(body.value))) if blen.value else b""
    _lib.orc_free(body)
    return data, rows.value, stats


def to_n_b(c, w):
    return _lib.orc_to_n_b(c, w)


def byte_count_pack(n, b):
    return _lib.orc_byte_count_pack(n, b)


def transpose_bits(mat: np.ndarray, nrows, ncols):
    mat = np.ascontiguousarray(mat, dtype=np.uint8)
    out = np.zeros(ncols * (nrows // 8), dtype=np.uint8)
    _lib.orc_transpose_bits(mat.ctypes.data, out.ctypes.data, nrows, ncols)
    return out
