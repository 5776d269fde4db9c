"""ctypes binding of libkmx.so (include/kmx.h).  No fallback of any kind."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMX_LIB") or os.path.join(_HERE, "libkmx.so")      # (KMX_LIB: a tuning build, scripts/dev/build_variant.sh)
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(libkmx has no CPU fallback)")
_lib = C.CDLL(LIB_PATH)

MODE_COUNT, MODE_PA, MODE_BF, MODE_BFC, MODE_BFT = 0, 1, 2, 3, 4
STATS_ROWS = 6


class KmxList(C.Structure):
    _fields_ = [("recs", C.c_void_p), ("n", C.c_uint64)]


class KmxMergeTask(C.Structure):
    _fields_ = [("n_lists", C.c_uint32), ("key_words", C.c_uint32), ("lists", C.POINTER(KmxList)),
                ("soft_min", C.POINTER(C.c_uint32)), ("rec_min", C.c_uint32), ("share_min", C.c_uint32),
                ("mode", C.c_uint32), ("bitw", C.c_uint32), ("lower", C.c_uint64), ("upper", C.c_uint64),
                ("rows_hint", C.c_uint64), ("list_on_device", C.c_void_p)]


_vp = C.c_void_p
_lib.kmx_version.restype = C.c_int
_lib.kmx_create.argtypes = [C.c_int, C.POINTER(_vp)]
_lib.kmx_destroy.argtypes = [_vp]
_lib.kmx_last_error.restype = C.c_char_p
_lib.kmx_last_error.argtypes = [_vp]
_lib.kmx_stream.restype = _vp
_lib.kmx_stream.argtypes = [_vp]
_lib.kmx_set_profiling.argtypes = [_vp, C.c_int]
_lib.kmx_set_file_order.argtypes = [_vp, C.c_int]
KMX_VERSION = 2
if _lib.kmx_version() != KMX_VERSION:
    raise ImportError(f"{LIB_PATH} is version {_lib.kmx_version()}, this binding is for version {KMX_VERSION} (kmx_merge_task layout)")
_lib.kmx_result_kernel.restype = C.c_char_p
_lib.kmx_result_kernel.argtypes = [_vp]
_lib.kmx_result_kernel_ms.restype = C.c_double
_lib.kmx_result_kernel_ms.argtypes = [_vp]
_lib.kmx_result_kernel_parts_ms.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
_lib.kmx_result_transpose_ms.restype = C.c_double
_lib.kmx_result_transpose_ms.argtypes = [_vp]
_lib.kmx_result_body_dev.restype = _vp
_lib.kmx_result_body_dev.argtypes = [_vp, C.c_uint32]
_lib.kmx_merge_dev.argtypes = [_vp, C.POINTER(KmxMergeTask), C.c_uint32, C.POINTER(_vp)]
_lib.kmx_result_wait.argtypes = [_vp]
for _f in ("kmx_result_rows", "kmx_result_sparse_rows", "kmx_result_row_bytes", "kmx_result_body_bytes", "kmx_result_algo_bytes"):
    getattr(_lib, _f).restype = C.c_uint64
    getattr(_lib, _f).argtypes = [_vp, C.c_uint32]
_lib.kmx_result_copy_body.argtypes = [_vp, C.c_uint32, _vp, C.c_uint64]
_lib.kmx_result_copy_stats.argtypes = [_vp, C.c_uint32, _vp]
_lib.kmx_result_copy_body_dev.argtypes = [_vp, C.c_uint32, _vp, C.c_uint64]
_lib.kmx_result_prepare_body.argtypes = [_vp, C.c_uint32]
_lib.kmx_result_arena.argtypes = [_vp, C.c_uint32, C.POINTER(_vp), C.POINTER(C.c_uint64)]
_lib.kmx_result_copy_order.argtypes = [_vp, C.c_uint32, _vp]
_lib.kmx_result_free.argtypes = [_vp]
_lib.kmx_merge.argtypes = [_vp, C.POINTER(KmxMergeTask), C.POINTER(_vp), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_uint64), _vp]
_lib.kmx_free.argtypes = [_vp]
_lib.kmx_count_kmer.argtypes = [_vp, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_vp), C.POINTER(_vp),
                                C.POINTER(C.c_uint64)]
_lib.kmx_count_hash.argtypes = [_vp, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32,
                                C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint64)]
_lib.kmx_count_batch.argtypes = [_vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_int,
                                 C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(_vp), C.POINTER(_vp),
                                 C.POINTER(C.c_uint64)]
_lib.kmx_transpose_bits.argtypes = [_vp, _vp, C.c_uint64, C.c_uint64, _vp]
_lib.kmx_superk_partition.argtypes = [_vp, C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, C.c_uint32,
                                      C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]

class KmxSuperkStats(C.Structure):
    _fields_ = [("part_counters", _vp), ("minim_superks", _vp), ("minim_kmers", _vp), ("minim_kxmers", _vp),
                ("nb_superk", C.c_uint64)]


PINFO_STRIDE = 2 + 5 * 256
_lib.kmx_superk_partition_stats.argtypes = [_vp, C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, C.c_uint32,
                                            C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(KmxSuperkStats)]

_lib.kmx_merge_host.argtypes = [_vp, C.POINTER(KmxMergeTask), C.c_uint32, C.POINTER(_vp)]
_lib.kmx_alloc_pinned.restype = _vp
_lib.kmx_reads_upload.argtypes = [_vp, _vp, C.c_uint64, C.POINTER(_vp)]
_lib.kmx_reads_release.argtypes = [_vp, _vp]
_lib.kmx_reads_release.restype = None
_lib.kmx_alloc_pinned.argtypes = [C.c_size_t]
_lib.kmx_free_pinned.argtypes = [_vp]

_lib.kmx_count_reads.argtypes = [_vp, C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_int, C.c_uint64, C.c_uint32,
                                 C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                 C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(KmxSuperkStats)]

_lib.kmx_superk_sample.argtypes = [_vp, C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(KmxSuperkStats),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]


class KmxSuperkRaw(C.Structure):
    _fields_ = [("part_radix", _vp), ("minim_superks", _vp), ("minim_kmers", _vp), ("nb_superk", C.c_uint64),
                ("minim_sparse", _vp), ("minim_sparse_cap", C.c_uint64), ("minim_sparse_n", C.c_uint64)]


_lib.kmx_device_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
_lib.kmx_store_create.argtypes = [C.c_int, C.c_uint64, C.POINTER(_vp)]
_lib.kmx_store_destroy.argtypes = [_vp]
_lib.kmx_store_used.restype = C.c_uint64
_lib.kmx_store_used.argtypes = [_vp]
_lib.kmx_store_limit.restype = C.c_uint64
_lib.kmx_store_limit.argtypes = [_vp]
_lib.kmx_copy_to_host.argtypes = [_vp, _vp, _vp, C.c_uint64]
_lib.kmx_count_reads_dev_multi.argtypes = [_vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_int, C.c_uint64, C.c_uint32,
                                            _vp, C.c_uint32, _vp, _vp, _vp, _vp]
_lib.kmx_count_reads_dev_multi.restype = C.c_int
_lib.kmx_count_reads_dev.argtypes = [_vp, C.c_char_p, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_int, C.c_uint64, C.c_uint32,
                                     C.POINTER(_vp), C.c_uint32, C.POINTER(KmxList), C.POINTER(C.c_uint64),
                                     C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(KmxSuperkStats),
                                     C.POINTER(KmxSuperkRaw)]
_lib.kmx_hist_reset.argtypes = [_vp]
_lib.kmx_hist_off.argtypes = [_vp]
_lib.kmx_hist_read.argtypes = [_vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]
_lib.kmx_peer_access.argtypes = [C.c_int, C.c_int]
EXPORTS = ["kmx_copy_to_host_async", "kmx_copy_wait", "kmx_reads_upload", "kmx_reads_release", "kmx_peer_access", "kmx_set_file_order", "kmx_count_reads_dev_multi", "kmx_version", "kmx_result_prepare_body", "kmx_result_arena", "kmx_result_copy_order", "kmx_result_sparse_rows", "kmx_device_memory", "kmx_superk_sample", "kmx_store_create", "kmx_store_destroy", "kmx_store_used", "kmx_store_limit", "kmx_copy_to_host", "kmx_count_reads_dev", "kmx_hist_reset", "kmx_hist_read", "kmx_hist_off", "kmx_device_count", "kmx_count_reads", "kmx_result_copy_body_dev", "kmx_merge_host", "kmx_alloc_pinned", "kmx_free_pinned", "kmx_superk_partition_stats", "kmx_result_transpose_ms", "kmx_result_body_dev", "kmx_set_profiling", "kmx_result_kernel_ms", "kmx_result_kernel", "kmx_create", "kmx_destroy", "kmx_last_error", "kmx_stream", "kmx_merge_dev",
           "kmx_result_wait", "kmx_result_rows", "kmx_result_row_bytes", "kmx_result_body_bytes",
           "kmx_result_algo_bytes", "kmx_result_copy_body", "kmx_result_copy_stats", "kmx_result_free",
           "kmx_merge", "kmx_count_kmer", "kmx_count_hash", "kmx_count_batch", "kmx_transpose_bits", "kmx_superk_partition",
           "kmx_free"]


def key_words_of(k):
    """64-bit words of a k-mer key: ceil(k / 32) (kmer.hpp:215 m_n_data; io/kmer_file.hpp:84 kmer_slots) -- 3 for 65 ... 96, 4 for 97 ... 127"""
    return (k + 31) // 32


class KmxError(RuntimeError):
    pass


def pack_records(keys, counts, key_words=1):
    """(keys uint64[n] or [n, kw], counts uint32[n]) -> packed record bytes (key words + u32 count)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, key_words)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n = len(counts)
    rec = np.zeros((n, key_words * 2 + 1), dtype=np.uint32)
    rec[:, :key_words * 2] = keys.view(np.uint32).reshape(n, key_words * 2)
    rec[:, key_words * 2] = counts
    return rec


class Context:
    """One engine context = one GPU + one HIP stream (a merge task pool thread in the reference)."""

    def __init__(self, device=0):
        h = _vp()
        rc = _lib.kmx_create(device, C.byref(h))
        if rc != 0:
            raise KmxError(f"kmx_create failed ({rc}): {_lib.kmx_last_error(None).decode()}")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            _lib.kmx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise KmxError(f"{what} failed ({rc}): {_lib.kmx_last_error(self._h).decode()}")

    def set_profiling(self, on=True):
        self._check(_lib.kmx_set_profiling(self._h, 1 if on else 0), "kmx_set_profiling")

    def set_file_order(self, on=True):
        """COUNT / PA rows at their final place out of the column-blocked kernels (default), or where the kernels leave them"""
        self._check(_lib.kmx_set_file_order(self._h, 1 if on else 0), "kmx_set_file_order")

    @property
    def stream(self):
        return _lib.kmx_stream(self._h)

    @staticmethod
    def _task(lists_ptr_n, key_words, soft_min, rec_min, share_min, mode, lower, upper, bitw, rows_hint, keep):
        n = len(lists_ptr_n)
        arr = (KmxList * n)()
        for i, (p, cnt) in enumerate(lists_ptr_n):
            arr[i].recs = p
            arr[i].n = cnt
        sm = (C.c_uint32 * n)(*[int(x) for x in soft_min])
        keep.extend([arr, sm])
        t = KmxMergeTask()
        t.n_lists, t.key_words, t.lists, t.soft_min = n, key_words, arr, sm
        t.rec_min, t.share_min, t.mode, t.bitw = rec_min, share_min, mode, bitw
        t.lower, t.upper, t.rows_hint = lower, upper, rows_hint
        return t

    def merge(self, lists, key_words, soft_min, rec_min, share_min, mode, lower=0, upper=0, bitw=2, rows_hint=0):
        """Host-buffer merge of one partition (kmx_merge).  lists: [(keys, counts)] numpy arrays.
        -> (body bytes, rows, stats uint64[6, N])"""
        keep, recs = [], []
        for k, c in lists:
            r = pack_records(k, c, key_words)
            recs.append(r)
        t = self._task([(r.ctypes.data if len(r) else None, len(r)) for r in recs], key_words, soft_min, rec_min,
                       share_min, mode, lower, upper, bitw, rows_hint, keep)
        body, nb, rows = _vp(), C.c_uint64(), C.c_uint64()
        stats = np.zeros((STATS_ROWS, len(lists)), dtype=np.uint64)
        self._check(_lib.kmx_merge(self._h, C.byref(t), C.byref(body), C.byref(nb), C.byref(rows), stats.ctypes.data),
                    "kmx_merge")
        data = (C.string_at(body.value, nb.value) if nb.value < (1 << 31) else bytes((C.c_ubyte * nb.value).from_address(body.value))) if nb.value else b""      # (string_at takes an int)
        _lib.kmx_free(body)
        return data, rows.value, stats

    def _take(self, kp, cp, n, width):
        keys = np.ctypeslib.as_array((C.c_uint64 * (n * width)).from_address(kp.value)).copy().reshape(n, width) if n \
            else np.zeros((0, width), np.uint64)
        cnts = np.ctypeslib.as_array((C.c_uint32 * n).from_address(cp.value)).copy() if n else np.zeros(0, np.uint32)
        _lib.kmx_free(kp)
        _lib.kmx_free(cp)
        return keys, cnts

    def count_kmer(self, superk: bytes, k, hard_min):
        """kmx_count_kmer: super-k-mer record stream -> (canonical k-mers uint64[n, ceil(k/32)], counts)"""
        kp, cp, n = _vp(), _vp(), C.c_uint64()
        self._check(_lib.kmx_count_kmer(self._h, superk, len(superk), k, hard_min, C.byref(kp), C.byref(cp),
                                        C.byref(n)), "kmx_count_kmer")
        return self._take(kp, cp, n.value, key_words_of(k))

    def count_hash(self, superk: bytes, k, window, partition, hard_min):
        kp, cp, n = _vp(), _vp(), C.c_uint64()
        self._check(_lib.kmx_count_hash(self._h, superk, len(superk), k, window, partition, hard_min, C.byref(kp),
                                        C.byref(cp), C.byref(n)), "kmx_count_hash")
        keys, cnts = self._take(kp, cp, n.value, 1)
        return keys.reshape(-1), cnts

    def count_batch(self, streams, k, hard_min, window=0, partitions=None):
        """kmx_count_batch: the partition streams of one sample -> [(keys, counts)] per stream.
        window != 0 selects window hashes (partitions[p] = window index of stream p)."""
        n = len(streams)
        sp = (C.c_char_p * n)(*[s if s else None for s in streams])
        ln = (C.c_uint64 * n)(*[len(s) for s in streams])
        pid = (C.c_uint64 * n)(*(partitions if partitions is not None else range(n)))
        kp, cp, no = (_vp * n)(), (_vp * n)(), (C.c_uint64 * n)()
        self._check(_lib.kmx_count_batch(self._h, n, sp, ln, k, 1 if window else 0, window, pid, hard_min, kp, cp, no),
                    "kmx_count_batch")
        width = 1 if window else key_words_of(k)
        out = []
        for p in range(n):
            keys, cnts = self._take(_vp(kp[p]), _vp(cp[p]), no[p], width)
            out.append((keys.reshape(-1) if window else keys, cnts))
        return out

    def transpose_bits(self, mat, nrows, ncols):
        mat = np.ascontiguousarray(mat, dtype=np.uint8)
        out = np.zeros(ncols * (nrows // 8), dtype=np.uint8)
        self._check(_lib.kmx_transpose_bits(self._h, mat.ctypes.data, nrows, ncols, out.ctypes.data),
                    "kmx_transpose_bits")
        return out

    @staticmethod
    def pack_reads(reads):
        """list of reads (str/bytes) -> (concatenated bases, uint64 offsets[n + 1]) as kmx_superk_partition takes them"""
        bs = [r if isinstance(r, bytes) else r.encode() for r in reads]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        return b"".join(bs), offs

    def superk_partition(self, reads, k, m, repart, nb_parts):
        """kmx_superk_partition: list of reads (str/bytes), or pack_reads() output -> [(record stream bytes, n_kmers)]
        per partition"""
        blob, offs = reads if isinstance(reads, tuple) else self.pack_reads(reads)
        rep = np.ascontiguousarray(repart, dtype=np.uint16)
        ob = (_vp * nb_parts)()
        ol = (C.c_uint64 * nb_parts)()
        okm = (C.c_uint64 * nb_parts)()
        self._check(_lib.kmx_superk_partition(self._h, blob, offs.ctypes.data, len(offs) - 1, k, m, rep.ctypes.data, nb_parts,
                                              ob, ol, okm), "kmx_superk_partition")
        out = []
        for p in range(nb_parts):
            out.append((C.string_at(ob[p], ol[p]) if ol[p] else b"", int(okm[p])))
            _lib.kmx_free(ob[p])
        return out

    def superk_partition_stats(self, reads, k, m, repart, nb_parts, streams=True):
        """kmx_superk_partition_stats -> ([(record stream, n_kmers)] or None, pinfo[nb_parts, 2 + 5*256], minim_superks,
        minim_kmers, minim_kxmers); streams=False: the statistics-only pass"""
        blob, offs = reads if isinstance(reads, tuple) else self.pack_reads(reads)
        rep = np.ascontiguousarray(repart, dtype=np.uint16)
        pin = np.zeros(nb_parts * PINFO_STRIDE, dtype=np.uint64)
        ms, mk, mx = (np.zeros(4 ** m, dtype=np.uint64) for _ in range(3))
        st = KmxSuperkStats(pin.ctypes.data, ms.ctypes.data, mk.ctypes.data, mx.ctypes.data, 0)
        ob = (_vp * nb_parts)()
        ol = (C.c_uint64 * nb_parts)()
        okm = (C.c_uint64 * nb_parts)()
        self._check(_lib.kmx_superk_partition_stats(self._h, blob, offs.ctypes.data, len(offs) - 1, k, m, rep.ctypes.data,
                                                    nb_parts, ob if streams else None, ol, okm, C.byref(st)),
                    "kmx_superk_partition_stats")
        out = None
        if streams:
            out = []
            for p in range(nb_parts):
                out.append((C.string_at(ob[p], ol[p]) if ol[p] else b"", int(okm[p])))
                _lib.kmx_free(ob[p])
        assert st.nb_superk == int(ms.sum())
        return out, pin.reshape(nb_parts, -1), ms, mk, mx

    def count_reads(self, reads, k, m, repart, nb_parts, hard_min, window=0, streams=False):
        """kmx_count_reads: reads -> [(keys, counts)] per partition without the super-k-mer streams leaving HBM
        -> (counts per partition, k-mers per partition, [(stream, ...)] or None, info numbers uint64[nb_parts, 2])"""
        blob, offs = reads if isinstance(reads, tuple) else self.pack_reads(reads)
        rep = np.ascontiguousarray(repart, dtype=np.uint16)
        kp, cp, no, nk = (_vp * nb_parts)(), (_vp * nb_parts)(), (C.c_uint64 * nb_parts)(), (C.c_uint64 * nb_parts)()
        ob, ol = (_vp * nb_parts)(), (C.c_uint64 * nb_parts)()
        info = (C.c_uint64 * (2 * nb_parts))()
        self._check(_lib.kmx_count_reads(self._h, blob, offs.ctypes.data, len(offs) - 1, k, m, rep.ctypes.data, nb_parts,
                                         1 if window else 0, window, hard_min, kp, cp, no, nk, ob if streams else None,
                                         ol if streams else None, info, None), "kmx_count_reads")
        width = 1 if window else key_words_of(k)
        out = []
        for p in range(nb_parts):
            keys, cnts = self._take(_vp(kp[p]), _vp(cp[p]), no[p], width)
            out.append((keys.reshape(-1) if window else keys, cnts))
        st = None
        if streams:
            st = []
            for p in range(nb_parts):
                st.append(C.string_at(ob[p], ol[p]) if ol[p] else b"")
                _lib.kmx_free(ob[p])
        return out, [int(x) for x in nk], st, np.array(list(info), dtype=np.uint64).reshape(nb_parts, 2)

    def upload_reads(self, blob, offs=None):
        """kmx_reads_upload: the bases of a batch sent to the device ahead of the count call (from page-locked memory, on the context's
        upload stream) -> a handle for count_reads_dev(resident=...) and release_reads.  What `kmx pipeline` does with the NEXT
        sample while this one is counted.  offs: the reads' offsets are put behind the bases in the same page-locked block, as
        `kmx pipeline` does, and count_reads_dev(resident=handle) hands the call THAT array (a DMA, no staging copy inside the call)."""
        n = len(blob)
        at = (n + 63) & ~63
        nb = 0 if offs is None else 8 * len(offs)
        pin = _lib.kmx_alloc_pinned(max(at + nb, 1))
        C.memmove(pin, blob, n)
        po = None
        if offs is not None:
            po = np.frombuffer((C.c_char * nb).from_address(pin + at), dtype=np.uint64)
            po[:] = np.ascontiguousarray(offs, dtype=np.uint64)
        dev = _vp()
        self._check(_lib.kmx_reads_upload(self._h, pin, n, C.byref(dev)), "kmx_reads_upload")
        return (dev, pin, po)

    def release_reads(self, handle):
        _lib.kmx_reads_release(self._h, handle[0])
        _lib.kmx_free_pinned(handle[1])

    def count_reads_dev(self, reads, k, m, repart, nb_parts, hard_min, stores, window=0, raw=False, sparse=False, ahead=False, resident=None):
        """kmx_count_reads_dev: as count_reads, the results left on the device as packed records in `stores` (partition p ->
        stores[p % len(stores)]) -> ([(device pointer, records)] per partition, k-mers per partition, raw tables or None).
        resident: a handle of upload_reads for the same bases (the call is given the device pointer; the handle stays the caller's)"""
        blob, offs = reads if isinstance(reads, tuple) else self.pack_reads(reads)
        rep = np.ascontiguousarray(repart, dtype=np.uint16)
        sp = (_vp * len(stores))(*[s._h for s in stores])
        lists, nk = (KmxList * nb_parts)(), (C.c_uint64 * nb_parts)()
        rw, tabs = None, None
        spt = None
        if raw:
            tabs = (np.zeros(nb_parts * 1280, np.uint32), np.zeros(4 ** m, np.uint32), np.zeros(4 ** m, np.uint32))
            rw = KmxSuperkRaw(tabs[0].ctypes.data, tabs[1].ctypes.data, tabs[2].ctypes.data, 0, None, 0, 0)
            if sparse:      # the per-minimizer records as {minimizer, super-k-mers, k-mers} triples: turned back into the tables here
                spt = np.zeros((4 ** m, 3), np.uint32)
                rw = KmxSuperkRaw(tabs[0].ctypes.data, None, None, 0, spt.ctypes.data, 4 ** m, 0)
        dev = None
        if resident is not None:
            ahead = False
            blob_arg = C.cast(resident[0], C.c_char_p)
            if len(resident) > 2 and resident[2] is not None:
                offs = resident[2]
        elif ahead:      # kmx_reads_upload: the bases sent to the device ahead of the call (from page-locked memory), the call given the device pointer
            n = len(blob)
            pin = _lib.kmx_alloc_pinned(max(n, 1))
            C.memmove(pin, blob, n)
            dev = _vp()
            self._check(_lib.kmx_reads_upload(self._h, pin, n, C.byref(dev)), "kmx_reads_upload")
            blob_arg = C.cast(dev, C.c_char_p)
        elif resident is None:
            blob_arg = blob
        try:
            self._check(_lib.kmx_count_reads_dev(self._h, blob_arg, offs.ctypes.data, len(offs) - 1, k, m, rep.ctypes.data, nb_parts,
                                                 1 if window else 0, window, hard_min, sp, len(stores), lists, nk, None, None, None, None,
                                                 C.byref(rw) if raw else None), "kmx_count_reads_dev")
        finally:
            if ahead:
                _lib.kmx_reads_release(self._h, dev)
                _lib.kmx_free_pinned(pin)
        if spt is not None:
            t = spt[:int(rw.minim_sparse_n)]
            assert len(np.unique(t[:, 0])) == len(t)
            tabs[1][t[:, 0]] = t[:, 1]; tabs[2][t[:, 0]] = t[:, 2]
        return [(lists[p].recs, int(lists[p].n)) for p in range(nb_parts)], [int(x) for x in nk], (tabs + (int(rw.nb_superk),)) if raw else None

    def count_reads_dev_multi(self, samples, k, m, repart, nb_parts, hard_min, stores, window=0, raw=False):
        """kmx_count_reads_dev_multi: several samples (lists of reads) in ONE call -> per sample what count_reads_dev returns
        (raw: the sparse form, turned back into tables here)"""
        S = len(samples)
        packed = [self.pack_reads(r) for r in samples]
        rep = np.ascontiguousarray(repart, dtype=np.uint16)
        bp = (C.c_char_p * S)(*[b for b, _ in packed])
        op = (_vp * S)(*[o.ctypes.data for _, o in packed])
        ns = (C.c_uint64 * S)(*[len(o) - 1 for _, o in packed])
        sp = (_vp * len(stores))(*[s._h for s in stores])
        lists, nk = (KmxList * (S * nb_parts))(), (C.c_uint64 * (S * nb_parts))()
        info = np.zeros((S * nb_parts, 2), np.uint64)
        rws, tabs, spts = None, [], []
        if raw:
            rws = (KmxSuperkRaw * S)()
            for i in range(S):
                t = (np.zeros(nb_parts * 1280, np.uint32), np.zeros(4 ** m, np.uint32), np.zeros(4 ** m, np.uint32))
                q = np.zeros((4 ** m, 3), np.uint32)
                tabs.append(t); spts.append(q)
                rws[i] = KmxSuperkRaw(t[0].ctypes.data, None, None, 0, q.ctypes.data, 4 ** m, 0)
        self._check(_lib.kmx_count_reads_dev_multi(self._h, S, bp, op, ns, k, m, rep.ctypes.data, nb_parts, 1 if window else 0, window, hard_min,
                                                   sp, len(stores), lists, nk, info.ctypes.data, rws), "kmx_count_reads_dev_multi")
        out = []
        for i in range(S):
            r = None
            if raw:
                t = spts[i][:int(rws[i].minim_sparse_n)]
                assert len(np.unique(t[:, 0])) == len(t)
                tabs[i][1][t[:, 0]] = t[:, 1]; tabs[i][2][t[:, 0]] = t[:, 2]
                r = tabs[i] + (int(rws[i].nb_superk),)
            out.append(([(lists[i * nb_parts + p].recs, int(lists[i * nb_parts + p].n)) for p in range(nb_parts)],
                        [int(nk[i * nb_parts + p]) for p in range(nb_parts)], info[i * nb_parts:(i + 1) * nb_parts], r))
        return out

    def read_list(self, dev_ptr, n, key_words=1):
        """a device-resident count list -> (keys uint64[n, key_words], counts uint32[n])"""
        w = key_words * 2 + 1
        rec = np.zeros((n, w), np.uint32)
        if n:
            self._check(_lib.kmx_copy_to_host(self._h, rec.ctypes.data, dev_ptr, n * w * 4), "kmx_copy_to_host")
        keys = np.ascontiguousarray(rec[:, :key_words * 2]).view(np.uint64).reshape(n, key_words)
        return keys, rec[:, key_words * 2].copy()

    def hist_reset(self):
        """kmx_hist_reset: zero the abundance histogram; the count calls that follow add their distinct keys to it"""
        self._check(_lib.kmx_hist_reset(self._h), "kmx_hist_reset")

    def hist_off(self):
        self._check(_lib.kmx_hist_off(self._h), "kmx_hist_off")

    def hist_read(self, lower=1, upper=255):
        """kmx_hist_read -> dict(unique, total (bins lower..upper), oob [lower unique, upper unique, lower total, upper total], sums [unique, total])"""
        n = upper - lower + 1
        h = dict(unique=np.zeros(n, np.uint64), total=np.zeros(n, np.uint64), oob=np.zeros(4, np.uint64), sums=np.zeros(2, np.uint64))
        self._check(_lib.kmx_hist_read(self._h, lower, upper, h["unique"].ctypes.data, h["total"].ctypes.data, h["oob"].ctypes.data, h["sums"].ctypes.data), "kmx_hist_read")
        return h

    def superk_sample(self, reads, k, m, budget):
        """kmx_superk_sample -> (reads used, their super-k-mers, kx-mers per minimizer)"""
        blob, offs = reads if isinstance(reads, tuple) else self.pack_reads(reads)
        mx = np.zeros(4 ** m, dtype=np.uint64)
        st = KmxSuperkStats(None, None, None, mx.ctypes.data, 0)
        used, nsk = C.c_uint64(), C.c_uint64()
        self._check(_lib.kmx_superk_sample(self._h, blob, offs.ctypes.data, len(offs) - 1, k, m, budget, C.byref(st),
                                           C.byref(used), C.byref(nsk)), "kmx_superk_sample")
        return used.value, nsk.value, mx

    def prepare(self, tasks):
        """Builds the kmx_merge_task array once (so a timed loop does no Python marshalling).
        tasks: list of dicts with keys lists=[(device_ptr, n)], key_words, soft_min, rec_min, share_min,
        mode, [lower, upper, bitw, rows_hint]."""
        keep = []
        arr = (KmxMergeTask * len(tasks))()
        for i, d in enumerate(tasks):
            arr[i] = self._task(d["lists"], d["key_words"], d["soft_min"], d["rec_min"], d["share_min"], d["mode"],
                                d.get("lower", 0), d.get("upper", 0), d.get("bitw", 2), d.get("rows_hint", 0), keep)
        return (arr, len(tasks), [len(d["lists"]) for d in tasks], keep)

    def merge_host(self, tasks):
        """kmx_merge_host: like merge_dev with HOST record pointers (uploaded by libkmx on its upload stream)"""
        prep = tasks if isinstance(tasks, tuple) else self.prepare(tasks)
        res = _vp()
        self._check(_lib.kmx_merge_host(self._h, prep[0], prep[1], C.byref(res)), "kmx_merge_host")
        return MergeResult(self, res, prep[2])

    def merge_dev(self, tasks):
        """Device-resident batch merge (kmx_merge_dev) of a task list or a prepare()d batch.
        -> MergeResult (asynchronous; call .wait())."""
        prep = tasks if isinstance(tasks, tuple) else self.prepare(tasks)
        res = _vp()
        self._check(_lib.kmx_merge_dev(self._h, prep[0], prep[1], C.byref(res)), "kmx_merge_dev")
        return MergeResult(self, res, prep[2])


class MergeResult:
    def __init__(self, ctx, h, n_lists):
        self._ctx, self._h, self._n = ctx, h, n_lists

    def wait(self):
        self._ctx._check(_lib.kmx_result_wait(self._h), "kmx_result_wait")

    def kernel_ms(self):
        return _lib.kmx_result_kernel_ms(self._h)

    def kernel_parts_ms(self):
        """(k_merge_cols ms, k_cols_sparse ms) of a result of the column-blocked pair; (-1, -1) otherwise"""
        a, b = C.c_double(-1.0), C.c_double(-1.0)
        self._ctx._check(_lib.kmx_result_kernel_parts_ms(self._h, C.byref(a), C.byref(b)), "kmx_result_kernel_parts_ms")
        return a.value, b.value

    def kernel(self):
        return _lib.kmx_result_kernel(self._h).decode()

    def transpose_ms(self):
        return _lib.kmx_result_transpose_ms(self._h)

    def body_dev(self, t=0):
        """device address of a BF / BFC / BFT body (None for COUNT / PA)"""
        return _lib.kmx_result_body_dev(self._h, t)

    def rows(self, t=0):
        return _lib.kmx_result_rows(self._h, t)

    def sparse_rows(self, t=0):
        return _lib.kmx_result_sparse_rows(self._h, t)

    def row_bytes(self, t=0):
        return _lib.kmx_result_row_bytes(self._h, t)

    def body_bytes(self, t=0):
        return _lib.kmx_result_body_bytes(self._h, t)

    def algo_bytes(self, t=0):
        return _lib.kmx_result_algo_bytes(self._h, t)

    def body(self, t=0):
        nb = self.body_bytes(t)
        buf = np.zeros(max(nb, 1), dtype=np.uint8)
        self._ctx._check(_lib.kmx_result_copy_body(self._h, t, buf.ctypes.data, nb), "kmx_result_copy_body")
        return buf[:nb].tobytes()

    def body_from_arena(self, t=0):
        """the body put together on the HOST from the arena and the order of its rows (kmx_result_arena + kmx_result_copy_order):
        what the pipeline's file writer does with pwrite"""
        rows, rb = self.rows(t), self.row_bytes(t)
        if rows == 0:
            return b""
        dev, nar = _vp(), C.c_uint64()
        self._ctx._check(_lib.kmx_result_arena(self._h, t, C.byref(dev), C.byref(nar)), "kmx_result_arena")
        arena = np.zeros((nar.value, rb), np.uint8)
        self._ctx._check(_lib.kmx_copy_to_host(self._ctx._h, arena.ctypes.data, dev, nar.value * rb), "kmx_copy_to_host")
        order = np.zeros(rows, np.uint32)
        self._ctx._check(_lib.kmx_result_copy_order(self._h, t, order.ctypes.data), "kmx_result_copy_order")
        return arena[order].tobytes()

    def body_to_device(self, t, dev_ptr, nbytes):
        """BF / BFC / BFT body -> device memory of the caller (a torch tensor's data_ptr())"""
        self._ctx._check(_lib.kmx_result_copy_body_dev(self._h, t, dev_ptr, nbytes), "kmx_result_copy_body_dev")

    def stats(self, t=0):
        st = np.zeros((STATS_ROWS, self._n[t]), dtype=np.uint64)
        self._ctx._check(_lib.kmx_result_copy_stats(self._h, t, st.ctypes.data), "kmx_result_copy_stats")
        return st

    def free(self):
        if self._h:
            _lib.kmx_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Store:
    """kmx_store: count lists resident in HBM between the count and the merge stage"""

    def __init__(self, device=0, limit_bytes=0):
        h = _vp()
        rc = _lib.kmx_store_create(device, limit_bytes, C.byref(h))
        if rc != 0:
            raise KmxError(f"kmx_store_create failed ({rc}): {_lib.kmx_last_error(None).decode()}")
        self._h = h

    def used(self):
        return _lib.kmx_store_used(self._h)

    def limit(self):
        return _lib.kmx_store_limit(self._h)

    def close(self):
        if self._h:
            _lib.kmx_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
