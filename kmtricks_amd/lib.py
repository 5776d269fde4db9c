"""ctypes binding of libkmx.so (include/kmx.h).  No fallback of any kind."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkmx.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(libkmx has no CPU fallback)")
_lib = C.CDLL(LIB_PATH)

MODE_COUNT, MODE_PA, MODE_BF, MODE_BFC = 0, 1, 2, 3
STATS_ROWS = 6


class KmxList(C.Structure):
    _fields_ = [("recs", C.c_void_p), ("n", C.c_uint64)]


class KmxMergeTask(C.Structure):
    _fields_ = [("n_lists", C.c_uint32), ("key_words", C.c_uint32), ("lists", C.POINTER(KmxList)),
                ("soft_min", C.POINTER(C.c_uint32)), ("rec_min", C.c_uint32), ("share_min", C.c_uint32),
                ("mode", C.c_uint32), ("bitw", C.c_uint32), ("lower", C.c_uint64), ("upper", C.c_uint64),
                ("rows_hint", C.c_uint64)]


_vp = C.c_void_p
_lib.kmx_version.restype = C.c_int
_lib.kmx_create.argtypes = [C.c_int, C.POINTER(_vp)]
_lib.kmx_destroy.argtypes = [_vp]
_lib.kmx_last_error.restype = C.c_char_p
_lib.kmx_last_error.argtypes = [_vp]
_lib.kmx_stream.restype = _vp
_lib.kmx_stream.argtypes = [_vp]
_lib.kmx_merge_dev.argtypes = [_vp, C.POINTER(KmxMergeTask), C.c_uint32, C.POINTER(_vp)]
_lib.kmx_result_wait.argtypes = [_vp]
for _f in ("kmx_result_rows", "kmx_result_row_bytes", "kmx_result_body_bytes", "kmx_result_algo_bytes"):
    getattr(_lib, _f).restype = C.c_uint64
    getattr(_lib, _f).argtypes = [_vp, C.c_uint32]
_lib.kmx_result_copy_body.argtypes = [_vp, C.c_uint32, _vp, C.c_uint64]
_lib.kmx_result_copy_stats.argtypes = [_vp, C.c_uint32, _vp]
_lib.kmx_result_free.argtypes = [_vp]
_lib.kmx_merge.argtypes = [_vp, C.POINTER(KmxMergeTask), C.POINTER(_vp), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_uint64), _vp]
_lib.kmx_free.argtypes = [_vp]

EXPORTS = ["kmx_version", "kmx_create", "kmx_destroy", "kmx_last_error", "kmx_stream", "kmx_merge_dev",
           "kmx_result_wait", "kmx_result_rows", "kmx_result_row_bytes", "kmx_result_body_bytes",
           "kmx_result_algo_bytes", "kmx_result_copy_body", "kmx_result_copy_stats", "kmx_result_free",
           "kmx_merge", "kmx_count_kmer", "kmx_count_hash", "kmx_transpose_bits", "kmx_superk_partition",
           "kmx_free"]


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
        data = C.string_at(body.value, nb.value) if nb.value else b""
        _lib.kmx_free(body)
        return data, rows.value, stats

    def merge_dev(self, tasks):
        """Device-resident batch merge (kmx_merge_dev).  tasks: list of dicts with keys
        lists=[(device_ptr, n)], key_words, soft_min, rec_min, share_min, mode, [lower, upper, bitw, rows_hint].
        -> MergeResult (asynchronous; call .wait())."""
        keep = []
        arr = (KmxMergeTask * len(tasks))()
        for i, d in enumerate(tasks):
            arr[i] = self._task(d["lists"], d["key_words"], d["soft_min"], d["rec_min"], d["share_min"], d["mode"],
                                d.get("lower", 0), d.get("upper", 0), d.get("bitw", 2), d.get("rows_hint", 0), keep)
        res = _vp()
        self._check(_lib.kmx_merge_dev(self._h, arr, len(tasks), C.byref(res)), "kmx_merge_dev")
        return MergeResult(self, res, [len(d["lists"]) for d in tasks])


class MergeResult:
    def __init__(self, ctx, h, n_lists):
        self._ctx, self._h, self._n = ctx, h, n_lists

    def wait(self):
        self._ctx._check(_lib.kmx_result_wait(self._h), "kmx_result_wait")

    def rows(self, t=0):
        return _lib.kmx_result_rows(self._h, t)

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
