"""Minimal numpy readers for the kmtricks on-disk formats used by the tests
(independent of the product's C++ format library).  Layouts: SURVEY.md Appendix A;
reference include/kmtricks/io/{io_common,kmer_file,hash_file}.hpp."""
import struct
import numpy as np

KM_MAGIC = 0x736b636972746d6b
KMER_MAGIC = 0x72656d6b
HASH_MAGIC = 0x68736168


def read_kmer_file(path):
    raw = open(path, "rb").read()
    magic, ver, comp, kmagic, k, slots, cslots, sid, part = struct.unpack_from("<QIBQIIIII", raw, 0)
    assert magic == KM_MAGIC and kmagic == KMER_MAGIC and comp == 0
    rec = slots * 8 + cslots
    body = raw[41:]
    n = len(body) // rec
    a = np.frombuffer(body, dtype=np.uint8).reshape(n, rec)
    keys = a[:, :slots * 8].copy().view(np.uint64).reshape(n, slots)
    cnt = np.zeros(n, dtype=np.uint32)
    for b in range(cslots):
        cnt |= a[:, slots * 8 + b].astype(np.uint32) << (8 * b)
    return dict(k=k, slots=slots, count_slots=cslots, id=sid, partition=part, keys=keys, counts=cnt)


def read_hash_file(path):
    raw = open(path, "rb").read()
    magic, ver, comp, hmagic, cslots, sid, part = struct.unpack_from("<QIBQIII", raw, 0)
    assert magic == KM_MAGIC and hmagic == HASH_MAGIC and comp == 0
    off = 33
    hs, cs = [], []
    cdt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[cslots]
    while off < len(raw):
        (n,) = struct.unpack_from("<Q", raw, off); off += 8
        hs.append(np.frombuffer(raw, dtype=np.uint64, count=n, offset=off)); off += 8 * n
        cs.append(np.frombuffer(raw, dtype=cdt, count=n, offset=off).astype(np.uint32)); off += cslots * n
    keys = np.concatenate(hs) if hs else np.zeros(0, np.uint64)
    cnt = np.concatenate(cs) if cs else np.zeros(0, np.uint32)
    return dict(count_slots=cslots, id=sid, partition=part, keys=keys, counts=cnt)
