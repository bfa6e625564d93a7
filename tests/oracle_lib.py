"""ctypes binding of the CPU oracle (oracle/sbwt_oracle.h).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "build", "libsbwt_oracle.so")
CLI = os.path.join(ORACLE_DIR, "build", "sbwt_oracle")

HIT_DTYPE = np.dtype([("pos0", "<u8"), ("chr_id", "<u4"), ("is_rc", "<u4")])
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp, u64 = C.c_void_p, C.c_uint64
        L.sbo_last_error.restype = C.c_char_p
        L.sbo_build.argtypes = [C.c_char_p, C.c_char_p, u64]
        L.sbo_index_load.argtypes = [C.c_char_p]
        L.sbo_index_load.restype = vp
        L.sbo_index_free.argtypes = [vp]
        L.sbo_index_free.restype = None
        L.sbo_occ.argtypes = [vp, u64, C.c_int]
        L.sbo_occ.restype = u64
        L.sbo_map_batch.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, u64]
        L.sbo_map_batch.restype = C.c_int64
        L.sbo_map_fastq.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(u64 * 5)]
        L.sbo_index_wrap.argtypes = [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp, vp, u64, u64, vp, u64,
                                     vp, vp]
        L.sbo_index_wrap.restype = vp
        _lib = L
    return _lib


def build_index(fasta, prefix, interval=64):
    rc = lib().sbo_build(os.fsencode(fasta), os.fsencode(prefix), interval)
    if rc != 0:
        raise RuntimeError(lib().sbo_last_error().decode())


class OracleIndex:
    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    @classmethod
    def load(cls, prefix):
        h = lib().sbo_index_load(os.fsencode(prefix))
        if not h:
            raise RuntimeError(lib().sbo_last_error().decode())
        return cls(h)

    def map_batch(self, reads, limit=1000, strand=2, nthreads=1):
        """reads: list of ASCII strings -> (lo, hi, nh, first, hits)"""
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        n = len(bs)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        if n:
            offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        seqs = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
        return self.map_batch_raw(seqs, offsets, n, limit, strand, nthreads)

    def map_batch_raw(self, seqs, offsets, n, limit=1000, strand=2, nthreads=1, hits_cap=None):
        lo = np.zeros(n, dtype=np.uint64); hi = np.zeros(n, dtype=np.uint64)
        nh = np.zeros(n, dtype=np.int32); first = np.zeros(n, dtype=np.uint64)
        cap = int(hits_cap or max(4 * n, 1 << 16))
        while True:
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            tot = lib().sbo_map_batch(self._h, seqs.ctypes.data, offsets.ctypes.data, n, limit, strand, nthreads,
                                      lo.ctypes.data, hi.ctypes.data, nh.ctypes.data, first.ctypes.data,
                                      hits.ctypes.data, cap)
            if tot == -3:
                cap *= 4
                continue
            if tot < 0:
                raise RuntimeError("oracle status %d: %s" % (tot, lib().sbo_last_error().decode()))
            return lo, hi, nh, first, hits[:tot]

    def occ(self, i, c):
        return int(lib().sbo_occ(self._h, int(i), int(c)))

    def map_fastq(self, fastq, sam, limit=1000, strand=2):
        st = (C.c_uint64 * 5)()
        rc = lib().sbo_map_fastq(self._h, os.fsencode(fastq), os.fsencode(sam), limit, strand, C.byref(st))
        return rc, list(st)

    def close(self):
        if self._h:
            lib().sbo_index_free(self._h)
            self._h = None
