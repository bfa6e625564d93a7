"""ctypes binding of include/sbwt_amd.h (one Python wrapper per C entry point, same names)."""
import ctypes as C
import importlib.util
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RF_SUPPRESSED = 1
RF_TRUNCATED = 2


class SbwtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sbwt error %d: %s" % (code, msg))
        self.code = code


class Hit(C.Structure):
    _fields_ = [("pos0", C.c_uint64), ("chr_id", C.c_uint32), ("is_rc", C.c_uint32)]


class ReadResult(C.Structure):
    _fields_ = [("sa_lo", C.c_uint64), ("sa_hi", C.c_uint64), ("first_hit", C.c_uint64),
                ("nh", C.c_int32), ("flags", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("real_size", C.c_uint64), ("interval", C.c_uint64),
                ("first_location", C.c_uint64), ("n_chr", C.c_uint64), ("n_npos", C.c_uint64),
                ("n_loc", C.c_uint64), ("device_bytes", C.c_uint64), ("device", C.c_int),
                ("kmer_order", C.c_uint32), ("hash_order", C.c_uint32), ("hash_text", C.c_uint32)]


class HostTables(C.Structure):
    """sbwt_host_tables: reference-format tables in host memory (owned by the library)."""
    _fields_ = [("interval", C.c_uint64), ("n", C.c_uint64), ("first_location", C.c_uint64),
                ("real_size", C.c_uint64), ("occ_len", C.c_uint64), ("n_loc", C.c_uint64),
                ("c_function4", C.c_uint64 * 4), ("c_functions", C.c_void_p), ("occ", C.c_void_p * 4),
                ("location_table", C.c_void_p), ("jbwt_seq", C.c_void_p), ("fbwt", C.c_void_p)]


HIT_DTYPE = np.dtype([("pos0", "<u8"), ("chr_id", "<u4"), ("is_rc", "<u4")])
RESULT_DTYPE = np.dtype([("sa_lo", "<u8"), ("sa_hi", "<u8"), ("first_hit", "<u8"), ("nh", "<i4"), ("flags", "<u4")])

COMPACT_DTYPE = np.dtype([("w0", "<u8"), ("w1", "<u8")])
WIDE_DTYPE = np.dtype([("r", RESULT_DTYPE), ("hit", HIT_DTYPE)])
MAP_HITS_ONLY = 1
CR_UNIQUE, CR_RC, CR_MULTI, CR_SUPPRESSED, CR_TRUNCATED, CR_INVALID, CR_WIDE, CR_NO_SA = 1, 2, 4, 8, 16, 32, 64, 128

_lib = None


def lib_path():
    return os.path.join(_HERE, "lib", "libsbwt_amd.so")


def _one_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's) next to
    their own HSA runtime.  libsbwt_amd.so needs `libamdhip64.so.7`: loaded after torch it binds to torch's copy (one
    runtime, fine); loaded BEFORE torch it brings in /opt/rocm's, torch then loads its own by file name, and the
    second runtime to initialise sees no device.  So when torch is installed and not yet imported, its runtime is
    loaded here first: whatever the import order, the process ends up with that one copy.
    SBWT_HIP_RUNTIME=system keeps the system runtime (hosts that never import torch)."""
    if os.environ.get("SBWT_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def lib():
    """Loads libsbwt_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SbwtError(-3, "%s is missing: run `python -m sbwt_amd.build` (hipcc, gfx950)" % path)
    _one_hip_runtime()
    L = C.CDLL(path)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    p = C.POINTER
    L.sbwt_last_error.restype = C.c_char_p
    L.sbwt_device_count.argtypes = [p(C.c_int)]
    L.sbwt_index_load.argtypes = [C.c_char_p, i32, p(vp)]
    L.sbwt_index_from_tables.argtypes = [i32, u64, u64, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64, vp,
                                         vp, u64, vp, vp, p(vp)]
    L.sbwt_index_free.argtypes = [vp]
    L.sbwt_index_free.restype = None
    L.sbwt_index_get_info.argtypes = [vp, p(IndexInfo)]
    L.sbwt_index_chr_name.argtypes = [vp, u64]
    L.sbwt_index_chr_name.restype = C.c_char_p
    L.sbwt_build_index.argtypes = [C.c_char_p, C.c_char_p, u64, u64, i32]
    L.sbwt_index_build_device.argtypes = [i32, vp, u64, u64, u64, vp, vp, u64, vp, vp, p(vp), p(p(HostTables))]
    L.sbwt_host_tables_free.argtypes = [p(HostTables)]
    L.sbwt_host_tables_free.restype = None
    L.sbwt_host_tables_save.argtypes = [p(HostTables), C.c_char_p]
    L.sbwt_ctx_create.argtypes = [vp, u64, p(vp)]
    L.sbwt_ctx_free.argtypes = [vp]
    L.sbwt_ctx_free.restype = None
    L.sbwt_shard_range.argtypes = [u64, u32, u32, p(u64), p(u64)]
    L.sbwt_words_per_read.argtypes = [u32]
    L.sbwt_words_per_read.restype = u32
    L.sbwt_pack_reads.argtypes = [vp, vp, u64, u32, vp, vp, p(u64)]
    L.sbwt_map_batch_device.argtypes = [vp, vp, vp, u64, u32, i32, i32, vp, vp, u64, p(u64)]
    L.sbwt_map_batch.argtypes = [vp, vp, vp, u64, u32, i32, i32, vp, vp, u64, p(u64)]
    L.sbwt_map_batch_device_compact.argtypes = [vp, vp, vp, u64, u32, i32, i32, u32, vp, vp, u64, vp, u64, p(u64 * 2)]
    L.sbwt_map_batch_compact.argtypes = [vp, vp, vp, u64, u32, i32, i32, u32, vp, vp, u64, vp, u64, p(u64 * 2)]
    L.sbwt_compact_expand.argtypes = [vp, u64, vp, u64, vp, u64, vp, vp]
    L.sbwt_map_batch_device_static.argtypes = [vp, vp, vp, vp, u64, u32, i32, i32, vp, vp, u64, p(u64)]
    L.sbwt_host_alloc.argtypes = [u64, p(vp)]
    L.sbwt_host_free.argtypes = [vp]
    L.sbwt_host_free.restype = None
    L.sbwt_ctx_last_timing.argtypes = [vp, p(C.c_float * 3)]
    L.sbwt_ctx_last_stats.argtypes = [vp, p(u64 * 4)]
    L.sbwt_ctx_last_timing_ex.argtypes = [vp, p(C.c_float * 5), u32]
    L.sbwt_ctx_last_stats_ex.argtypes = [vp, p(u64 * 14), u32]
    L.sbwt_probe_random_access.argtypes = [vp, u32, u32, u32, p(C.c_double), p(u64)]
    L.sbwt_probe_gather_stream.argtypes = [vp, u64, u32, p(C.c_double)]
    L.sbwt_ctx_stream.argtypes = [vp]
    L.sbwt_ctx_stream.restype = vp
    L.sbwt_map_fastq.argtypes = [vp, C.c_char_p, C.c_char_p, i32, i32, i32, p(u64 * 5)]
    L.sbwt_map_fastq_multi.argtypes = [p(vp), i32, C.c_char_p, C.c_char_p, i32, i32, i32, p(u64 * 5)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise SbwtError(rc, lib().sbwt_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    _check(lib().sbwt_device_count(C.byref(n)))
    return n.value


class PinnedBuffer:
    """sbwt_host_alloc: page-locked host memory, viewed as numpy arrays (keep the buffer alive while they are used)"""

    def __init__(self, nbytes):
        ptr = C.c_void_p()
        _check(lib().sbwt_host_alloc(int(nbytes), C.byref(ptr)))
        self.ptr, self.nbytes = ptr.value, int(nbytes)

    def array(self, dtype, count, offset=0):
        dtype = np.dtype(dtype)
        assert offset + count * dtype.itemsize <= self.nbytes
        raw = (C.c_char * (count * dtype.itemsize)).from_address(self.ptr + offset)
        return np.frombuffer(raw, dtype=dtype, count=count)

    def close(self):
        if self.ptr:
            lib().sbwt_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(n_items, parts, k):
    """sbwt_shard_range: contiguous share k of `parts` of n_items"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    _check(lib().sbwt_shard_range(int(n_items), int(parts), int(k), C.byref(a), C.byref(b)))
    return a.value, b.value


def words_per_read(max_len):
    return int(lib().sbwt_words_per_read(int(max_len)))


def pack_reads(reads, wpr=None):
    """reads: list of ASCII strings/bytes -> (packed uint32 [n, wpr], lens uint32 [n])."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    if wpr is None:
        wpr = words_per_read(max((len(b) for b in bs), default=1))
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    seqs = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
    packed = np.zeros((len(bs), wpr), dtype=np.uint32)
    lens = np.zeros(len(bs), dtype=np.uint32)
    bad = C.c_uint64(0)
    _check(lib().sbwt_pack_reads(seqs.ctypes.data, offsets.ctypes.data, len(bs), wpr, packed.ctypes.data,
                                 lens.ctypes.data, C.byref(bad)))
    return packed, lens


def build_index(fasta, prefix, sort_len=256, interval=64, device=0):
    _check(lib().sbwt_build_index(os.fsencode(fasta), os.fsencode(prefix), sort_len, interval, device))


def build_index_device(d_fwd_packed, g, chr_start, npos_key, npos_val, interval=64, device=0, chr_names=None,
                       want_index=True, want_tables=False):
    """sbwt_index_build_device: d_fwd_packed is a raw device address. Returns (Index|None, HostTables ptr|None)."""
    cs = np.ascontiguousarray(chr_start, dtype=np.uint64)
    nk = np.ascontiguousarray(npos_key, dtype=np.uint64)
    nv = np.ascontiguousarray(npos_val, dtype=np.uint64)
    names = None
    if chr_names is not None:
        arr = (C.c_char_p * len(chr_names))(*[s.encode() for s in chr_names])
        names = C.cast(arr, C.c_void_p)
    h = C.c_void_p()
    t = C.POINTER(HostTables)()
    _check(lib().sbwt_index_build_device(device, d_fwd_packed, g, interval, cs.size, cs.ctypes.data, names, nk.size,
                                         nk.ctypes.data, nv.ctypes.data, C.byref(h) if want_index else None,
                                         C.byref(t) if want_tables else None))
    return (Index(h) if want_index else None), (t if want_tables else None)


def compact_expand(rec, list_hits, wide):
    """sbwt_compact_expand: compact records -> (results, hits) in the plain layout of Context.map_batch"""
    rec = np.ascontiguousarray(rec, dtype=COMPACT_DTYPE)
    list_hits = np.ascontiguousarray(list_hits, dtype=HIT_DTYPE)
    wide = np.ascontiguousarray(wide, dtype=WIDE_DTYPE)
    n = rec.size
    res = np.zeros(n, dtype=RESULT_DTYPE)
    hits = np.zeros(n + list_hits.size, dtype=HIT_DTYPE)
    _check(lib().sbwt_compact_expand(rec.ctypes.data, n, wide.ctypes.data if wide.size else None, wide.size,
                                     list_hits.ctypes.data if list_hits.size else None, list_hits.size, res.ctypes.data,
                                     hits.ctypes.data))
    return res, hits


def map_fastq_multi(indexes, fastq, sam, limit=1000, strand=2, n_threads=1):
    """sbwt_map_fastq_multi: `indexes` are replicas of one index, one per device"""
    arr = (C.c_void_p * len(indexes))(*[ix._h for ix in indexes])
    st = (C.c_uint64 * 5)()
    _check(lib().sbwt_map_fastq_multi(arr, len(indexes), os.fsencode(fastq), os.fsencode(sam), limit, strand, n_threads,
                                      C.byref(st)))
    return list(st)


def host_tables_free(t):
    lib().sbwt_host_tables_free(t)


def host_tables_save(t, path):
    _check(lib().sbwt_host_tables_save(t, os.fsencode(path)))


class Index:
    """sbwt_index handle (Aligner_table<BWT_Aligner> resident in HBM)."""

    def __init__(self, handle):
        self._h = handle
        self._contexts = weakref.WeakSet()   # a ctx must not outlive its index (sbwt_ctx_free reads it): closed first

    @classmethod
    def load(cls, prefix, device=0):
        h = C.c_void_p()
        _check(lib().sbwt_index_load(os.fsencode(prefix), device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_tables(cls, device, interval, n, first_location, real_size, c_function4, c_functions, occ, jbwt_seq,
                    fbwt, location_table, chr_start, npos_key, npos_val, chr_names=None):
        def a(x, dt):
            return np.ascontiguousarray(x, dtype=dt)
        c4 = a(c_function4, np.uint64); cfs = a(c_functions, np.uint64)
        occs = [a(o, np.uint64) for o in occ]
        jb = a(jbwt_seq, np.uint8); fb = a(fbwt, np.uint8); loc = a(location_table, np.uint64)
        cs = a(chr_start, np.uint64); nk = a(npos_key, np.uint64); nv = a(npos_val, np.uint64)
        names = None
        if chr_names is not None:
            arr = (C.c_char_p * len(chr_names))(*[s.encode() for s in chr_names])
            names = C.cast(arr, C.c_void_p)
        h = C.c_void_p()
        _check(lib().sbwt_index_from_tables(device, interval, n, first_location, real_size, c4.ctypes.data,
                                            cfs.ctypes.data, occs[0].ctypes.data, occs[1].ctypes.data,
                                            occs[2].ctypes.data, occs[3].ctypes.data, jb.ctypes.data, fb.ctypes.data,
                                            loc.ctypes.data, loc.size // 2, cs.size, cs.ctypes.data, names, nk.size,
                                            nk.ctypes.data, nv.ctypes.data, C.byref(h)))
        return cls(h)

    def info(self):
        i = IndexInfo()
        _check(lib().sbwt_index_get_info(self._h, C.byref(i)))
        return i

    def probe_random_access(self, chains_per_lane=1, chain_len=64, workgroups_per_cu=8):
        """sbwt_probe_random_access -> (line accesses per second, bytes of the table probed)"""
        rate, nbytes = C.c_double(0.0), C.c_uint64(0)
        _check(lib().sbwt_probe_random_access(self._h, chains_per_lane, chain_len, workgroups_per_cu, C.byref(rate),
                                              C.byref(nbytes)))
        return rate.value, nbytes.value

    def probe_gather_stream(self, n_accesses, workgroups_per_cu=6):
        """sbwt_probe_gather_stream -> independent random line accesses per second, records streamed in and out"""
        rate = C.c_double(0.0)
        _check(lib().sbwt_probe_gather_stream(self._h, n_accesses, workgroups_per_cu, C.byref(rate)))
        return rate.value

    def chr_names(self):
        return [lib().sbwt_index_chr_name(self._h, k).decode() for k in range(self.info().n_chr)]

    def map_fastq(self, fastq, sam, limit=1000, strand=2, n_threads=1):
        st = (C.c_uint64 * 5)()
        _check(lib().sbwt_map_fastq(self._h, os.fsencode(fastq), os.fsencode(sam), limit, strand, n_threads,
                                    C.byref(st)))
        return list(st)

    def map_batch_device_static(self, d_packed, d_lens, n_reads, wpr, d_results, d_hits, hits_cap, limit=1000,
                                strand=2, stream=None):
        """sbwt_map_batch_device_static: device pointers, no ctx, allocates nothing (static scratch of the device)."""
        nh = C.c_uint64(0)
        _check(lib().sbwt_map_batch_device_static(self._h, stream, d_packed, d_lens, n_reads, wpr, limit, strand,
                                                  d_results, d_hits, hits_cap, C.byref(nh)))
        return nh.value

    def close(self):
        if self._h:
            for ctx in list(self._contexts):
                ctx.close()
            lib().sbwt_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """sbwt_ctx handle: one stream + workspace."""

    def __init__(self, index, max_reads):
        self.index = index
        self._h = C.c_void_p()
        _check(lib().sbwt_ctx_create(index._h, int(max_reads), C.byref(self._h)))
        index._contexts.add(self)

    def map_batch(self, packed, lens, limit=1000, strand=2, hits_cap=None, out=None):
        """Host-buffer entry point. Returns (results structured array, hits structured array).
        out = (results, hits) arrays to fill instead of fresh ones (e.g. views of a PinnedBuffer)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = lens.size
        wpr = packed.shape[1] if packed.ndim == 2 else packed.size // max(n, 1)
        if out is not None:
            res, hits = out
            assert res.dtype == RESULT_DTYPE and hits.dtype == HIT_DTYPE and res.size >= n
            nh = C.c_uint64(0)
            _check(lib().sbwt_map_batch(self._h, packed.ctypes.data, lens.ctypes.data, n, wpr, limit, strand,
                                        res.ctypes.data, hits.ctypes.data, hits.size, C.byref(nh)))
            return res[:n], hits[: nh.value]
        res = np.zeros(n, dtype=RESULT_DTYPE)
        cap = int(hits_cap if hits_cap is not None else max(2 * n, 1024))
        while True:
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            nh = C.c_uint64(0)
            rc = lib().sbwt_map_batch(self._h, packed.ctypes.data, lens.ctypes.data, n, wpr, limit, strand,
                                      res.ctypes.data, hits.ctypes.data, cap, C.byref(nh))
            if rc == -7 and hits_cap is None:
                cap = int(nh.value) + 16
                continue
            _check(rc)
            return res, hits[: nh.value]

    def map_batch_compact(self, packed, lens, limit=1000, strand=2, mode=0, hits_cap=None, wide_cap=1024, out=None):
        """sbwt_map_batch_compact (host buffers) -> (compact records, hits of the multi-hit lists, wide records).
        out = (compact, hits, wide) arrays to fill (e.g. views of a PinnedBuffer)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = lens.size
        wpr = packed.shape[1] if packed.ndim == 2 else packed.size // max(n, 1)
        cap = int(hits_cap if hits_cap is not None else 1024)
        wcap = int(wide_cap)
        while True:
            if out is not None:
                rec, hits, wide = out
                cap, wcap = hits.size, wide.size
            else:
                rec = np.zeros(n, dtype=COMPACT_DTYPE)
                hits = np.zeros(max(cap, 1), dtype=HIT_DTYPE)
                wide = np.zeros(max(wcap, 1), dtype=WIDE_DTYPE)
            cnt = (C.c_uint64 * 2)()
            rc = lib().sbwt_map_batch_compact(self._h, packed.ctypes.data, lens.ctypes.data, n, wpr, limit, strand, mode,
                                              rec.ctypes.data, hits.ctypes.data, cap, wide.ctypes.data, wcap, C.byref(cnt))
            if rc == -7 and out is None and (hits_cap is None or cnt[1] > wcap):
                if hits_cap is None:
                    cap = max(cap, int(cnt[0]) + 16)
                wcap = max(wcap, int(cnt[1]) + 16)
                continue
            _check(rc)
            return rec[:n], hits[: cnt[0]], wide[: cnt[1]]

    def map_batch_device_compact(self, d_packed, d_lens, n_reads, wpr, d_out, d_hits, hits_cap, d_wide, wide_cap,
                                 limit=1000, strand=2, mode=0):
        """Device-pointer compact entry point -> (hit slots in use, wide slots in use)."""
        cnt = (C.c_uint64 * 2)()
        _check(lib().sbwt_map_batch_device_compact(self._h, d_packed, d_lens, n_reads, wpr, limit, strand, mode, d_out,
                                                   d_hits, hits_cap, d_wide, wide_cap, C.byref(cnt)))
        return int(cnt[0]), int(cnt[1])

    def map_batch_device(self, d_packed, d_lens, n_reads, wpr, d_results, d_hits, hits_cap, limit=1000, strand=2):
        """Device-pointer entry point (integers are raw device addresses)."""
        nh = C.c_uint64(0)
        _check(lib().sbwt_map_batch_device(self._h, d_packed, d_lens, n_reads, wpr, limit, strand, d_results, d_hits,
                                           hits_cap, C.byref(nh)))
        return nh.value

    def last_timing(self):
        ms = (C.c_float * 3)()
        _check(lib().sbwt_ctx_last_timing(self._h, C.byref(ms)))
        return list(ms)

    def last_stats(self):
        st = (C.c_uint64 * 4)()
        _check(lib().sbwt_ctx_last_stats(self._h, C.byref(st)))
        return list(st)

    def last_timing_ex(self):
        ms = (C.c_float * 5)()
        _check(lib().sbwt_ctx_last_timing_ex(self._h, C.byref(ms), 5))
        return list(ms)

    def last_stats_ex(self):
        st = (C.c_uint64 * 14)()
        _check(lib().sbwt_ctx_last_stats_ex(self._h, C.byref(st), 14))
        return list(st)

    def close(self):
        if self._h:
            lib().sbwt_ctx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
