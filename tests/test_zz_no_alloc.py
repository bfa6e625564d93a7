"""GPU: sbwt_map_batch_device_static (the entry point that allocates nothing: its kernels' scratch is a file-scope
device array) against sbwt_map_batch_device on the same device buffers, bit for bit, on two calls in a row, the
second with a smaller input; then its size limit and its capacity report.  Collected last (file name)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COMP = str.maketrans("ACGT", "TGCA")


class Dev:
    """device buffers through the HIP runtime the library itself uses"""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so.7")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def alloc(self, nbytes, fill=0xA5):
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        assert self.hip.hipMemset(p, fill, max(nbytes, 16)) == 0
        self.bufs.append(p)
        return p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        assert self.hip.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def download(self, ptr, dtype, count):
        out = np.zeros(count, dtype=dtype)
        assert self.hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0
        return out

    def close(self):
        for p in self.bufs:
            self.hip.hipFree(p)
        self.bufs = []


def genome_text(fa):
    return "".join(l.strip() for l in open(fa) if not l.startswith(">")).upper()


def make_reads(seq, n, seed):
    rng = np.random.default_rng(seed)
    reads = []
    while len(reads) < n:
        L = int(rng.integers(12, 90))
        p = int(rng.integers(0, len(seq) - L))
        s = seq[p:p + L]
        if "N" in s:
            continue
        if rng.random() < 0.5:
            s = s.translate(COMP)[::-1]
        if rng.random() < 0.3:
            k = int(rng.integers(0, L))
            s = s[:k] + "ACGT"[int(rng.integers(0, 4))] + s[k + 1:]
        reads.append(s)
    return reads


def hit_lists(res, hits):
    nh = np.maximum(res["nh"], 0)
    idx = np.repeat(res["first_hit"], nh) + (np.arange(int(nh.sum())) - np.repeat(np.cumsum(nh) - nh, nh))
    return hits[idx.astype(np.int64)]


@pytest.mark.parametrize("interval,orders", [(64, None), (256, None), (64, "14:19")])
def test_static_entry_point_equals_ctx_entry_point(oracle, golden_dir, tmp_path, interval, orders):
    import sbwt_amd as sb
    assert sb.device_count() >= 1
    prefix = str(tmp_path / "g2")
    oracle.build_index(os.path.join(golden_dir, "g2.fa"), prefix, interval=interval)   # 256: the depth-first kernel and its stacks
    if orders:
        # a text-carrying hash table forced onto the small index: the ctx path then runs k_direct + the gathered pending
        # batch and shares lists between reads of one interval; the static path (general kernels, no grouping) beside it
        os.environ["SBWT_KMER_ORDER"], os.environ["SBWT_HASH_ORDER"] = orders.split(":")
    try:
        ix = sb.Index.load(prefix)
    finally:
        os.environ.pop("SBWT_KMER_ORDER", None)
        os.environ.pop("SBWT_HASH_ORDER", None)
    assert (ix.info().hash_text == 1) == bool(orders)
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    rep = seq[1000:1300]
    dev = Dev()
    # two calls in a row on the shared scratch, the second smaller: what the first left behind must not matter
    for n, seed in ((6000, 1), (700, 2)):
        reads = make_reads(seq, n, seed) + [rep[a:a + 30] for a in range(0, 250, 7)] * (2 if orders else 1)   # (twice: list sharing)
        packed, lens = sb.pack_reads(reads)
        n_reads, wpr = len(reads), packed.shape[1]
        d_packed, d_lens = dev.upload(packed), dev.upload(lens)
        cap = 4 * n_reads
        out = []
        ctx = sb.Context(ix, n_reads)
        for which in ("ctx", "static"):
            d_res, d_hits = dev.alloc(n_reads * 32), dev.alloc(cap * 16)
            if which == "ctx":
                used = ctx.map_batch_device(d_packed, d_lens, n_reads, wpr, d_res, d_hits, cap)
                ctx_fast = ctx.last_timing_ex()[3]
            else:
                used = ix.map_batch_device_static(d_packed, d_lens, n_reads, wpr, d_res, d_hits, cap)
            res = dev.download(d_res, sb.api.RESULT_DTYPE, n_reads)
            hits = dev.download(d_hits, sb.api.HIT_DTYPE, cap)
            out.append((used, res, hits))
        ctx.close()
        (ua, ra, ha), (ub, rb, hb) = out
        assert ua == ub and ua > n_reads          # (followers of a shared list keep their reserved slots: same count)
        if orders:
            assert ctx_fast > 0                    # the ctx call went through the fast pass
        for f in ("sa_lo", "sa_hi", "nh", "flags"):
            np.testing.assert_array_equal(ra[f], rb[f])
        assert (ra["nh"] > 1).sum() >= 30 and (ra["nh"] == 1).sum() > n // 3
        # per-read hit lists in emission order (where a list lies in hits[] is not part of the contract: the ctx path
        # shares lists between reads of one interval and lays them out in the order its pending pass met them)
        assert np.array_equal(hit_lists(ra, ha), hit_lists(rb, hb))
        # and against the oracle, so that "equal" cannot mean "equally wrong"
        oix = oracle.OracleIndex.load(prefix)
        lo, hi, nh, first, ohits = oix.map_batch(reads)
        np.testing.assert_array_equal(rb["sa_lo"], lo)
        np.testing.assert_array_equal(rb["nh"], nh)
        assert np.array_equal(hit_lists(rb, hb), ohits)
        oix.close()
        # the lists do not fit: same verdict and the size a retry needs
        with pytest.raises(sb.SbwtError) as e:
            ix.map_batch_device_static(d_packed, d_lens, n_reads, wpr, d_res, d_hits, n_reads)
        assert e.value.code == -7
        dev.close()
    # beyond the static maximum: refused with a clear message, nothing launched
    with pytest.raises(sb.SbwtError) as e:
        ix.map_batch_device_static(1, 1, (1 << 20) + 1, 4, 1, 1, 1 << 21)
    assert e.value.code == -4 and "1048576" in str(e.value)
    assert ix.map_batch_device_static(None, None, 0, 4, None, None, 0) == 0
    ix.close()
