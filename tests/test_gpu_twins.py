"""GPU: twin slots of hash_tab (device_layout.h): a 20-mer that lies on exactly two located rows has two slots, and
the fast passes settle a read on the twin whose text it goes on with.  A genome of duplicated segments whose copies
continue differently; reads that leave the duplicate (one twin matches), stay inside it (both: two hits, general
kernels), carry a substitution behind the prefix (neither, or one), both strands -- against the oracle through the
hits-only, the interval-reporting compact and the plain layout, and against an index built without twins."""
import numpy as np
import pytest

from test_gpu_scale import CHR_START, NPOS_KEY, NPOS_VAL, build_index   # noqa: F401
import ctypes as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def twin_world(oracle):
    import sbwt_amd as sb
    sb.device_count()
    rng = np.random.default_rng(77)
    g = 3_000_000
    codes = rng.integers(0, 4, g).astype(np.uint8)
    # 4,000 segments of 24..90 bases, each planted twice (a quarter of them: the second copy reverse-complemented)
    slots = rng.permutation(g // 128 - 2)[:8000] * 128 + 11
    segs = []
    for k in range(4000):
        n = int(rng.integers(24, 91))
        s = rng.integers(0, 4, n).astype(np.uint8)
        a, b = int(slots[2 * k]), int(slots[2 * k + 1])
        codes[a:a + n] = s
        codes[b:b + n] = (3 - s[::-1]) if k % 4 == 0 else s
        segs.append((a, b, n))
    ix, tables = build_index(sb, codes, {"SBWT_HASH_ORDER": "20"}, want_tables=True)
    ix_plain, _ = build_index(sb, codes, {"SBWT_HASH_ORDER": "20", "SBWT_NO_TWINS": "1"})
    T = tables.contents
    L_ = oracle.lib()
    c4 = (C.c_uint64 * 4)(*T.c_function4)
    h = L_.sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions, T.occ[0],
                          T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt, T.location_table, T.n_loc,
                          CHR_START.size, CHR_START.ctypes.data, NPOS_KEY.size, NPOS_KEY.ctypes.data, NPOS_VAL.ctypes.data)
    oix = oracle.OracleIndex(h, keep=(c4, CHR_START, NPOS_KEY, NPOS_VAL))
    yield sb, ix, ix_plain, oix, codes, segs, rng
    oix.close()
    ix.close()
    ix_plain.close()
    sb.host_tables_free(tables)


def twin_reads(rng, codes, segs, lens):
    """reads around the duplicated segments: the search runs right to left, so a read that ENDS inside a copy and
    starts in front of it has its first 20 searched bases on both copies and leaves the duplicate further on"""
    out = []
    g = len(codes)
    for (a, b, n) in segs:
        for start in (a, b):
            L = int(rng.choice(lens))
            kind = int(rng.integers(0, 4))
            if kind == 0:                       # inside the copy, when it fits: both twins match
                L = min(L, n)
                p = start + int(rng.integers(0, n - L + 1))
            elif kind == 1:                     # ends inside the copy (20 bases of it at least), begins in front of it
                end = start + int(rng.integers(20, max(20, min(n, L - 1)) + 1))
                p = end - L
            else:                               # begins inside the copy, ends behind it (the other strand's search sees it as kind 1)
                p = start + int(rng.integers(min(max(0, n - L + 1), n - 20), n - 19))
            p = max(0, min(p, g - L))
            r = codes[p:p + L].copy()
            if rng.random() < 0.5:
                r = 3 - r[::-1]
            if rng.random() < 0.3:
                q = int(rng.integers(0, L))
                r[q] = (r[q] + 1 + rng.integers(0, 3)) & 3
            out.append(bytes(b"ACGT"[c] for c in r))
    return out


def lists_of(res, hits):
    k = np.maximum(res["nh"], 0)
    gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
    return hits[gi.astype(np.int64)]


@pytest.mark.parametrize("lens_", [[20, 21, 24, 30, 36, 50, 52], [53, 64, 70, 100, 128], [129, 150, 180, 192]])
def test_twin_slots_settle_reads_of_prefixes_on_two_rows(twin_world, lens_):
    sb, ix, ix_plain, oix, codes, segs, rng = twin_world
    reads = twin_reads(rng, codes, segs, lens_)
    packed, lens = sb.pack_reads(reads)
    lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
    assert (nh == 2).sum() > 200 and (nh == 1).sum() > 1500 and (nh == 0).sum() > 200
    want = ohits[(np.repeat(first, np.maximum(nh, 0)) + (np.arange(int(np.maximum(nh, 0).sum())) -
                  np.repeat(np.cumsum(np.maximum(nh, 0)) - np.maximum(nh, 0), np.maximum(nh, 0)))).astype(np.int64)]
    pending = {}
    for name, index in (("twins", ix), ("no twins", ix_plain)):
        ctx = sb.Context(index, len(reads))
        # hits only: NH, hits, flags
        rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb.MAP_HITS_ONLY)
        st = ctx.last_stats_ex()
        pending[name] = int(st[7])
        res, hits = sb.compact_expand(rec, lists, wide)
        np.testing.assert_array_equal(res["nh"], nh)
        assert np.array_equal(lists_of(res, hits), want)
        no_sa = ((rec["w1"] >> np.uint64(56)).astype(np.uint32) & sb.CR_NO_SA) != 0
        np.testing.assert_array_equal(res["sa_lo"][~no_sa], lo[~no_sa])
        np.testing.assert_array_equal(res["sa_hi"][~no_sa], hi[~no_sa])
        # with SA intervals, compact and plain: every field
        rec, lists, wide = ctx.map_batch_compact(packed, lens)
        res, hits = sb.compact_expand(rec, lists, wide)
        for got in (res, ):
            np.testing.assert_array_equal(got["nh"], nh)
            np.testing.assert_array_equal(got["sa_lo"], lo)
            np.testing.assert_array_equal(got["sa_hi"], hi)
        assert np.array_equal(lists_of(res, hits), want)
        res, hits = ctx.map_batch(packed, lens)
        np.testing.assert_array_equal(res["nh"], nh)
        np.testing.assert_array_equal(res["sa_lo"], lo)
        np.testing.assert_array_equal(res["sa_hi"], hi)
        assert np.array_equal(lists_of(res, hits), want)
        for limit, strand in ((0, 2), (1000, 0), (1000, 1)):
            l2, h2, n2, f2, oh2 = oix.map_batch(reads[:3000], limit=limit, strand=strand, nthreads=8)
            rec, lists, wide = ctx.map_batch_compact(packed[:3000], lens[:3000], limit=limit, strand=strand, mode=sb.MAP_HITS_ONLY)
            res, hits = sb.compact_expand(rec, lists, wide)
            np.testing.assert_array_equal(res["nh"], n2)
            k = np.maximum(n2, 0)
            oi = np.repeat(f2, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            assert np.array_equal(lists_of(res, hits), oh2[oi.astype(np.int64)])
        ctx.close()
    # the twins took work off the general kernels: the reads that leave their duplicate are settled by the fast pass
    assert pending["twins"] < 0.8 * pending["no twins"], pending

