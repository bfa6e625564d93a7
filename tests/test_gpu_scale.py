"""GPU: parity at a size the oracle still finishes in seconds, on the shapes of BASELINE.json's other configs:
36-, 50- and 100-bp reads (configs 2-4) and 18-30-nt reads over an injected repeat family with 2..1500 copies
(config 5: multi-hit locate, suppression above -m 1000), both strands, a fifth of the reads with one substitution.
The index is built on the GPU from the packed genome (sbwt_index_build_device); the oracle runs on the
reference-format tables the same build exports, so both sides see the same index."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def pack_codes(codes):
    n = len(codes)
    words = (n + 15) // 16
    buf = np.zeros(words * 16, dtype=np.uint64)
    buf[:n] = codes
    sh = (2 * np.arange(16, dtype=np.uint64))
    return (buf.reshape(words, 16) << sh).sum(axis=1).astype(np.uint32)


CHR_START = np.array([0, 1_200_000, 2_100_000], dtype=np.uint64)
NPOS_KEY = np.array([0, 400_000, 1_200_000, 2_100_000, 2_600_000, 3_000_000], dtype=np.uint64)
NPOS_VAL = np.array([1000, 51000, 500, 0, 77, 0], dtype=np.uint64)


def build_index(sb, codes, env, want_tables=False):
    """index of the packed genome, built on the GPU under the given environment (prefix-table knobs)"""
    hip = C.CDLL("libamdhip64.so.7")       # the runtime libsbwt_amd.so loaded (no torch here: a second HIP runtime in one
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]   # process that initialises after the first sees no GPU)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    host = np.concatenate([pack_codes(codes), np.zeros(4, np.uint32)])
    d_fwd = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_fwd), host.nbytes) == 0
    assert hip.hipMemcpy(d_fwd, host.ctypes.data, host.nbytes, 1) == 0
    os.environ.update(env)
    try:
        return sb.build_index_device(d_fwd.value, len(codes), CHR_START, NPOS_KEY, NPOS_VAL, interval=64, device=0,
                                     want_index=True, want_tables=want_tables)
    finally:
        for k in env:
            os.environ.pop(k)
        hip.hipFree(d_fwd)


@pytest.fixture(scope="module")
def world(oracle):
    import sbwt_amd as sb
    sb.device_count()                      # loads libsbwt_amd.so and, with it, the HIP runtime
    rng = np.random.default_rng(2024)
    g = 3_000_000
    codes = rng.integers(0, 4, g).astype(np.uint8)
    # repeat family: 40 distinct 30-mers with 2..1500 copies each, planted at random non-overlapping slots
    slots = rng.permutation(g // 64 - 2)[:6000] * 64 + 7
    fam = rng.integers(0, 4, (40, 30)).astype(np.uint8)
    copies = [2, 3, 5, 8, 13, 50, 200, 999, 1000, 1001, 1500] + [4] * 29
    k = 0
    for f, c in zip(fam, copies):
        for _ in range(c):
            codes[slots[k]: slots[k] + 30] = f
            k += 1
    chr_start, npos_key, npos_val = CHR_START, NPOS_KEY, NPOS_VAL
    # (an index of 2^22 suffixes or more gets the order-20 hash table by itself; stated for clarity)
    ix, tables = build_index(sb, codes, {"SBWT_HASH_ORDER": "20"}, want_tables=True)
    T = tables.contents
    L_ = oracle.lib()
    c4 = (C.c_uint64 * 4)(*T.c_function4)
    h = L_.sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions, T.occ[0],
                          T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt, T.location_table, T.n_loc,
                          chr_start.size, chr_start.ctypes.data, npos_key.size, npos_key.ctypes.data,
                          npos_val.ctypes.data)
    oix = oracle.OracleIndex(h, keep=(c4, chr_start, npos_key, npos_val))
    yield sb, ix, oix, codes, fam, rng
    oix.close()
    ix.close()
    sb.host_tables_free(tables)


def sample_reads(rng, codes, n, lens, mutate=0.2):
    g = len(codes)
    out = []
    for _ in range(n):
        L = int(rng.choice(lens))
        p = int(rng.integers(0, g - L))
        r = codes[p:p + L].copy()
        if rng.random() < 0.5:
            r = 3 - r[::-1]
        if rng.random() < mutate:
            q = int(rng.integers(0, L))
            r[q] = (r[q] + 1 + rng.integers(0, 3)) & 3
        out.append(bytes(b"ACGT"[c] for c in r))
    return out


def check(sb, ix, oix, reads, limit=1000, strand=2):
    packed, lens = sb.pack_reads(reads)
    ctx = sb.Context(ix, len(reads))
    res, hits = ctx.map_batch(packed, lens, limit=limit, strand=strand)
    stats = ctx.last_stats()
    ctx.close()
    lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand, nthreads=8)
    np.testing.assert_array_equal(res["sa_lo"], lo)
    np.testing.assert_array_equal(res["sa_hi"], hi)
    np.testing.assert_array_equal(res["nh"], nh)
    # ordered hit lists, all reads at once
    gi = np.repeat(res["first_hit"], np.maximum(nh, 0)) + (np.arange(int(nh.sum())) - np.repeat(np.cumsum(nh) - nh, nh))
    oi = np.repeat(first, np.maximum(nh, 0)) + (np.arange(int(nh.sum())) - np.repeat(np.cumsum(nh) - nh, nh))
    assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
    return res, nh, stats


@pytest.mark.parametrize("L", [36, 50, 100])
def test_fixed_length_reads(world, L):
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 20000, [L])
    res, nh, stats = check(sb, ix, oix, reads)
    assert (nh == 1).mean() > 0.7
    assert stats[0] > 0                                  # the kernel reports its index accesses


def test_small_rna_multi_hit_reads(world):
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 8000, list(range(18, 31)))
    for f in fam:                                        # reads inside the repeat family: 2..1500 hits
        for a, b in ((0, 30), (0, 18), (5, 27), (12, 30)):
            s = f[a:b]
            reads.append(bytes(b"ACGT"[c] for c in s))
            reads.append(bytes(b"ACGT"[c] for c in (3 - s[::-1])))
    res, nh, stats = check(sb, ix, oix, reads)
    assert nh.max() >= 999 and (res["flags"] & sb.RF_SUPPRESSED).any()
    for limit, strand in ((5, 2), (1000, 0), (1000, 1), (-1, 2)):
        check(sb, ix, oix, reads[-400:], limit=limit, strand=strand)


@pytest.mark.parametrize("knob", ["1", "2"])
def test_small_multi_hit_reads_several_per_wave_or_one_at_a_time(world, monkeypatch, knob):
    """k_multi_sort's wave teams locate reads of at most 8 / 64 rows several at a time (segments of the team's 256
    entries); SBWT_NO_SEGMENTS, read when the ctx is made, switches that off (1) or leaves only the 64-entry segments (2):
    the same records and lists, and the oracle's"""
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 3000, list(range(18, 31)))
    for f, c in zip(fam, [2, 3, 5, 8, 13, 50, 200, 999, 1000, 1001, 1500] + [4] * 29):
        for a, b in ((0, 30), (2, 24), (7, 30)):
            s = f[a:b]
            reads.append(bytes(b"ACGT"[c_] for c_ in s))
            reads.append(bytes(b"ACGT"[c_] for c_ in (3 - s[::-1])))
    res_a, nh, _ = check(sb, ix, oix, reads)
    assert ((nh >= 2) & (nh <= 8)).sum() > 100 and ((nh > 8) & (nh <= 64)).sum() >= 6
    monkeypatch.setenv("SBWT_NO_SEGMENTS", knob)
    res_b, nh_b, _ = check(sb, ix, oix, reads)
    np.testing.assert_array_equal(nh, nh_b)


@pytest.mark.parametrize("env,runs", [({"SBWT_HASH_ORDER": "0", "SBWT_KMER_ORDER": "12"}, True),
                                      ({"SBWT_HASH_ORDER": "0", "SBWT_KMER_ORDER": "14"}, True),
                                      ({"SBWT_HASH_ORDER": "0", "SBWT_KMER_ORDER": "15"}, True),
                                      ({"SBWT_HASH_ORDER": "16", "SBWT_HASH_TEXT": "0", "SBWT_KMER_ORDER": "12"}, True)])
def test_lockstep_rank_pass_on_an_index_without_slot_text(world, monkeypatch, env, runs):
    """k_rank (rank_pass.h): an index without the text-carrying hash table maps compact batches of 64-base records through
    prefix entry -> two-base steps -> sa_full -> text comparison in lockstep and leaves the rest pending: every field of
    every record (dead reads, multi-hit reads, short reads included) equals the oracle's, with the pass and without it
    (SBWT_NO_RANK_PASS); memoised prefixes of even and odd order"""
    sb, ix0, oix, codes, fam, rng = world
    ix, _ = build_index(sb, codes, env)
    try:
        assert ix.info().hash_text == 0
        for lens_ in ([12, 13, 20, 31, 36, 50, 51, 64], [30, 65, 80, 100, 127, 128]):   # records of 64 and of 128 bases
            reads = sample_reads(rng, codes, 12000, lens_)
            for f in fam[:12]:
                reads.append(bytes(b"ACGT"[c] for c in f))
            packed, lens = sb.pack_reads(reads)
            lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
            k = np.maximum(nh, 0)
            oi = np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            for off in (False, True):
                if off:
                    monkeypatch.setenv("SBWT_NO_RANK_PASS", "1")
                else:
                    monkeypatch.delenv("SBWT_NO_RANK_PASS", raising=False)
                ctx = sb.Context(ix, len(reads))
                # the 32-byte records of sbwt_map_batch take the same pass
                res, hits = ctx.map_batch(packed, lens)
                assert (ctx.last_timing_ex()[3] > 0) == (runs and not off)
                np.testing.assert_array_equal(res["nh"], nh)
                np.testing.assert_array_equal(res["sa_lo"], lo)
                np.testing.assert_array_equal(res["sa_hi"], hi)
                gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
                assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
                for mode in (0, sb.MAP_HITS_ONLY):
                    rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=mode)
                    t, st = ctx.last_timing_ex(), ctx.last_stats_ex()
                    assert (t[3] > 0) == (runs and not off)          # the fast pass ran / did not
                    if runs and not off:
                        assert 0 < st[7] < 0.5 * len(reads)           # ... and settled most reads itself
                    res, hits = sb.compact_expand(rec, lists, wide)
                    np.testing.assert_array_equal(res["nh"], nh)
                    np.testing.assert_array_equal(res["sa_lo"], lo)
                    np.testing.assert_array_equal(res["sa_hi"], hi)
                    gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
                    assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
                ctx.close()
    finally:
        ix.close()


def test_reads_of_257_to_512_rows_two_per_workgroup(oracle, monkeypatch):
    """k_multi_sort's workgroup teams locate reads of 257..512 rows two at a time (segments of 512 entries): a genome with
    repeat families of 256, 257, 300, 400, 512, 513 and 700 copies, reads inside them on both strands, against the oracle;
    with SBWT_NO_SEGMENTS the same reads one at a time"""
    import sbwt_amd as sb
    sb.device_count()
    rng = np.random.default_rng(512)
    g = 3_000_000
    codes = rng.integers(0, 4, g).astype(np.uint8)
    copies = [256, 257, 300, 300, 400, 400, 512, 512, 513, 700]
    slots = rng.permutation(g // 64 - 2)[:sum(copies)] * 64 + 5
    fam = rng.integers(0, 4, (len(copies), 32)).astype(np.uint8)
    k = 0
    for f, c in zip(fam, copies):
        for _ in range(c):
            codes[slots[k]: slots[k] + 32] = f
            k += 1
    ix, tables = build_index(sb, codes, {"SBWT_HASH_ORDER": "20"}, want_tables=True)
    T = tables.contents
    c4 = (C.c_uint64 * 4)(*T.c_function4)
    h = oracle.lib().sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions, T.occ[0],
                                    T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt, T.location_table, T.n_loc,
                                    CHR_START.size, CHR_START.ctypes.data, NPOS_KEY.size, NPOS_KEY.ctypes.data, NPOS_VAL.ctypes.data)
    oix = oracle.OracleIndex(h, keep=(c4, CHR_START, NPOS_KEY, NPOS_VAL))
    try:
        reads = sample_reads(rng, codes, 500, [30, 40])
        for f in fam:
            for a, b in ((0, 32), (0, 24), (3, 30), (8, 32), (1, 25)):
                s_ = f[a:b]
                reads.append(bytes(b"ACGT"[c_] for c_ in s_))
                reads.append(bytes(b"ACGT"[c_] for c_ in (3 - s_[::-1])))
        res, nh, _ = check(sb, ix, oix, reads)
        assert ((nh > 256) & (nh <= 512)).sum() >= 60 and (nh > 512).sum() >= 10
        for lim in (300, 1000):
            check(sb, ix, oix, reads[500:], limit=lim)
        monkeypatch.setenv("SBWT_NO_SEGMENTS", "1")
        res_b, nh_b, _ = check(sb, ix, oix, reads)
        np.testing.assert_array_equal(nh, nh_b)
    finally:
        oix.close()
        ix.close()
        sb.host_tables_free(tables)


def test_hits_only_fast_pass_settles_most_reads(world):
    """SBWT_MAP_HITS_ONLY on an index with text in its hash slots: k_settle answers the reads its buckets settle with
    one access each and leaves the rest to the general kernels; hits, NH and flags equal the oracle's for every
    read, and SA intervals are reported only by the reads that took the general path"""
    sb, ix, oix, codes, fam, rng = world
    assert ix.info().hash_order == 20
    reads = sample_reads(rng, codes, 30000, [32, 36, 50, 51, 52, 53, 60])
    for f in fam[:8]:
        reads.append(bytes(b"ACGT"[c] for c in f))
    packed, lens = sb.pack_reads(reads)
    ctx = sb.Context(ix, len(reads))
    lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
    for limit, strand in ((1000, 2), (0, 2), (1000, 0), (1000, 1)):
        if (limit, strand) != (1000, 2):
            lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand, nthreads=8)
        rec, lists, wide = ctx.map_batch_compact(packed, lens, limit=limit, strand=strand, mode=sb.MAP_HITS_ONLY)
        meta = (rec["w1"] >> np.uint64(56)).astype(np.uint32)
        no_sa = (meta & sb.CR_NO_SA) != 0
        L = lens
        assert no_sa[(L > 20) & (L <= 52)].mean() > 0.8 and not no_sa[L > 52].any()
        res, hits = sb.compact_expand(rec, lists, wide)
        np.testing.assert_array_equal(res["nh"], nh)
        np.testing.assert_array_equal(res["sa_lo"][~no_sa], lo[~no_sa])
        np.testing.assert_array_equal(res["sa_hi"][~no_sa], hi[~no_sa])
        k = np.maximum(nh, 0)
        gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
        oi = np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
        assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
        t = ctx.last_timing_ex()
        assert t[3] > 0 and t[4] > 0                      # both passes ran
    ctx.close()


@pytest.mark.parametrize("lens_", [[21, 30, 40, 50, 52], [70, 100, 128], [150, 181, 192]])
def test_a_miss_in_the_complete_hash_table_settles_the_read(world, lens_):
    """hash_tab holds EVERY 20-base prefix whose search interval is not empty (located, or as a marker), so a prefix
    that is in neither bucket while one has a free slot occurs nowhere: the hits-only fast passes (k_settle for records
    of 64 bases, k_direct beyond) report such a read as unmapped with one access.  Reads with a substitution in their
    first 20 bases and random reads: same NH and hits as the oracle, and most of them counted in stats[10]"""
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 4000, lens_, mutate=0.0)
    n_exact = len(reads)
    for r in sample_reads(rng, codes, 4000, lens_, mutate=0.0):
        q = int(rng.integers(0, 20))
        b = bytearray(r)
        b[q] = b"ACGT"[(b"ACGT".index(b[q]) + 1 + int(rng.integers(0, 3))) & 3]
        reads.append(bytes(b))
    for _ in range(2000):
        reads.append(bytes(b"ACGT"[c] for c in rng.integers(0, 4, int(rng.choice(lens_)))))
    for f in fam[:8]:                                    # multi-row prefixes: markers, not misses
        reads.append(bytes(b"ACGT"[c] for c in f) + b"ACGTACGTACGTACGTACGTAC"[:max(0, min(lens_) - 30)])
    packed, lens = sb.pack_reads(reads)
    ctx = sb.Context(ix, len(reads))
    lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
    rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb.MAP_HITS_ONLY)
    st = ctx.last_stats_ex()
    res, hits = sb.compact_expand(rec, lists, wide)
    np.testing.assert_array_equal(res["nh"], nh)
    k = np.maximum(nh, 0)
    gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
    oi = np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
    assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
    unmapped = int((nh[n_exact:n_exact + 6000] == 0).sum())
    assert unmapped > 5900
    assert st[10] > 0.85 * unmapped and st[10] <= int((nh == 0).sum())   # (the rest found both buckets full)
    # the full records (SA interval of every read) are the general kernels' business: nothing is settled by a miss
    rec2, lists2, wide2 = ctx.map_batch_compact(packed, lens)
    assert ctx.last_stats_ex()[10] == 0
    res2, hits2 = sb.compact_expand(rec2, lists2, wide2)
    np.testing.assert_array_equal(res2["nh"], nh)
    np.testing.assert_array_equal(res2["sa_lo"], lo)
    np.testing.assert_array_equal(res2["sa_hi"], hi)
    ctx.close()


@pytest.mark.parametrize("env", [{"SBWT_HASH_ORDER": "16"}, {"SBWT_HASH_ORDER": "16", "SBWT_HASH_GROUP": "2"},
                                 {"SBWT_HASH_ORDER": "17", "SBWT_HASH_GROUP": "1"}, {"SBWT_HASH_ORDER": "16", "SBWT_NO_HASH_COMPLETE": "1"}])
def test_crowded_hash_table_groups_of_buckets(world, env):
    """an order-16 table over this genome is as crowded as the order-20 one of a human genome (1.6 slots per suffix,
    where the order-20 table of the other tests is nearly empty: its tag needs 2^25 buckets): prefixes overflow into
    the other buckets of their group (eight, four 128-byte lines; SBWT_HASH_GROUP lowers it) and some misses meet full
    buckets.  Same results as the oracle whatever the group; a miss settles reads only on a complete table"""
    sb, ix0, oix, codes, fam, rng = world
    ix, _ = build_index(sb, codes, env)
    H = int(env["SBWT_HASH_ORDER"])
    try:
        assert ix.info().hash_order == H
        reads = sample_reads(rng, codes, 6000, [30, 36, 48, 80, 100], mutate=0.0)
        for r in sample_reads(rng, codes, 6000, [30, 36, 48, 80, 100], mutate=0.0):
            q = int(rng.integers(0, H))
            b = bytearray(r)
            b[q] = b"ACGT"[(b"ACGT".index(b[q]) + 1 + int(rng.integers(0, 3))) & 3]
            reads.append(bytes(b))
        for f in fam[:12]:
            reads.append(bytes(b"ACGT"[c] for c in f))
        lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
        for lens_sel in (lambda L: L <= 48, lambda L: L > 0):     # records of 64 bases (k_settle), then of 128 (k_direct)
            sel = [r for r in reads if lens_sel(len(r))]
            idx = np.array([i for i, r in enumerate(reads) if lens_sel(len(r))])
            packed, lens = sb.pack_reads(sel)
            ctx = sb.Context(ix, len(sel))
            rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb.MAP_HITS_ONLY)
            st = ctx.last_stats_ex()
            res, hits = sb.compact_expand(rec, lists, wide)
            np.testing.assert_array_equal(res["nh"], nh[idx])
            k = np.maximum(nh[idx], 0)
            gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            oi = np.repeat(first[idx], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
            unmapped = int((nh[idx] == 0).sum())
            if "SBWT_NO_HASH_COMPLETE" in env:
                assert st[10] == 0
            else:
                assert 0.5 * unmapped < st[10] <= unmapped
            assert st[5] > 0.8 * int((nh[idx] == 1).sum())        # the located prefixes are found, wherever they sit
            # full records: the SA interval of every read
            rec2, lists2, wide2 = ctx.map_batch_compact(packed, lens)
            res2, hits2 = sb.compact_expand(rec2, lists2, wide2)
            np.testing.assert_array_equal(res2["nh"], nh[idx])
            np.testing.assert_array_equal(res2["sa_lo"], lo[idx])
            np.testing.assert_array_equal(res2["sa_hi"], hi[idx])
            ctx.close()
    finally:
        ix.close()


def every_prefix_check(sb, ix, H, monkeypatch):
    """EVERY string of H bases as a read of H bases, hits only: the fast pass (located prefixes, markers, misses that
    prove absence, full groups) against the general kernels alone -- same NH, flags and hits for all 4^H of them"""
    N = 4 ** H
    packed = np.zeros((N, 4), dtype=np.uint32)
    packed[:, 0] = np.arange(N, dtype=np.uint32)               # base t of read v = (v >> 2t) & 3
    lens = np.full(N, H, dtype=np.uint32)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("SBWT_NO_SETTLE", "1")
        else:
            monkeypatch.delenv("SBWT_NO_SETTLE", raising=False)
        ctx = sb.Context(ix, N)
        rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb.MAP_HITS_ONLY)
        st, t = ctx.last_stats_ex(), ctx.last_timing_ex()
        ctx.close()
        res, hits = sb.compact_expand(rec, lists, wide)
        del rec
        outs.append((res, hits, st, t))
    monkeypatch.delenv("SBWT_NO_SETTLE", raising=False)
    (ra, ha, sa, ta), (rb, hb, stb, tb) = outs
    assert ta[3] > 0 and tb[3] == 0                            # with the fast pass; without
    np.testing.assert_array_equal(ra["nh"], rb["nh"])
    np.testing.assert_array_equal(ra["flags"], rb["flags"])
    k = np.maximum(ra["nh"], 0).astype(np.int64)
    within = np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k)
    ga = np.repeat(ra["first_hit"].astype(np.int64), k) + within
    gb = np.repeat(rb["first_hit"].astype(np.int64), k) + within
    assert np.array_equal(ha[ga], hb[gb])
    return ra, sa


def test_every_possible_prefix_fast_pass_equals_general_kernels(world, monkeypatch):
    """an order-13 hash table over this genome (6 x 10^6 rows, 4^13 = 6.7 x 10^7 possible prefixes, one in twelve
    occurs): all 4^13 reads.  What a miss in the complete table proves is thereby checked for every prefix there is"""
    sb, ix0, oix, codes, fam, rng = world
    ix, _ = build_index(sb, codes, {"SBWT_HASH_ORDER": "13"})
    try:
        assert ix.info().hash_order == 13
        res, st = every_prefix_check(sb, ix, 13, monkeypatch)
        absent = int(((res["nh"] == 0) & (res["flags"] == 0)).sum())
        assert 0.9 * absent < st[10] <= absent                  # nearly every read that maps nowhere was settled by a miss
        assert int((res["nh"] > 0).sum()) > 4_000_000
    finally:
        ix.close()


def test_reads_with_one_interval_share_a_located_list(world, monkeypatch):
    """small-RNA batches hold the same molecule many times: reads with the same interval and length are located once
    and the records of the others point at that list.  Same NH and lists as the oracle, which locates every read;
    same again with the grouping switched off."""
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 2000, list(range(18, 31)))
    members = []
    for f in fam:
        for a, b in ((0, 30), (0, 18), (0, 19), (5, 27), (12, 30)):   # (0, 18) and (0, 19): intervals may share lo, not hi
            s = f[a:b]
            members.append(bytes(b"ACGT"[c] for c in s))
            members.append(bytes(b"ACGT"[c] for c in (3 - s[::-1])))
    for rep in range(6):
        reads += [members[i] for i in rng.permutation(len(members))]
    reads = [reads[i] for i in rng.permutation(len(reads))]
    packed, lens = sb.pack_reads(reads)
    lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
    k = np.maximum(nh, 0)
    oi = (np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))).astype(np.int64)

    def run(ctx):
        out = []
        res, hits = ctx.map_batch(packed, lens)
        out.append((res, hits, ctx.last_stats_ex()))
        for mode in (0, sb.MAP_HITS_ONLY):
            rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=mode)
            res, hits = sb.compact_expand(rec, lists, wide)
            out.append((res, hits, ctx.last_stats_ex()))
        for res, hits, stats in out:
            np.testing.assert_array_equal(res["nh"], nh)
            gi = (np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))).astype(np.int64)
            assert np.array_equal(hits[gi], ohits[oi])
        return [s for _, _, s in out]

    ctx = sb.Context(ix, len(reads))
    stats = run(ctx)
    ctx.close()
    multi = int((nh > 1).sum())
    for s in stats:
        assert s[8] > multi // 2 and s[8] < s[1]          # most multi-hit reads share a list; the leaders own theirs
    monkeypatch.setenv("SBWT_NO_SHARED_LISTS", "1")
    ctx = sb.Context(ix, len(reads))
    stats = run(ctx)
    ctx.close()
    assert all(s[8] == 0 for s in stats)


@pytest.mark.parametrize("lens_", [[32, 36, 50, 51, 52, 53, 60, 64], [65, 80, 100, 101, 120, 128], [129, 140, 150, 161, 162, 163, 170, 192]])
def test_two_access_fast_pass_reports_intervals(world, monkeypatch, lens_):
    """compact records WITH SA intervals, and records of up to 128 bases: k_direct answers the reads whose located
    prefix and PosLine settle them (bucket + one line each) and leaves the rest to the general kernels; every record
    equals the oracle's, interval included -- and equals what the general kernels alone produce"""
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 20000, lens_, mutate=0.3)
    for f in fam[:8]:
        reads.append(bytes(b"ACGT"[c] for c in f) * 2)      # 60 bases, multi-hit prefix
    packed, lens = sb.pack_reads(reads)
    for limit, strand, mode in ((1000, 2, 0), (0, 2, 0), (1000, 0, 0), (1000, 1, 0), (1000, 2, sb.MAP_HITS_ONLY)):
        lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand, nthreads=8)
        outs = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("SBWT_NO_DIRECT", "1")
            else:
                monkeypatch.delenv("SBWT_NO_DIRECT", raising=False)
            ctx = sb.Context(ix, len(reads))
            rec, lists, wide = ctx.map_batch_compact(packed, lens, limit=limit, strand=strand, mode=mode)
            t, st = ctx.last_timing_ex(), ctx.last_stats_ex()
            ctx.close()
            ran_direct = not (mode == sb.MAP_HITS_ONLY and max(lens_) <= 64) and not off   # (hits only, short records: k_settle)
            if ran_direct:
                assert t[3] > 0                                # the fast pass ran ...
                if max(lens_) <= 128:
                    assert st[7] < 0.5 * len(reads)            # ... and settled most reads (beyond 162 bases it cannot)
            if off and mode == 0:
                assert t[3] == 0
            meta = (rec["w1"] >> np.uint64(56)).astype(np.uint32)
            no_sa = (meta & sb.CR_NO_SA) != 0
            if mode == 0:
                assert not no_sa.any()
            res, hits = sb.compact_expand(rec, lists, wide)
            np.testing.assert_array_equal(res["nh"], nh)
            np.testing.assert_array_equal(res["sa_lo"][~no_sa], lo[~no_sa])
            np.testing.assert_array_equal(res["sa_hi"][~no_sa], hi[~no_sa])
            k = np.maximum(nh, 0)
            gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            oi = np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
            outs.append((res, no_sa))
        np.testing.assert_array_equal(outs[0][0]["flags"], outs[1][0]["flags"])
    monkeypatch.delenv("SBWT_NO_DIRECT", raising=False)


@pytest.mark.parametrize("lens_", [[32, 36, 50, 64], [65, 100, 128], [129, 150, 192]])
def test_32_byte_records_take_the_two_access_fast_pass(world, monkeypatch, lens_):
    """sbwt_map_batch / sbwt_map_batch_device (32-byte records, the unique hit at hits[read index]) go through k_direct
    too: the settled reads' records and hits are written in place, the pending reads' by k_scatter_pending_plain.
    Every field equals the oracle's and what the general kernels alone produce, on one call and on sub-batches"""
    sb, ix, oix, codes, fam, rng = world
    reads = sample_reads(rng, codes, 12000, lens_, mutate=0.3)
    for f in fam[:10]:
        reads.append(bytes(b"ACGT"[c] for c in f) * 2)      # 60 bases, multi-hit prefix
        reads.append(bytes(b"ACGT"[c] for c in f))          # 30 bases, 2..1500 hits
    packed, lens = sb.pack_reads(reads)
    for limit, strand, subbatch in ((1000, 2, None), (0, 2, None), (1000, 0, None), (5, 1, "4096")):
        lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand, nthreads=8)
        outs = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("SBWT_NO_DIRECT", "1")
            else:
                monkeypatch.delenv("SBWT_NO_DIRECT", raising=False)
            if subbatch:
                monkeypatch.setenv("SBWT_SUBBATCH", subbatch)
            ctx = sb.Context(ix, len(reads))
            res, hits = ctx.map_batch(packed, lens, limit=limit, strand=strand)
            t, st = ctx.last_timing_ex(), ctx.last_stats_ex()
            ctx.close()
            monkeypatch.delenv("SBWT_SUBBATCH", raising=False)
            assert (t[3] > 0) == (not off)                     # the fast pass ran, or was switched off
            if not off and max(lens_) <= 128:
                assert st[7] < 0.5 * len(reads)                # ... and settled most reads
            np.testing.assert_array_equal(res["sa_lo"], lo)
            np.testing.assert_array_equal(res["sa_hi"], hi)
            np.testing.assert_array_equal(res["nh"], nh)
            uniq = nh == 1
            k = np.maximum(nh, 0)
            gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            oi = np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)])
            outs.append((res, uniq))
        np.testing.assert_array_equal(outs[0][0]["flags"], outs[1][0]["flags"])
    monkeypatch.delenv("SBWT_NO_DIRECT", raising=False)


@pytest.mark.parametrize("layout", ["hits", "compact", "plain"])
def test_pending_pass_launched_without_waiting_for_the_fast_pass(world, monkeypatch, layout):
    """Calls in a row on ONE ctx.  The first sizes the buffers of the pending reads on the host; from the second on gather,
    search kernel and scatter are launched with the count still in device memory (last_stats_ex [13]) and the locate
    kernels only when the batch turns out to hold multi-hit reads; a batch with more pending reads than the buffers hold
    is done over the first way.  Every call equals the oracle, and the path that always waits (the knob)."""
    sb, ix, oix, codes, fam, rng = world
    fam_reads = [bytes(b"ACGT"[c] for c in f[a:b]) for f in fam[:12] for a, b in ((0, 30), (3, 28))]
    batches = [sample_reads(rng, codes, 3000, [32, 40, 50]) + fam_reads,            # sizes the buffers
               sample_reads(rng, codes, 3000, [32, 40, 50]),                        # no multi-hit read left pending ...
               sample_reads(rng, codes, 2500, [32, 40, 50]) + fam_reads * 2,        # ... multi-hit reads: the second phase
               sample_reads(rng, codes, 3000, [60, 64]) + fam_reads,                # everything pending (hits only): outgrows the buffers
               sample_reads(rng, codes, 2000, [36, 50], mutate=0.0)]
    outs = {}
    for knob in (False, True):
        if knob:
            monkeypatch.setenv("SBWT_NO_SPECULATIVE_PENDING", "1")
        ctx = sb.Context(ix, max(len(b) for b in batches))
        for bi, reads in enumerate(batches):
            packed, lens = sb.pack_reads(reads)
            if layout == "plain":
                res, hits = ctx.map_batch(packed, lens)
            else:
                rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb.MAP_HITS_ONLY if layout == "hits" else 0)
                res, hits = sb.compact_expand(rec, lists, wide)
            st = ctx.last_stats_ex()
            lo, hi, nh, first, ohits = oix.map_batch(reads, nthreads=8)
            np.testing.assert_array_equal(res["nh"], nh)
            k = np.maximum(nh, 0)
            gi = np.repeat(res["first_hit"], k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            oi = np.repeat(first, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))
            assert np.array_equal(hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)]), (knob, bi)
            if layout != "hits":
                np.testing.assert_array_equal(res["sa_lo"], lo)
                np.testing.assert_array_equal(res["sa_hi"], hi)
            outs[(knob, bi)] = (int(st[7]), int(st[13]), int(st[1]))
        ctx.close()
    monkeypatch.delenv("SBWT_NO_SPECULATIVE_PENDING", raising=False)
    # (the first call, and one that outgrows the buffers, wait -- unless the wrapper had to call twice for room in hits[])
    assert outs[(False, 1)][1] == 1 and outs[(False, 2)][1] == 1 and outs[(False, 4)][1] == 1
    assert outs[(False, 2)][2] > 0                           # multi-hit reads among the pending ones of the third call
    assert all(outs[(True, b)][1] == 0 for b in range(5))    # the knob: always the waiting path
    assert all(outs[(True, b)][0] == outs[(False, b)][0] for b in range(5))   # the same reads left pending either way


def test_roofline_probes_run(world):
    """sbwt_probe_random_access / sbwt_probe_gather_stream: measurement entry points; here only that they run on a small
    index and report a rate"""
    sb, ix, oix, codes, fam, rng = world
    rate, nbytes = ix.probe_random_access(chains_per_lane=2, chain_len=8, workgroups_per_cu=2)
    assert rate > 0 and nbytes >= 128
    assert ix.probe_gather_stream(200_000, 2) > 0
    with pytest.raises(sb.SbwtError):
        ix.probe_gather_stream(0)
