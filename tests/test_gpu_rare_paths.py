"""GPU: code that only unusual indexes reach, each against the oracle.

* k_search_sm<false, *> / k_multi_sort without LDS tables: more than 1,536 chromosome + N-gap keys (2,000 contigs),
  as a GRCh38 build with alt/decoy contigs has; also forced on the golden genome through SBWT_DEBUG_LDS_KEYS.
* the R_DIRECT bypass of the result ring: chr_id above 65,535 (70,000 contigs) and, through the SBWT_DEBUG_RING_*
  knobs, intervals wider than a ring entry holds.
* the sizing of the two search-prefix tables under a constrained memory budget (SBWT_DEBUG_FREE_BYTES caps what
  hipMemGetInfo reports) and their allocation failing (SBWT_DEBUG_FAIL_TABLE_ALLOC): the index loads without the
  table, slower, with identical results.
"""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_parity import COMP, assert_same, fq_reads, genome_text, sb  # noqa: F401  (sb is a fixture)

pytestmark = pytest.mark.gpu


def contig_genome(tmp_path, oracle, n_contigs, contig_len, seed, name):
    """many small contigs, a few with N runs, one 24-mer planted in 40 of them (multi-hit reads)"""
    rng = np.random.default_rng(seed)
    rep = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 24))
    seqs = []
    with open(tmp_path / (name + ".fa"), "w") as f:
        for k in range(n_contigs):
            s = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, contig_len))
            if k % 50 == 7 and contig_len >= 40:
                s = s[:8] + rep + s[32:]
            if k % 97 == 3:
                s = s[:contig_len // 2] + "NNNN" + s[contig_len // 2:]
            seqs.append(s)
            f.write(">c%d\n%s\n" % (k, s))
    prefix = str(tmp_path / name)
    oracle.build_index(str(tmp_path / (name + ".fa")), prefix)
    return prefix, seqs, rep


def contig_reads(seqs, rep, lens, n, seed):
    rng = np.random.default_rng(seed)
    reads = [rep, rep[:16], rep[4:].translate(COMP)[::-1]]
    while len(reads) < n:
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.choice(lens))
        if L > len(s):
            continue
        p = int(rng.integers(0, len(s) - L + 1))
        r = s[p:p + L]
        if "N" in r:
            continue
        if rng.random() < 0.5:
            r = r.translate(COMP)[::-1]
        if rng.random() < 0.2:
            k = int(rng.integers(0, L))
            r = r[:k] + "ACGT"[int(rng.integers(0, 4))] + r[k + 1:]
        reads.append(r)
    return reads


def test_more_keys_than_the_lds_tables_hold(sb, oracle, tmp_path):
    prefix, seqs, rep = contig_genome(tmp_path, oracle, 2000, 160, 5, "many")
    ix, oix = sb.Index.load(prefix), oracle.OracleIndex.load(prefix)
    info = ix.info()
    assert info.n_chr + info.n_npos > 1536
    for lens in ([20, 36, 50], [70, 100, 128, 150]):           # 64-base records and the long-read variant
        reads = contig_reads(seqs, rep, lens, 4000, lens[0])
        ctx = sb.Context(ix, len(reads))
        for strand in (0, 1, 2):
            res, _ = assert_same(sb, ctx, oix, reads, strand=strand)
        assert (res["nh"] == 1).sum() > len(reads) // 2
        if lens[0] == 20:
            assert res["nh"].max() >= 20 and ctx.last_stats()[1] >= 2    # the sorted multi-hit locate ran too
        ctx.close()
    # reads that straddle two contigs are found by the search and dropped by the filters
    span = [seqs[k][-25:] + seqs[k + 1][:25] for k in range(0, 300, 7) if "N" not in seqs[k][-25:] + seqs[k + 1][:25]]
    ctx = sb.Context(ix, len(span))
    res, _ = assert_same(sb, ctx, oix, span)
    assert (res["sa_hi"] > res["sa_lo"]).all() and (res["nh"] == 0).all()
    ctx.close(); ix.close(); oix.close()


@pytest.mark.parametrize("orders", ["12:0", "14:19"])
def test_lds_tables_forced_out_on_the_golden_genomes(sb, oracle, golden_indexes, golden_dir, monkeypatch, orders):
    monkeypatch.setenv("SBWT_KMER_ORDER", orders.split(":")[0])
    monkeypatch.setenv("SBWT_HASH_ORDER", orders.split(":")[1])
    monkeypatch.setenv("SBWT_DEBUG_LDS_KEYS", "0")            # read when the index (launch plan) and the ctx are made
    for g in ("g1", "g2"):
        ix, oix = sb.Index.load(golden_indexes[g]), oracle.OracleIndex.load(golden_indexes[g])
        seq = genome_text(os.path.join(golden_dir, g + ".fa"))
        rng = np.random.default_rng(3)
        reads = [r for r in fq_reads(os.path.join(golden_dir, g + ".fq")) if len(r) >= 12]
        for L in (12, 13, 20, 31, 50, 64, 65, 100, 128, 129, 200):
            for _ in range(60):
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
        short = [r for r in reads if len(r) <= 64]
        for batch in (short, reads):
            ctx = sb.Context(ix, len(batch))
            for strand in (2, 0):
                assert_same(sb, ctx, oix, batch, strand=strand)
            ctx.close()
        ix.close(); oix.close()


def test_chromosome_ids_above_16_bits(sb, oracle, tmp_path):
    """70,000 contigs: chr_id does not fit a ring entry (R_DIRECT) nor the sort payload (depth-first locate)"""
    prefix, seqs, rep = contig_genome(tmp_path, oracle, 70000, 40, 9, "wide")
    ix, oix = sb.Index.load(prefix), oracle.OracleIndex.load(prefix)
    assert ix.info().n_chr == 70000
    reads = contig_reads(seqs[60000:], rep, [14, 20, 30, 40], 3000, 1) + contig_reads(seqs, rep, [16, 25, 40], 3000, 2)
    ctx = sb.Context(ix, len(reads))
    res, hits = assert_same(sb, ctx, oix, reads)
    assert hits["chr_id"].max() > 65535 and res["nh"].max() >= 20
    assert ctx.last_stats()[3] >= 2                            # multi-hit reads went to the depth-first kernel
    ctx.close(); ix.close(); oix.close()


@pytest.mark.parametrize("knob,value", [("SBWT_DEBUG_RING_WIDTH", "2"), ("SBWT_DEBUG_RING_WIDTH", "0"), ("SBWT_DEBUG_RING_CHR", "0")])
def test_results_that_bypass_the_ring(sb, oracle, golden_indexes, golden_dir, monkeypatch, knob, value):
    monkeypatch.setenv(knob, value)
    ix, oix = sb.Index.load(golden_indexes["g2"]), oracle.OracleIndex.load(golden_indexes["g2"])
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    rng = np.random.default_rng(8)
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    rep = seq[1000:1300]
    reads += [rep[a:a + 25] for a in range(0, 270, 5)] * 3   # (three times: reads with one interval share a list, wide records too)
    while len(reads) < 3000:
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
    ctx = sb.Context(ix, len(reads))
    for strand in (2, 1):
        res, hits = assert_same(sb, ctx, oix, reads, strand=strand)
    assert (res["nh"] > 1).sum() >= 50
    ctx.close(); ix.close(); oix.close()


# ---- prefix-table sizing under a budget ---------------------------------------------------------
GB = 1 << 30
HEADROOM = 16 * GB                # device_index.hip: kKmerHeadroom
HASH8 = (1 << 25) * 32           # order-20 hash table of an index below 2^27 suffixes: 2^25 buckets of four 8-byte slots
HASH16 = (1 << 25) * 64          # ... of four 16-byte slots (with the text behind the prefix)
DIRECT13 = (4 ** 13) * 8


@pytest.fixture(scope="module")
def budget_world(oracle):
    import sbwt_amd as sb_
    from test_gpu_scale import pack_codes
    sb_.device_count()
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.default_rng(77)
    g = 2_500_000                                               # n = 5,000,001 >= 2^22: the loader wants a hash table
    codes = rng.integers(0, 4, g).astype(np.uint8)
    chr_start = np.array([0, 1_000_000], dtype=np.uint64)
    npos_key = np.array([0, 1_000_000, g], dtype=np.uint64)
    npos_val = np.array([0, 10, 0], dtype=np.uint64)
    host = np.concatenate([pack_codes(codes), np.zeros(4, np.uint32)])
    d_fwd = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_fwd), host.nbytes) == 0
    assert hip.hipMemcpy(d_fwd, host.ctypes.data, host.nbytes, 1) == 0

    def build(want_tables=False):
        return sb_.build_index_device(d_fwd.value, g, chr_start, npos_key, npos_val, interval=64, device=0,
                                      want_index=True, want_tables=want_tables)

    ix0, tables = build(want_tables=True)
    T = tables.contents
    c4 = (C.c_uint64 * 4)(*T.c_function4)
    h = oracle.lib().sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions,
                                    T.occ[0], T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt,
                                    T.location_table, T.n_loc, chr_start.size, chr_start.ctypes.data, npos_key.size,
                                    npos_key.ctypes.data, npos_val.ctypes.data)
    oix = oracle.OracleIndex(h, keep=(c4, chr_start, npos_key, npos_val))
    from test_gpu_scale import sample_reads
    reads = sample_reads(rng, codes, 6000, [20, 36, 50, 100])
    want = oix.map_batch(reads, nthreads=8)
    yield sb_, build, ix0, reads, want
    oix.close()
    ix0.close()
    sb_.host_tables_free(tables)
    hip.hipFree(d_fwd)


def same_as(sb_, ix, reads, want):
    lo, hi, nh, first, ohits = want
    packed, lens = sb_.pack_reads(reads)
    ctx = sb_.Context(ix, len(reads))
    res, hits = ctx.map_batch(packed, lens)
    lines = ctx.last_stats()[0]
    ctx.close()
    np.testing.assert_array_equal(res["sa_lo"], lo)
    np.testing.assert_array_equal(res["sa_hi"], hi)
    np.testing.assert_array_equal(res["nh"], nh)
    idx = lambda f, k: (np.repeat(f, k) + (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k))).astype(np.int64)  # noqa: E731
    assert np.array_equal(hits[idx(res["first_hit"], nh)], ohits[idx(first, nh)])
    ctx = sb_.Context(ix, len(reads))
    rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb_.MAP_HITS_ONLY)
    ctx.close()
    res2, hits2 = sb_.compact_expand(rec, lists, wide)
    np.testing.assert_array_equal(res2["nh"], nh)
    assert np.array_equal(hits2[idx(res2["first_hit"], nh)], ohits[idx(first, nh)])
    return lines


def orders(ix):
    info = ix.info()
    return (info.kmer_order, info.hash_order, info.hash_text)


def test_unconstrained_choice(budget_world):
    sb_, build, ix0, reads, want = budget_world
    assert orders(ix0) == (13, 20, 1)
    same_as(sb_, ix0, reads, want)


@pytest.mark.parametrize("free_bytes,expect", [
    (HEADROOM + HASH16 + DIRECT13 + 4096, (13, 20, 1)),   # everything just fits
    (HEADROOM + HASH16 + 4096, (12, 20, 1)),              # the hash table with text fits, no direct table beside it
    (HEADROOM + HASH8 + DIRECT13 + 4096, (13, 20, 0)),    # 8-byte slots and the direct table
    (HEADROOM + DIRECT13 + 4096, (13, 0, 0)),             # no room for a hash table: direct table alone
    (1 * GB, (12, 0, 0)),                                 # nothing fits: c_functions and rank queries only
])
def test_table_sizes_follow_the_budget(budget_world, monkeypatch, free_bytes, expect):
    sb_, build, ix0, reads, want = budget_world
    monkeypatch.setenv("SBWT_DEBUG_FREE_BYTES", str(free_bytes))
    ix, _ = build()
    assert orders(ix) == expect
    lines = same_as(sb_, ix, reads, want)
    if expect == (12, 0, 0):
        assert lines > 3 * len(reads)                   # no memoised prefix: c_functions, rank steps, then the text
    ix.close()


@pytest.mark.parametrize("fail,expect", [("hash", (13, 0, 0)), ("kmer", (12, 20, 1)), ("both", (12, 0, 0))])
def test_table_allocation_failure_is_survived(budget_world, monkeypatch, fail, expect):
    sb_, build, ix0, reads, want = budget_world
    monkeypatch.setenv("SBWT_DEBUG_FAIL_TABLE_ALLOC", fail)
    ix, _ = build()
    assert orders(ix) == expect
    same_as(sb_, ix, reads, want)
    ix.close()
