"""GPU: parity of the HIP path (through the C ABI) with the oracle and with the golden SAM files."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = str.maketrans("ACGT", "TGCA")


def fq_reads(path):
    lines = open(path).read().split("\n")
    return [lines[i + 1].strip() for i in range(0, len(lines) - 3, 4)]


def genome_text(fa):
    return "".join(l.strip() for l in open(fa) if not l.startswith(">")).upper()


def assert_same(sb, ctx, oix, reads, limit=1000, strand=2):
    packed, lens = sb.pack_reads(reads)
    res, hits = ctx.map_batch(packed, lens, limit=limit, strand=strand)
    lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand)
    np.testing.assert_array_equal(res["sa_lo"], lo)     # identical SA intervals, dead reads included
    np.testing.assert_array_equal(res["sa_hi"], hi)
    np.testing.assert_array_equal(res["nh"], nh)
    assert int(nh.sum()) == len(ohits)
    for r in np.nonzero(nh)[0]:
        a = hits[int(res["first_hit"][r]): int(res["first_hit"][r]) + int(nh[r])]
        b = ohits[int(first[r]): int(first[r]) + int(nh[r])]
        assert np.array_equal(a, b), "read %d: hit list differs (order matters)" % r
    assert_same_compact(sb, ctx, reads, packed, lens, lo, hi, nh, first, ohits, limit, strand)
    return res, hits


def assert_same_compact(sb, ctx, reads, packed, lens, lo, hi, nh, first, ohits, limit, strand):
    """the same batch through the compact records (sbwt_map_batch_compact), with SA intervals and hits-only: every
    parity test of this suite covers both layouts"""
    for mode in (0, sb.MAP_HITS_ONLY):
        rec, lists, wide = ctx.map_batch_compact(packed, lens, limit=limit, strand=strand, mode=mode)
        meta = (rec["w1"] >> np.uint64(56)).astype(np.uint32)
        res, hits = sb.compact_expand(rec, lists, wide)
        np.testing.assert_array_equal(res["nh"], nh)
        no_sa = (meta & sb.CR_NO_SA) != 0
        if mode == 0:
            assert not no_sa.any()
        else:
            # settled without an interval: exactly one row (nh 1, or 0 after the filters / when suppressed) or dead
            assert (nh[no_sa] <= 1).all()
        keep = ~no_sa
        np.testing.assert_array_equal(res["sa_lo"][keep], lo[keep])
        np.testing.assert_array_equal(res["sa_hi"][keep], hi[keep])
        np.testing.assert_array_equal(res["flags"][keep], ctx_flags(lo[keep], hi[keep], limit))
        for r in np.nonzero(nh)[0]:
            a = hits[int(res["first_hit"][r]): int(res["first_hit"][r]) + int(nh[r])]
            b = ohits[int(first[r]): int(first[r]) + int(nh[r])]
            assert np.array_equal(a, b), "compact mode %d, read %d: hit list differs" % (mode, r)


def ctx_flags(lo, hi, limit):
    """SBWT_RF_SUPPRESSED exactly where the raw interval exceeds -m (unsigned compare, searcher.hpp:415)"""
    width = np.where(hi > lo, hi - lo, 0).astype(np.uint64)
    lim = np.uint64(limit & 0xFFFFFFFFFFFFFFFF)
    return (width > lim).astype(np.uint32)


@pytest.fixture(scope="module")
def sb():
    import sbwt_amd
    assert os.path.exists(sbwt_amd.lib_path()), "libsbwt_amd.so must be built in-tree before GPU tests"
    assert sbwt_amd.device_count() >= 1
    return sbwt_amd


@pytest.fixture(scope="module", params=["12:0", "13:0", "14:0", "15:0", "16:0", "12:14", "14:19", "16:20", "13:28"])
def loaded(request, sb, oracle, golden_indexes):
    """every test runs with no memoised search prefix (12) and with the 13- to 16-base tables the loader picks for
    larger indexes (forced here: the golden genomes are tiny); odd orders take one single-base step after the
    prefix; order 17, what a GRCh38-sized index gets, has a test of its own below (its table is 137 GB)"""
    # "direct order:hash order" -- the second is the hash table of located prefixes (0 = none) that larger indexes
    # get on top of the direct table
    os.environ["SBWT_KMER_ORDER"], os.environ["SBWT_HASH_ORDER"] = request.param.split(":")
    out = {}
    for g, prefix in golden_indexes.items():
        out[g] = (sb.Index.load(prefix), oracle.OracleIndex.load(prefix))
    os.environ.pop("SBWT_KMER_ORDER")
    os.environ.pop("SBWT_HASH_ORDER")
    yield out
    for ix, oix in out.values():
        ix.close()
        oix.close()


def test_order_17_prefix_table(sb, oracle, golden_indexes, golden_dir):
    """the 4^17-entry table (34-bit entry numbers) on the repeat genome: golden reads, every length 12..60 on
    both strands, substitutions"""
    os.environ["SBWT_KMER_ORDER"] = "17"
    os.environ["SBWT_HASH_ORDER"] = "0"
    try:
        ix = sb.Index.load(golden_indexes["g2"])
    finally:
        os.environ.pop("SBWT_KMER_ORDER")
        os.environ.pop("SBWT_HASH_ORDER")
    oix = oracle.OracleIndex.load(golden_indexes["g2"])
    chroms = []
    for line in open(os.path.join(golden_dir, "g2.fa")):
        if line.startswith(">"):
            chroms.append("")
        else:
            chroms[-1] += line.strip().upper()
    seq = chroms[0].replace("N", "")
    rng = np.random.default_rng(17)
    comp = str.maketrans("ACGT", "TGCA")
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    for L in list(range(12, 61)) + [64, 65, 100, 129]:
        for _ in range(24):
            p = int(rng.integers(0, len(seq) - L))
            s = seq[p:p + L]
            if rng.random() < 0.5:
                s = s.translate(comp)[::-1]
            if rng.random() < 0.4:
                q = int(rng.integers(0, L))
                s = s[:q] + "ACGT"[int(rng.integers(0, 4))] + s[q + 1:]
            reads.append(s)
    ctx = sb.Context(ix, len(reads))
    for strand in (2, 0, 1):
        assert_same(sb, ctx, oix, reads, strand=strand)
    ctx.close()
    ix.close()
    oix.close()


@pytest.mark.parametrize("g", ["g1", "g2"])
@pytest.mark.parametrize("strand", [2, 0, 1])
def test_golden_reads_match_oracle(sb, loaded, golden_dir, g, strand):
    ix, oix = loaded[g]
    reads = [r for r in fq_reads(os.path.join(golden_dir, g + ".fq")) if len(r) >= 12]
    ctx = sb.Context(ix, len(reads))
    assert_same(sb, ctx, oix, reads, strand=strand)
    ctx.close()


@pytest.mark.parametrize("limit", [0, 1, 2, 3, 1000, -1])
def test_suppression_limit(sb, loaded, golden_dir, limit):
    ix, oix = loaded["g2"]
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    ctx = sb.Context(ix, len(reads))
    res, _ = assert_same(sb, ctx, oix, reads, limit=limit)
    if limit == 2:
        assert (res["flags"][:2] & sb.RF_SUPPRESSED).all()
    ctx.close()


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_reads_of_65_to_128_bases(sb, loaded, golden_dir, g):
    """batches whose records hold 65..128 bases run the kernel variant that compares up to 128 bases in one
    step: every length 60..128 (a few shorter ones ride in the same records), both strands, substitutions at
    every depth, three strand filters"""
    ix, oix = loaded[g]
    seq = genome_text(os.path.join(golden_dir, g + ".fa"))
    rng = np.random.default_rng(128)
    reads = []
    for L in list(range(60, 129)) + [12, 20, 21, 33, 50]:
        for _ in range(40):
            p = int(rng.integers(0, len(seq) - L))
            s = seq[p:p + L]
            if "N" in s:
                continue
            if rng.random() < 0.5:
                s = s.translate(COMP)[::-1]
            if rng.random() < 0.4:
                k = int(rng.integers(0, L))
                s = s[:k] + "ACGT"[int(rng.integers(0, 4))] + s[k + 1:]
            reads.append(s)
    assert max(len(r) for r in reads) == 128 and sb.words_per_read(128) == 8
    ctx = sb.Context(ix, len(reads))
    for strand in (2, 0, 1):
        res, _ = assert_same(sb, ctx, oix, reads, strand=strand)
    ctx.close()


def test_reads_of_129_to_260_bases(sb, loaded, golden_dir):
    """three to five 64-base pieces: the long-read variant compares the first piece alone and then two pieces at a
    time; substitutions at every depth kill the comparison in any piece"""
    ix, oix = loaded["g1"]
    seq = genome_text(os.path.join(golden_dir, "g1.fa"))
    rng = np.random.default_rng(260)
    reads = []
    for L in list(range(129, 261, 2)) + [64, 100, 128]:
        for _ in range(24):
            p = int(rng.integers(0, len(seq) - L))
            s = seq[p:p + L]
            if "N" in s:
                continue
            if rng.random() < 0.5:
                s = s.translate(COMP)[::-1]
            if rng.random() < 0.5:
                k = int(rng.integers(0, L))
                s = s[:k] + "ACGT"[int(rng.integers(0, 4))] + s[k + 1:]
            reads.append(s)
    ctx = sb.Context(ix, len(reads))
    res, _ = assert_same(sb, ctx, oix, reads)
    assert (res["nh"] > 0).sum() > len(reads) // 3
    ctx.close()


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_random_reads_all_lengths(sb, loaded, golden_dir, g):
    """substrings of both strands, lengths 12..200 (odd and even step counts, 1..4 sixteen-byte chunks),
    30 % with one substitution so the interval dies at every possible step parity"""
    ix, oix = loaded[g]
    seq = genome_text(os.path.join(golden_dir, g + ".fa"))
    rng = np.random.default_rng(11)
    reads = []
    for L in list(range(12, 40)) + [49, 50, 63, 64, 65, 66, 100, 127, 128, 129, 193, 200]:
        for _ in range(120):
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
    res, _ = assert_same(sb, ctx, oix, reads)
    assert (res["nh"] > 0).sum() > len(reads) // 2
    ctx.close()


def test_edge_reads(sb, loaded, golden_dir):
    """poly-A / poly-T (the '$' row and the false candidates at the end of the text), text ends,
    chromosome-spanning and N-gap-spanning reads, repeats with 3 and 6 hits"""
    ix, oix = loaded["g2"]
    fa = os.path.join(golden_dir, "g2.fa")
    chroms = []
    for line in open(fa):
        if line.startswith(">"):
            chroms.append("")
        else:
            chroms[-1] += line.strip().upper()
    fwd = "".join(c.replace("N", "") for c in chroms)
    rcs = fwd.translate(COMP)[::-1]
    reads = ["A" * 12, "T" * 12, "A" * 30, "C" * 13, "G" * 64, "ACGT" * 5,
             fwd[:12], fwd[:13], fwd[-12:], fwd[-13:], fwd[-40:], rcs[:12], rcs[-12:], rcs[-25:],
             fwd[3590:3610], fwd[3600 - 12:3600], fwd[3600:3612], fwd[3708:3732], fwd[3715:3727]]
    rep = chroms[0][1000:1300]
    for a, b in ((0, 12), (0, 100), (5, 90), (100, 112), (150, 300), (280, 300)):
        reads += [rep[a:b], rep[a:b].translate(COMP)[::-1]]
    ctx = sb.Context(ix, len(reads))
    res, _ = assert_same(sb, ctx, oix, reads)
    assert res["nh"].max() >= 3
    ctx.close()


def test_empty_batch_and_capacity(sb, loaded, golden_dir):
    ix, oix = loaded["g2"]
    ctx = sb.Context(ix, 16)
    res, hits = ctx.map_batch(np.zeros((0, 4), np.uint32), np.zeros(0, np.uint32))
    assert len(res) == 0 and len(hits) == 0
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    packed, lens = sb.pack_reads(reads)
    with pytest.raises(sb.SbwtError) as e:       # 9 hit slots needed
        ctx.map_batch(packed, lens, hits_cap=4)
    assert e.value.code == -7
    with pytest.raises(sb.SbwtError) as e:       # more reads than the ctx was sized for
        ctx.map_batch(np.zeros((17, 4), np.uint32), np.full(17, 12, np.uint32))
    assert e.value.code == -4
    ctx.close()


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_map_fastq_sam_is_byte_identical_to_reference(sb, loaded, golden_dir, tmp_path, g):
    gold = json.load(open(os.path.join(golden_dir, "golden.json")))
    ix, _ = loaded[g]
    for nt in (1, 3):
        sam = str(tmp_path / ("o%d.sam" % nt))
        stats = ix.map_fastq(os.path.join(golden_dir, g + ".fq"), sam, n_threads=nt)
        data = open(sam, "rb").read()
        assert hashlib.md5(data).hexdigest() == gold[g]["ref_sam_md5"]
        assert stats[0] == (20 if g == "g1" else 8)


def test_map_fastq_variants_match_oracle(sb, loaded, golden_dir, tmp_path):
    ix, oix = loaded["g2"]
    fq = os.path.join(golden_dir, "g2.fq")
    for kw in ({"limit": 2}, {"strand": 0}, {"strand": 1}, {"limit": 0}):
        a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
        st = ix.map_fastq(fq, a, **kw)
        rc, ost = oix.map_fastq(fq, b, **kw)
        assert rc == 0 and st == ost
        assert open(a, "rb").read() == open(b, "rb").read()


def test_map_fastq_abort_semantics(sb, loaded, golden_dir, tmp_path):
    """upstream prints every read before the offending one, then dies: same records, error status"""
    ix, oix = loaded["g2"]
    lines = open(os.path.join(golden_dir, "g2.fq")).read().split("\n")
    fq = tmp_path / "n.fq"
    fq.write_text("\n".join(lines[:8]) + "\n@bad\nACGTNACGTACGTA\n+\nIIIIIIIIIIIIII\n" + "\n".join(lines[8:]))
    a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    with pytest.raises(sb.SbwtError) as e:
        ix.map_fastq(str(fq), a)
    assert e.value.code == -5 and "read 2" in str(e.value)
    rc, _ = oix.map_fastq(str(fq), b)
    assert rc == -1
    assert open(a, "rb").read() == open(b, "rb").read()
    # final record without newline is dropped, like upstream's reader
    fq2 = tmp_path / "nonl.fq"
    fq2.write_text("\n".join(lines[:-1]))
    st = ix.map_fastq(str(fq2), a)
    rc, ost = oix.map_fastq(str(fq2), b)
    assert st == ost and st[0] == 7
    assert open(a, "rb").read() == open(b, "rb").read()


def test_corrupt_index_is_rejected(sb, golden_indexes, tmp_path):
    import shutil
    src = golden_indexes["g2"]
    for ext in (".t_table.bwt", ".NposLen.z", ".chrStart", ".chrLen"):
        shutil.copy(src + ext, str(tmp_path / ("c" + ext)))
    p = str(tmp_path / "c.t_table.bwt")
    data = bytearray(open(p, "rb").read())
    off = 40 + 8 + 8 + 256 * 8 + 8 + (4 ** 12 + 1) * 8 + 5 + 8 + 4 + 65 * 8 + 8 + 8 * 5   # occ_function['A'][5]
    data[off] ^= 1
    open(p, "wb").write(bytes(data))
    with pytest.raises(sb.SbwtError) as e:
        sb.Index.load(str(tmp_path / "c"))
    assert e.value.code == -2


@pytest.mark.parametrize("interval", [16, 256])
def test_other_sampling_intervals(sb, oracle, golden_dir, tmp_path, interval):
    """`sbwt build -s`: interval 16 (short locate walks) and 256 (the sorted multi-hit locate hands over to the
    depth-first kernel above 64)"""
    prefix = str(tmp_path / "g2")
    oracle.build_index(os.path.join(golden_dir, "g2.fa"), prefix, interval=interval)
    ix, oix = sb.Index.load(prefix), oracle.OracleIndex.load(prefix)
    assert ix.info().interval == interval
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    rng = np.random.default_rng(interval)
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    for L in (14, 20, 33, 50, 80):
        for _ in range(150):
            p = int(rng.integers(0, len(seq) - L))
            s = seq[p:p + L]
            if "N" in s:
                continue
            reads.append(s.translate(COMP)[::-1] if rng.random() < 0.5 else s)
    ctx = sb.Context(ix, len(reads))
    res, _ = assert_same(sb, ctx, oix, reads)
    assert res["nh"].max() >= 3
    st = ctx.last_stats()
    assert st[1] >= 2 and (st[3] >= 2 if interval > 64 else st[3] == 0)   # multi-hit reads; depth-first only above 64
    ctx.close(); ix.close(); oix.close()


def test_hits_layout_and_capacity_retry(sb, loaded, golden_dir):
    """unique hits live at their read's index, multi-hit lists behind the first n_reads slots; a buffer that holds
    the reads but not the lists fails with the size a retry needs"""
    ix, oix = loaded["g2"]
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    packed, lens = sb.pack_reads(reads)
    n = len(reads)
    ctx = sb.Context(ix, n)
    with pytest.raises(sb.SbwtError) as e:
        ctx.map_batch(packed, lens, hits_cap=n)          # room for the unique hits only
    assert e.value.code == -7 and "%d slots needed" % (n + 6) in str(e.value)
    res, hits = ctx.map_batch(packed, lens, hits_cap=n + 6)
    lo, hi, nh, first, ohits = oix.map_batch(reads)
    np.testing.assert_array_equal(res["nh"], nh)
    uniq = np.nonzero(nh == 1)[0]
    np.testing.assert_array_equal(res["first_hit"][uniq], uniq)
    multi = np.nonzero(nh > 1)[0]
    assert len(multi) == 2 and (res["first_hit"][multi] >= n).all()
    assert ctx.last_stats()[2] == n + 6 and ctx.last_stats()[0] > 0
    ctx.close()


def test_two_contexts_map_concurrently(sb, loaded, golden_dir):
    """calls on distinct sbwt_ctx objects may run concurrently (include/sbwt_amd.h; the upstream pool's functor
    calls do): two host threads, one ctx and stream each, same index, many batches"""
    import threading
    ix, oix = loaded["g2"]
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    rng = np.random.default_rng(2)
    batches = []
    for b in range(2):
        reads = []
        for _ in range(3000):
            L = int(rng.integers(12, 101))
            p = int(rng.integers(0, len(seq) - L))
            s = seq[p:p + L]
            if "N" in s:
                continue
            if rng.random() < 0.5:
                s = s.translate(COMP)[::-1]
            reads.append(s)
        batches.append(reads)
    want = [oix.map_batch(reads) for reads in batches]
    errors = []

    def work(k):
        try:
            ctx = sb.Context(ix, len(batches[k]))
            packed, lens = sb.pack_reads(batches[k])
            for _ in range(20):
                res, hits = ctx.map_batch(packed, lens)
                lo, hi, nh, first, ohits = want[k]
                assert np.array_equal(res["sa_lo"], lo) and np.array_equal(res["sa_hi"], hi) and np.array_equal(res["nh"], nh)
                for r in np.nonzero(nh)[0][:200]:
                    a = hits[int(res["first_hit"][r]): int(res["first_hit"][r]) + int(nh[r])]
                    assert np.array_equal(a, ohits[int(first[r]): int(first[r]) + int(nh[r])])
            ctx.close()
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_host_buffers_in_sub_batches(sb, loaded, golden_dir, monkeypatch):
    """sbwt_map_batch splits large batches so that copies overlap the kernels (SBWT_SUBBATCH = reads per
    sub-batch): forced to 128 here, over unique, dead and multi-hit reads, with a capacity retry"""
    ix, oix = loaded["g2"]
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    rng = np.random.default_rng(99)
    reads = [r for r in fq_reads(os.path.join(golden_dir, "g2.fq")) if len(r) >= 12]
    while len(reads) < 1500:
        L = int(rng.integers(12, 80))
        p = int(rng.integers(0, len(seq) - L))
        s_ = seq[p:p + L]
        if "N" in s_:
            continue
        if rng.random() < 0.5:
            s_ = s_.translate(COMP)[::-1]
        if rng.random() < 0.3:
            k = int(rng.integers(0, L))
            s_ = s_[:k] + "ACGT"[int(rng.integers(0, 4))] + s_[k + 1:]
        reads.append(s_)
    monkeypatch.setenv("SBWT_SUBBATCH", "128")
    ctx = sb.Context(ix, len(reads))
    for strand in (2, 0):
        res, _ = assert_same(sb, ctx, oix, reads, strand=strand)
    assert (res["nh"] > 1).sum() >= 2                     # multi-hit lists in more than one sub-batch's wake
    packed, lens = sb.pack_reads(reads)
    with pytest.raises(sb.SbwtError) as e:                # room for the unique slots only: the lists do not fit
        ctx.map_batch(packed, lens, hits_cap=len(reads))
    assert e.value.code == -7
    ctx.close()
    small = sb.Context(ix, 200)                           # a batch beyond the ctx capacity is refused before it is split
    with pytest.raises(sb.SbwtError) as e:
        small.map_batch(packed, lens)
    assert e.value.code == -4
    with pytest.raises(sb.SbwtError) as e:                # words_per_read must be a multiple of 4
        small.map_batch(np.zeros((100, 6), np.uint32), np.full(100, 12, np.uint32))
    assert e.value.code == -4
    res, _ = assert_same(sb, small, oix, reads[:200])     # and the ctx is still good
    small.close()
