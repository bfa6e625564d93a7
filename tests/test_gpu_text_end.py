"""GPU: the false candidates at the end of the text (searcher.hpp:504-539 on a wrapped position).

The c_functions table puts each of the 11 end-of-text suffixes shorter than 12 bases at the end of the preceding
present 12-mer's bucket.  A read whose reverse complement ends in that 12-mer K' therefore carries
the short suffix's row in its interval although the text does not match; extended along that row's BWT chain only
the false row survives.  Its reported position p has p + L > 2 * real_size, so upstream's
`position = 2 * _realSize - position - L` underflows and the hit is dropped by the upper_bound tests on the wrapped
value.  These tests BUILD such reads from a brute-force suffix array (no index tables involved) and compare the
HIP path with the oracle through the unique path, the sorted multi-hit locate and the depth-first locate.
"""
import os

import numpy as np
import pytest

from test_gpu_parity import COMP, assert_same, sb  # noqa: F401  (sb is a fixture)

pytestmark = pytest.mark.gpu


def chromosomes(fa):
    out = []
    for line in open(fa):
        if line.startswith(">"):
            out.append("")
        else:
            out[-1] += line.strip().upper()
    return out


def false_candidate_reads(chroms, max_ext=(0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 20, 38, 52, 70)):
    """reads that land on the rows of the short end-of-text suffixes: RC(ext + K'), ext = the t bases in front of
    the suffix; plus, for every K', the same reads with one wrong base in ext (they die) """
    fwd = "".join(c.replace("N", "") for c in chroms)
    text = fwd + fwd.translate(COMP)[::-1] + "$"
    n = len(text)
    sa = sorted(range(n), key=lambda p: text[p:])           # '$' < 'A' in ASCII
    rank = {p: i for i, p in enumerate(sa)}
    reads, rows = [], []
    for p in range(n - 12, n - 1):                          # suffixes with 1..11 bases before '$'
        i = rank[p]
        j = i - 1
        while j >= 0 and sa[j] + 12 > n - 1:                # the nearest row above that defines a 12-mer bucket
            j -= 1
        if j < 0:
            continue                                        # falls into the '$' bucket (code 0)
        kp = text[sa[j]:sa[j] + 12]
        rows.append(i)
        for t in max_ext:
            if t > p:
                continue
            query = text[p - t:p] + kp
            reads.append(query.translate(COMP)[::-1])
            if t:
                k = (7 * t + p) % t
                wrong = query[:k] + "ACGT"[("ACGT".index(query[k]) + 1) % 4] + query[k + 1:]
                reads.append(wrong.translate(COMP)[::-1])
    return reads, rows


@pytest.fixture(scope="module", params=["12:0", "13:0", "16:0", "12:14", "14:19", "16:20"])
def orders(request):
    return request.param


def load_with_orders(sb, prefix, orders):
    os.environ["SBWT_KMER_ORDER"], os.environ["SBWT_HASH_ORDER"] = orders.split(":")
    try:
        return sb.Index.load(prefix)
    finally:
        os.environ.pop("SBWT_KMER_ORDER")
        os.environ.pop("SBWT_HASH_ORDER")


def test_false_candidates_two_chromosomes(sb, oracle, golden_indexes, golden_dir, orders):
    reads, rows = false_candidate_reads(chromosomes(os.path.join(golden_dir, "g2.fa")))
    assert len(rows) >= 8 and len(reads) > 150
    ix = load_with_orders(sb, golden_indexes["g2"], orders)
    oix = oracle.OracleIndex.load(golden_indexes["g2"])
    ctx = sb.Context(ix, len(reads))
    for strand in (2, 0, 1):
        res, _ = assert_same(sb, ctx, oix, reads, strand=strand)
    # the reads do reach those rows: many end on a size-1 interval that reports nothing
    assert ((res["sa_hi"] - res["sa_lo"] == 1) & (res["nh"] == 0)).sum() >= len(rows)
    ctx.close(); ix.close(); oix.close()


@pytest.mark.parametrize("interval", [64, 256, 16])
def test_false_candidates_one_chromosome(sb, oracle, tmp_path, orders, interval):
    """one chromosome, no N: the chromosome test of push_position cannot reject (first and last segment are the
    same one), only the N-segment test on the wrapped value does; interval 256 takes the depth-first kernel"""
    rng = np.random.default_rng(1)
    seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 2500))
    seq = seq[:1200] + seq[300:420] + seq[1200:]           # a 120-base repeat: multi-hit reads near nothing special
    fa = tmp_path / "one.fa"
    fa.write_text(">solo\n" + seq + "\n")
    prefix = str(tmp_path / "one")
    oracle.build_index(str(fa), prefix, interval=interval)
    reads, rows = false_candidate_reads([seq])
    reads += [seq[300:340], seq[310:330].translate(COMP)[::-1], seq[-12:], seq[:12], seq[-30:].translate(COMP)[::-1]]
    ix = load_with_orders(sb, prefix, orders)
    oix = oracle.OracleIndex.load(prefix)
    ctx = sb.Context(ix, len(reads))
    for strand in (2, 0, 1):
        res, _ = assert_same(sb, ctx, oix, reads, strand=strand)
    assert ((res["sa_hi"] - res["sa_lo"] == 1) & (res["nh"] == 0)).sum() >= len(rows)
    assert (res["nh"] > 1).sum() >= 2
    ctx.close(); ix.close(); oix.close()
