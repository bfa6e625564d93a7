"""CPU: pins the oracle (oracle/sbwt_oracle.c) to the reference program's own outputs.

golden.json holds md5 sums of files the upstream binary produced on the G1/G2/G3 inputs
(SURVEY.md Appendix A).  The oracle must reproduce them byte for byte: SAM for G1/G2 (and G3 when
SBWT_TEST_G3=1, ~1 minute), the index file t_table.bwt and NposLen.z for G1.
"""
import hashlib
import json
import os
import struct
import subprocess
import sys

import pytest


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "golden.json")))


def test_committed_inputs_match_recorded_md5(gold, golden_dir):
    for g in ("g1", "g2"):
        assert md5(os.path.join(golden_dir, g + ".fa")) == gold[g]["input_fa_md5"]
        assert md5(os.path.join(golden_dir, g + ".fq")) == gold[g]["input_fq_md5"]
        assert md5(os.path.join(golden_dir, g + ".sam")) == gold[g]["ref_sam_md5"]


def test_oracle_index_files_match_reference(gold, golden_indexes):
    p = golden_indexes["g1"]
    assert os.path.getsize(p + ".t_table.bwt") == gold["g1"]["ref_t_table_bytes"]
    assert md5(p + ".t_table.bwt") == gold["g1"]["ref_t_table_md5"]
    assert md5(p + ".NposLen.z") == gold["g1"]["ref_nposlen_md5"]
    assert open(p + ".chrStart").read() == "chrA\t0\nchrB\t19863\n"
    assert open(p + ".chrLen").read() == "chrA\t19863\nchrB\t15000\n"


def test_oracle_npos_map_g2(gold, golden_indexes):
    d = open(golden_indexes["g2"] + ".NposLen.z", "rb").read()
    cnt = struct.unpack_from("<Q", d, 45)[0]
    body = d[45 + 8 + 4 + 5:]
    got = {str(struct.unpack_from("<Q", body, 16 * k)[0]): struct.unpack_from("<Q", body, 16 * k + 8)[0]
           for k in range(cnt)}
    assert got == gold["g2"]["ref_npos_map"]


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_oracle_sam_matches_reference(g, gold, golden_dir, golden_indexes, oracle, tmp_path):
    ix = oracle.OracleIndex.load(golden_indexes[g])
    sam = str(tmp_path / "o.sam")
    rc, stats = ix.map_fastq(os.path.join(golden_dir, g + ".fq"), sam)
    ix.close()
    assert rc == 0
    assert md5(sam) == gold[g]["ref_sam_md5"]
    assert open(sam, "rb").read() == open(os.path.join(golden_dir, g + ".sam"), "rb").read()
    if g == "g2":
        assert stats == [8, 5, 3, 2, 9]


def test_oracle_quirks(golden_dir, golden_indexes, oracle, tmp_path):
    """Behaviour SURVEY.md 8a lists as part of parity."""
    ix = oracle.OracleIndex.load(golden_indexes["g2"])
    reads = [l for l in open(os.path.join(golden_dir, "g2.fq")).read().split("\n")]
    # last record without trailing newline is silently dropped (fastq_reader_impl.hpp:94-98)
    fq = tmp_path / "nonl.fq"
    fq.write_text("\n".join(reads[:-1]))
    rc, st = ix.map_fastq(str(fq), str(tmp_path / "a.sam"))
    assert rc == 0 and st[0] == 7
    # a read with N aborts upstream (searcher.hpp:399); everything before it was already written
    fq2 = tmp_path / "n.fq"
    fq2.write_text("@a\n%s\n+\n%s\n@b\nACGTNACGTACGTA\n+\nIIIIIIIIIIIIII\n" % (reads[1], reads[3]))
    rc, st = ix.map_fastq(str(fq2), str(tmp_path / "b.sam"))
    assert rc == -1 and st[0] == 1
    # shorter than 12 aborts too (searcher.hpp:404)
    fq3 = tmp_path / "s.fq"
    fq3.write_text("@b\nACGTACGTACG\n+\nIIIIIIIIIII\n")
    rc, st = ix.map_fastq(str(fq3), str(tmp_path / "c.sam"))
    assert rc == -2
    # suppression compares the RAW interval with -m (searcher.hpp:415): the 3-hit reads vanish at -m 2
    rc, st = ix.map_fastq(os.path.join(golden_dir, "g2.fq"), str(tmp_path / "d.sam"), limit=2)
    assert st == [8, 3, 3, 0, 3]
    # strand filter (searcher.hpp:516-517)
    rc, s0 = ix.map_fastq(os.path.join(golden_dir, "g2.fq"), str(tmp_path / "e.sam"), strand=0)
    rc, s1 = ix.map_fastq(os.path.join(golden_dir, "g2.fq"), str(tmp_path / "f.sam"), strand=1)
    assert s0[4] + s1[4] == 9 and s0[4] == 6
    ix.close()


def test_oracle_rank_against_bruteforce(golden_dir, golden_indexes, oracle):
    """Table-level known answers from a brute-force suffix array of a tiny genome."""
    import numpy as np
    fa = os.path.join(golden_dir, "g2.fa")
    chroms = []
    for line in open(fa):
        if line.startswith(">"):
            chroms.append("")
        else:
            chroms[-1] += line.strip().upper()
    fwd = "".join(c.replace("N", "") for c in chroms)
    comp = str.maketrans("ACGT", "TGCA")
    text = fwd + fwd.translate(comp)[::-1] + "$"
    sa = sorted(range(len(text)), key=lambda i: text[i:])
    bwt = "".join(text[i - 1] if i else "$" for i in sa)
    ix = oracle.OracleIndex.load(golden_indexes["g2"])
    rng = np.random.default_rng(0)
    for i in list(rng.integers(0, len(text) + 1, 400)) + [0, len(text)]:
        for c, ch in enumerate("ACGT"):
            assert ix.occ(int(i), c) == bwt[:int(i)].count(ch)
    ix.close()


@pytest.mark.skipif(os.environ.get("SBWT_TEST_G3") != "1", reason="set SBWT_TEST_G3=1 (1M reads, ~1 min of CPU)")
def test_oracle_g3_matches_reference(gold, golden_dir, oracle, tmp_path):
    subprocess.check_call([sys.executable, os.path.join(golden_dir, "make_inputs.py"), str(tmp_path), "g3"])
    assert md5(str(tmp_path / "g3.fa")) == gold["g3"]["input_fa_md5"]
    oracle.build_index(str(tmp_path / "g3.fa"), str(tmp_path / "g3"))
    assert md5(str(tmp_path / "g3.t_table.bwt")) == gold["g3"]["ref_t_table_md5"]
    ix = oracle.OracleIndex.load(str(tmp_path / "g3"))
    rc, st = ix.map_fastq(str(tmp_path / "g3.fq"), str(tmp_path / "o.sam"))
    ix.close()
    assert rc == 0 and st[0] == 1000000
    assert md5(str(tmp_path / "o.sam")) == gold["g3"]["ref_sam_md5"]
