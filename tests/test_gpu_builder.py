"""GPU: the index builder (sbwt build) -- byte-identical files on repeat-free genomes, and mapping parity
on an index it built."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def sb():
    import sbwt_amd
    assert sbwt_amd.device_count() >= 1
    return sbwt_amd


def test_g1_index_files_are_byte_identical_to_upstream(sb, golden_dir, tmp_path):
    gold = json.load(open(os.path.join(golden_dir, "golden.json")))["g1"]
    prefix = str(tmp_path / "g1")
    sb.build_index(os.path.join(golden_dir, "g1.fa"), prefix)
    assert os.path.getsize(prefix + ".t_table.bwt") == gold["ref_t_table_bytes"]
    assert md5(prefix + ".t_table.bwt") == gold["ref_t_table_md5"]
    assert md5(prefix + ".NposLen.z") == gold["ref_nposlen_md5"]
    assert open(prefix + ".chrStart").read() == "chrA\t0\nchrB\t19863\n"
    assert open(prefix + ".chrLen").read() == "chrA\t19863\nchrB\t15000\n"


@pytest.mark.parametrize("interval", [64, 16, 256])
def test_builder_matches_oracle_builder_on_random_genome(sb, oracle, tmp_path, interval):
    rng = np.random.default_rng(interval)
    parts = ["".join("ACGT"[c] for c in rng.integers(0, 4, n)) for n in (70001, 1234, 50000)]
    fa = tmp_path / "r.fa"
    fa.write_text(">c1\n%s\n>c2\nNNN%sNNNNN%s\n" % (parts[0], parts[1], parts[2]))
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    sb.build_index(str(fa), a, interval=interval)
    oracle.build_index(str(fa), b, interval=interval)
    for ext in (".t_table.bwt", ".NposLen.z", ".chrStart", ".chrLen"):
        assert md5(a + ext) == md5(b + ext), ext


def test_repeat_genome_maps_like_oracle_on_the_built_index(sb, oracle, golden_dir, tmp_path):
    """G2 has 300-bp exact repeats: suffix ties deeper than sort_len may be ordered differently from the
    oracle's builder (any order is a valid k-order), so parity is checked on THIS index: the oracle loads
    the files the GPU builder wrote."""
    prefix = str(tmp_path / "g2")
    sb.build_index(os.path.join(golden_dir, "g2.fa"), prefix)
    ix = sb.Index.load(prefix)
    oix = oracle.OracleIndex.load(prefix)
    a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    st = ix.map_fastq(os.path.join(golden_dir, "g2.fq"), a)
    rc, ost = oix.map_fastq(os.path.join(golden_dir, "g2.fq"), b)
    assert rc == 0 and st == ost
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a, "rb").read() == open(os.path.join(golden_dir, "g2.sam"), "rb").read()
    ix.close(); oix.close()


def test_cli_build_then_map(sb, golden_dir, tmp_path):
    exe = os.path.join(os.path.dirname(sb.lib_path()), "..", "bin", "sbwt")
    prefix = str(tmp_path / "cli")
    r = subprocess.run([exe, "build", "-i", os.path.join(golden_dir, "g1.fa"), "-p", prefix], capture_output=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "build", "-i", os.path.join(golden_dir, "g1.fa"), "-p", prefix], capture_output=True)
    assert r.returncode == 2 and b"already exist" in r.stderr           # main.cpp:113-117
    sam = str(tmp_path / "o.sam")
    r = subprocess.run([exe, "map", "-i", os.path.join(golden_dir, "g1.fq"), "-p", prefix, "-o", sam, "-n", "2"],
                       capture_output=True)
    assert r.returncode == 0, r.stderr
    assert b"Total reads: 20\tSingle mapped reads: 20\tMultiple mapped reads: 0" in r.stderr
    assert open(sam, "rb").read() == open(os.path.join(golden_dir, "g1.sam"), "rb").read()
    r = subprocess.run([exe, "map", "-i", os.path.join(golden_dir, "g1.fq"), "-p", prefix], capture_output=True)
    assert r.stdout == open(os.path.join(golden_dir, "g1.sam"), "rb").read()  # stdout by default


def test_g3_config1_index_and_sam_md5_equal_the_reference_binary(sb, golden_dir, tmp_path):
    """BASELINE.json configs[0] stand-in: 4,641,652-bp genome, 1 M 50-bp reads.  The GPU builder's
    t_table.bwt and the GPU mapper's SAM must have the md5 sums recorded from the upstream binary's own files."""
    import hashlib
    import json
    import subprocess
    import sys
    gold = json.load(open(os.path.join(golden_dir, "golden.json")))["g3"]
    subprocess.check_call([sys.executable, os.path.join(golden_dir, "make_inputs.py"), str(tmp_path), "g3"])

    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
        return h.hexdigest()

    assert md5(str(tmp_path / "g3.fa")) == gold["input_fa_md5"]
    prefix = str(tmp_path / "g3")
    sb.build_index(str(tmp_path / "g3.fa"), prefix)
    assert md5(prefix + ".t_table.bwt") == gold["ref_t_table_md5"]
    ix = sb.Index.load(prefix)
    for nt in (1, 8):
        sam = str(tmp_path / ("o%d.sam" % nt))
        st = ix.map_fastq(str(tmp_path / "g3.fq"), sam, n_threads=nt)
        assert st[0] == 1_000_000
        assert md5(sam) == gold["ref_sam_md5"]
    ix.close()
