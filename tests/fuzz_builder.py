"""GPU builder fuzzer: random FASTA files through `sbwt build` (sbwt_build_index, builder.hip) and through the oracle's
restatement of the upstream builder (Indexer::build indexer.hpp:137-229, Genome_pre_handler::run_handler
genome_pre_handler.hpp:175-242); the four index files byte for byte, then FASTQ -> SAM on the index the GPU wrote
against the oracle on the same files.

Test infrastructure (it drives the oracle as the checker): `python tests/fuzz_builder.py --seeds 0:100` on a GPU box;
tests/test_gpu_fuzz.py runs a few seeds.  One seed = 1..6 contigs of random bases with N runs at the ends and inside
(what NposLen.z and chrStart / chrLen describe), random FASTA line width, lower-case stretches, a random SA sampling
interval; some seeds plant an exact duplication longer than sort_len 256 -- its suffixes tie, any order of them is a
valid index, so those seeds compare the mapping only (the oracle loads the GPU's files)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def make_fasta(rng):
    """-> (fasta text as bytes, list of N-free upper-case pieces for reads, has_long_duplication)"""
    n_chr = int(rng.integers(1, 7))
    out = []
    pieces = []
    dup = None
    has_dup = rng.random() < 0.3
    if has_dup:
        dup = bytes(b"ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(280, 900))))
    width = int(rng.choice([50, 60, 70, 80, 1000, 10 ** 9]))
    lower = rng.random() < 0.3
    for k in range(n_chr):
        seq = b""
        n_seg = int(rng.integers(1, 6))
        for s in range(n_seg):
            if rng.random() < 0.4:
                seq += b"N" * int(rng.choice([1, 2, 7, 63, 64, 65, 500]))
            L = int(rng.choice([1, 13, 300, 5000, 30000]))
            piece = bytes(b"ACGT"[c] for c in rng.integers(0, 4, L))
            if has_dup and rng.random() < 0.6 and L > len(dup):
                at = int(rng.integers(0, L - len(dup)))
                piece = piece[:at] + dup + piece[at + len(dup):]
            pieces.append(piece)
            if lower and rng.random() < 0.5:
                a, b = sorted(int(x) for x in rng.integers(0, L + 1, 2))
                piece = piece[:a] + piece[a:b].lower() + piece[b:]
            seq += piece
        if rng.random() < 0.3:
            seq += b"N" * int(rng.choice([1, 10, 100]))
        out.append(b">c%d\n" % (k + 1))
        for a in range(0, len(seq), width):
            out.append(seq[a:a + width] + b"\n")
    return b"".join(out), pieces, has_dup


def make_fastq(rng, pieces, n):
    big = [p for p in pieces if len(p) >= 12]
    out = []
    k = 0
    while k < n and big:
        p = big[int(rng.integers(0, len(big)))]
        L = min(len(p), int(rng.choice([12, 20, 33, 36, 50, 64, 65, 100, 150])))
        a = int(rng.integers(0, len(p) - L + 1))
        s = p[a:a + L]
        if rng.random() < 0.5:
            s = s.translate(COMP)[::-1]
        if rng.random() < 0.2:
            q = int(rng.integers(0, L))
            s = s[:q] + b"ACGT"[int(rng.integers(0, 4)):][:1] + s[q + 1:]
        out.append(b"@r%d\n%s\n+\n%s\n" % (k, s, b"I" * L))
        k += 1
    return b"".join(out), k


def run_seed(seed, sb, oracle, verbose=True):
    rng = np.random.default_rng(7_000_003 * seed + 5)
    fasta, pieces, has_dup = make_fasta(rng)
    interval = int(rng.choice([64, 64, 16, 32, 128, 256]))
    problems = []
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "g.fa")
        open(fa, "wb").write(fasta)
        a, b = os.path.join(td, "a"), os.path.join(td, "b")
        t0 = time.time()
        sb.build_index(fa, a, interval=interval)
        t_gpu = time.time() - t0
        t0 = time.time()
        oracle.build_index(fa, b, interval=interval)
        t_cpu = time.time() - t0
        for ext in (".t_table.bwt", ".NposLen.z", ".chrStart", ".chrLen"):
            if has_dup and ext == ".t_table.bwt":
                continue
            if open(a + ext, "rb").read() != open(b + ext, "rb").read():
                problems.append("%s differs (%d bytes, oracle %d)" % (ext, os.path.getsize(a + ext), os.path.getsize(b + ext)))
        fq = os.path.join(td, "r.fq")
        text, n = make_fastq(rng, pieces, 1500)
        open(fq, "wb").write(text)
        if n:
            ix = sb.Index.load(a)
            oix = oracle.OracleIndex.load(a)
            try:
                sa, sb_ = os.path.join(td, "a.sam"), os.path.join(td, "b.sam")
                st = ix.map_fastq(fq, sa, n_threads=int(rng.choice([1, 4])))
                rc, ost = oix.map_fastq(fq, sb_)
                if rc != 0 or list(st) != list(ost):
                    problems.append("map_fastq counters %s, oracle rc %d %s" % (list(st), rc, list(ost)))
                elif open(sa, "rb").read() != open(sb_, "rb").read():
                    problems.append("SAM differs")
            finally:
                ix.close()
                oix.close()
    desc = "seed %d: %d bytes of FASTA, interval %d, %s, %d reads | build %.2f s, oracle %.2f s" % (
        seed, len(fasta), interval, "duplication beyond sort_len" if has_dup else "files compared", n, t_gpu, t_cpu)
    if verbose:
        print(desc, "|", "ok" if not problems else "MISMATCH", flush=True)
        for p in problems:
            print("    " + p, flush=True)
    return problems, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:20")
    ap.add_argument("--seconds", type=float, default=1e9)
    a = ap.parse_args()
    lo_, hi_ = a.seeds.split(":")
    import oracle_lib
    import sbwt_amd as sb
    oracle_lib.build()
    sb.device_count()
    t0 = time.time()
    bad = []
    done = 0
    for s in range(int(lo_), int(hi_)):
        if time.time() - t0 > a.seconds:
            break
        problems, _ = run_seed(s, sb, oracle_lib)
        done += 1
        if problems:
            bad.append(s)
    print("fuzz_builder: %d seeds, %d with mismatches %s, %.0f s" % (done, len(bad), bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
