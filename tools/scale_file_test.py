#!/usr/bin/env python3
"""The drop-in path at full size, through FILES and the command line only:

  synthetic FASTA (GRCh38-like contigs with N runs)  ->  `sbwt build -i g.fa -p idx`  ->  index files on disk
  ->  `sbwt map -i reads.fq -p idx -o out.sam -n 16`  ->  SAM compared with the oracle, which loads the SAME files.

Exercises what bench.py does not: the FASTA parser, the > 4 GB index file written in the upstream archive layout,
its loader, and the FASTQ/SAM pipeline, with more than 2^32 suffixes at --scale 1.  Files go to --dir (default
/dev/shm/sbwt_scale, removed afterwards).   usage: scale_file_test.py [--scale 1.0] [--reads 2000000] [--check 50000]
"""
import argparse
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHR_MBP = [230.5, 240.5, 198.1, 189.8, 181.3, 170.1, 159.0, 144.8, 121.8, 133.3, 134.5, 133.1,
           97.98, 88.52, 84.64, 81.81, 82.92, 80.09, 58.44, 63.94, 40.09, 39.16, 154.9, 26.42]
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--check", type=int, default=50_000)
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--dir", default="/dev/shm/sbwt_scale")
    args = ap.parse_args()
    exe = os.path.join(ROOT, "sbwt_amd", "bin", "sbwt")
    d = args.dir
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    try:
        run(args, exe, d)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run(args, exe, d):
    rng = np.random.default_rng(38)
    L = args.read_len
    fa = os.path.join(d, "g.fa")
    t0 = time.time()
    sample = []          # (contig sequence slice source) for read sampling: keep a few contigs in memory
    with open(fa, "wb") as f:
        for k, mbp in enumerate(CHR_MBP):
            n = max(int(mbp * 1e6 * args.scale), 2000)
            seq = ACGT[rng.integers(0, 4, n, dtype=np.uint8)]
            if k < 3:
                sample.append(seq.copy())
            # a 10 kb N run in front, a 50 kb N gap every 40 Mbp
            parts = [np.full(10000, ord("N"), np.uint8)]
            for a in range(0, n, 40_000_000):
                if a:
                    parts.append(np.full(50000, ord("N"), np.uint8))
                parts.append(seq[a:a + 40_000_000])
            full = np.concatenate(parts)
            f.write(b">chr%d\n" % (k + 1))
            rows = len(full) // 60
            body = np.empty((rows, 61), np.uint8)
            body[:, :60] = full[:rows * 60].reshape(rows, 60)
            body[:, 60] = 10
            f.write(body.tobytes())
            if len(full) > rows * 60:
                f.write(full[rows * 60:].tobytes() + b"\n")
    print("FASTA %.2f GB written in %.0f s" % (os.path.getsize(fa) / 1e9, time.time() - t0), flush=True)
    prefix = os.path.join(d, "idx")
    t0 = time.time()
    r = subprocess.run([exe, "build", "-i", fa, "-p", prefix], capture_output=True)
    print("sbwt build: rc %d, %.0f s; t_table.bwt %.2f GB" % (r.returncode, time.time() - t0,
          os.path.getsize(prefix + ".t_table.bwt") / 1e9 if os.path.exists(prefix + ".t_table.bwt") else -1), flush=True)
    if r.returncode:
        print(r.stderr.decode()[-2000:])
        raise SystemExit(1)
    os.remove(fa)
    # reads from the first three contigs, both strands, a tenth with one substitution
    fq = os.path.join(d, "r.fq")
    with open(fq, "wb") as f:
        rows = []
        for i in range(args.reads):
            s = sample[i % 3]
            p = int(rng.integers(0, len(s) - L))
            rd = s[p:p + L].tobytes()
            if rng.random() < 0.5:
                rd = rd.translate(COMP)[::-1]
            if rng.random() < 0.1:
                q = int(rng.integers(0, L))
                rd = rd[:q] + b"ACGT"[int(rng.integers(0, 4))].to_bytes(1, "little") + rd[q + 1:]
            rows.append(b"@r%d\n%s\n+\n%s\n" % (i, rd, b"I" * L))
            if len(rows) == 200000:
                f.write(b"".join(rows))
                rows = []
        f.write(b"".join(rows))
    sam = os.path.join(d, "o.sam")
    t0 = time.time()
    r = subprocess.run([exe, "map", "-i", fq, "-p", prefix, "-o", sam, "-n", str(args.threads)], capture_output=True)
    dt = time.time() - t0
    print("sbwt map -n %d: rc %d, %.1f s for %d reads (index load included)\n%s" % (args.threads, r.returncode, dt, args.reads,
          r.stderr.decode()[-3000:]), flush=True)
    if r.returncode:
        raise SystemExit(1)
    # the oracle on the same index files, first --check reads
    import oracle_lib
    t0 = time.time()
    oix = oracle_lib.OracleIndex.load(prefix)
    print("oracle index load %.0f s" % (time.time() - t0), flush=True)
    head = os.path.join(d, "head.fq")
    with open(fq, "rb") as f, open(head, "wb") as g:
        for _ in range(4 * args.check):
            g.write(f.readline())
    osam = os.path.join(d, "head.sam")
    rc, st = oix.map_fastq(head, osam)
    oix.close()
    want = open(osam, "rb").read()
    got = open(sam, "rb").read(len(want))
    ok = rc == 0 and want == got and len(want) > 0
    print("oracle on the first %d reads: %d SAM bytes, identical prefix of the GPU output: %s" % (args.check, len(want), ok),
          flush=True)
    if not ok:
        raise SystemExit(2)


if __name__ == "__main__":
    main()
