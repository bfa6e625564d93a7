#!/usr/bin/env python3
"""FASTQ -> SAM rate of sbwt_map_fastq with its per-stage wall times (SBWT_TIMING), over batch sizes.

usage: fastq_rate.py [--scale 0.05] [--reads 8000000] [--threads 16] [--batches 1048576,262144]   (needs an MI355X)
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--reads", type=int, default=8_000_000)
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--batches", default="1048576,262144")
    args = ap.parse_args()
    import torch
    import bench
    import sbwt_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = args.read_len
    lens_chr, g, chr_start, npos_key, npos_val = bench.synth_tables(args.scale)
    codes, packed_genome = bench.make_genome(torch, dev, g, seed=38)
    names = ["chr%d" % (k + 1) for k in range(len(chr_start))]
    ix, _ = sbwt_amd.build_index_device(packed_genome.data_ptr(), g, chr_start, npos_key, npos_val, device=0, chr_names=names)
    wpr = sbwt_amd.words_per_read(L)
    n = args.reads
    packed, lens = bench.make_reads(torch, dev, codes, g, n, L, seed=50, wpr=wpr)
    h_packed = packed.cpu().numpy().view(np.uint32).reshape(n, wpr)
    h_lens = lens.cpu().numpy().astype(np.uint32)
    tmp = tempfile.mkdtemp(prefix="sbwt_rate_")
    fq, sam = os.path.join(tmp, "r.fq"), os.path.join(tmp, "o.sam")
    size = bench.write_fastq(fq, h_packed, h_lens, L)
    os.environ["SBWT_TIMING"] = "1"
    print("output on", os.popen("stat -f -c %%T %s" % tmp).read().strip(), flush=True)
    for b, wr in [(b, wr) for b in args.batches.split(",") for wr in ("one", "pool")]:
        os.environ["SBWT_FASTQ_BATCH"] = b
        os.environ["SBWT_SAM_WRITE"] = wr
        print("--- write through", wr, flush=True)
        for rep in range(2):
            if os.path.exists(sam):
                os.unlink(sam)                     # truncating 1 GB of page cache is not part of the run
            t0 = time.perf_counter()
            st = ix.map_fastq(fq, sam, n_threads=args.threads)
            dt = time.perf_counter() - t0
            print("batch %s, %d threads: %.3f s = %.2f M reads/s (FASTQ %.0f MB, SAM %.0f MB, %d records)"
                  % (b, args.threads, dt, n / dt / 1e6, size / 1e6, os.path.getsize(sam) / 1e6, st[4]), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)
    ix.close()


if __name__ == "__main__":
    main()
