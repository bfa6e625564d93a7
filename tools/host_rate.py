#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-facing entry points (never bench.py's `value`, which has its inputs in HBM).

  1. sbwt_map_batch: packed reads in host memory -> results + hits in host memory (H2D, kernels, D2H).
  2. sbwt_map_fastq: FASTQ file -> SAM file (parse, pack, map, format, write), n_threads host threads.

usage: host_rate.py [--scale 0.05] [--reads 4000000] [--threads 16]     (needs an MI355X)
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--threads", type=int, default=16)
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
    ix, _ = sbwt_amd.build_index_device(packed_genome.data_ptr(), g, chr_start, npos_key, npos_val, device=0,
                                        chr_names=names)
    wpr = sbwt_amd.words_per_read(L)
    n = args.reads
    packed, lens = bench.make_reads(torch, dev, codes, g, n, L, seed=50, wpr=wpr)
    h_packed = packed.cpu().numpy().view(np.uint32)
    h_lens = lens.cpu().numpy().astype(np.uint32)
    ctx = sbwt_amd.Context(ix, n)
    ctx.map_batch(h_packed, h_lens)                      # warm-up (staging buffers)
    t0 = time.perf_counter()
    res, hits = ctx.map_batch(h_packed, h_lens)
    dt = time.perf_counter() - t0
    print("sbwt_map_batch (host buffers, %d reads x %d bp): %.3f s = %.1f M reads/s; device part %.2f ms"
          % (n, L, dt, n / dt / 1e6, ctx.last_timing()[0]))
    # the same with caller-owned buffers that are reused (pageable), then page-locked (sbwt_host_alloc)
    from sbwt_amd import api
    res_p, hits_p = np.zeros(n, api.RESULT_DTYPE), np.zeros(2 * n, api.HIT_DTYPE)
    res_p[:] = res_p; hits_p[:] = hits_p                 # touch the pages
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.map_batch(h_packed, h_lens, out=(res_p, hits_p))
        dt = time.perf_counter() - t0
    print("  reused pageable buffers: %.3f s = %.1f M reads/s" % (dt, n / dt / 1e6))
    t0 = time.perf_counter()
    pin = api.PinnedBuffer(h_packed.nbytes + h_lens.nbytes + 32 * n + 16 * 2 * n)
    t_pin = time.perf_counter() - t0
    o = 0
    pk = pin.array(np.uint32, h_packed.size, o); o += h_packed.nbytes
    ln = pin.array(np.uint32, n, o); o += h_lens.nbytes
    rs = pin.array(api.RESULT_DTYPE, n, o); o += 32 * n
    ht = pin.array(api.HIT_DTYPE, 2 * n, o)
    pk[:] = h_packed.reshape(-1); ln[:] = h_lens
    for _ in range(2):
        t0 = time.perf_counter()
        r2, h2 = ctx.map_batch(pk.reshape(n, wpr), ln, out=(rs, ht))
        dt = time.perf_counter() - t0
    same = bool(np.array_equal(r2, res) and np.array_equal(h2, hits))
    print("  page-locked buffers (%.0f MB, allocated in %.3f s): %.4f s = %.1f M reads/s; identical results: %s"
          % (pin.nbytes / 1e6, t_pin, dt, n / dt / 1e6, same))
    del pk, ln, rs, ht, r2, h2
    pin.close()
    # FASTQ -> SAM
    tmp = tempfile.mkdtemp(prefix="sbwt_rate_")
    fq = os.path.join(tmp, "r.fq")
    seqs, offsets = bench.unpack_reads(h_packed, h_lens)
    seqs = seqs.reshape(n, L)
    qual = np.full((n, L), ord("I"), np.uint8)
    with open(fq, "wb") as f:
        step = 200000
        for a in range(0, n, step):
            b = min(n, a + step)
            rows = []
            for k in range(a, b):
                rows.append(b"@r%d\n" % k + seqs[k].tobytes() + b"\n+\n" + qual[k].tobytes() + b"\n")
            f.write(b"".join(rows))
    size = os.path.getsize(fq)
    for nt in (1, args.threads):
        sam = os.path.join(tmp, "o%d.sam" % nt)
        t0 = time.perf_counter()
        st = ix.map_fastq(fq, sam, n_threads=nt)
        dt = time.perf_counter() - t0
        print("sbwt_map_fastq -n %d: %.2f s = %.2f M reads/s (FASTQ %.0f MB in, SAM %.0f MB out, %d records)"
              % (nt, dt, n / dt / 1e6, size / 1e6, os.path.getsize(sam) / 1e6, st[4]))
    ctx.close()
    ix.close()


if __name__ == "__main__":
    main()
