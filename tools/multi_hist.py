#!/usr/bin/env python3
"""Histogram of the raw interval sizes of the multi-hit reads of a bench workload (who the locate kernels work for).
usage (GPU box): python tools/multi_hist.py [--shuffle-ties]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import sbwt_amd  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = bench.Workload(torch, dev, sbwt_amd, scale=1.0, genome="repeats", shuffle_ties="--shuffle-ties" in sys.argv)
    n, L = 20_000_000, 50
    wpr = sbwt_amd.words_per_read(L)
    packed, lens = bench.make_reads(torch, dev, wl.codes, wl.g, n, L, seed=50, wpr=wpr, slots=wl.slots)
    torch.cuda.synchronize()
    run = bench.Run(torch, dev, sbwt_amd, wl, packed, lens, n, L, wpr, "compact")
    run.step()
    torch.cuda.synchronize()
    rec = run.results.view(torch.int64).view(-1, 2)
    w0, w1 = rec[:, 0], rec[:, 1]
    meta = (w1 >> 56) & 0xFF
    width = (w0 >> 40) & 0xFFFFFF
    multi = width > 1
    ws = width[multi]
    print("reads %d, raw interval > 1: %d, rows %d" % (n, int(multi.sum()), int(ws.sum())))
    edges = [2, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 1 << 24]
    for a, b in zip(edges[:-1], edges[1:]):
        sel = (ws >= a) & (ws < b)
        print("  %6d .. %-8d reads %8d  rows %10d" % (a, b - 1, int(sel.sum()), int(ws[sel].sum())))


if __name__ == "__main__":
    main()
