#!/usr/bin/env python3
"""bench.py -- headline benchmark: mapped reads/s, 50-bp exact reads against a GRCh38-sized index.

Workload (BASELINE.json configs[2]): synthetic genome of 2.94 G non-N bases over 24 contigs with
GRCh38-like lengths and N runs, 100 M error-free 50-bp reads, both strands, resident in HBM.
One step = one pass of the hot path (sbwt_map_batch_device) over the whole read batch.
N > 1: one process per GPU, index replicated, each rank maps its own 100 M reads (weak scaling, no
data-path collective; torch.distributed/RCCL only for the barrier and the max-over-ranks).

Prints ONE JSON line on rank 0.  torch is plumbing here (device memory for the synthetic inputs,
barrier); the timed work is the library's HIP kernels behind the C ABI.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# non-N lengths (Mbp) of the 24 GRCh38 primary chromosomes, rounded; sum = 2.94 Gbp
CHR_MBP = [230.5, 240.5, 198.1, 189.8, 181.3, 170.1, 159.0, 144.8, 121.8, 133.3, 134.5, 133.1,
           97.98, 88.52, 84.64, 81.81, 82.92, 80.09, 58.44, 63.94, 40.09, 39.16, 154.9, 26.42]


def synth_tables(scale):
    """chromosome starts and the N-offset map (chr_umbiguous_starting_length) of the synthetic genome:
    every contig starts with a 10 kb N run and has a 50 kb N gap every 40 Mbp."""
    lens = [max(int(x * 1e6 * scale), 1000) for x in CHR_MBP]
    chr_start, npos_key, npos_val = [], [], []
    acc = 0
    for ln in lens:
        chr_start.append(acc)
        off = 10000
        npos_key.append(acc)
        npos_val.append(off)
        p = 40_000_000
        while p < ln:
            off += 50000
            npos_key.append(acc + p)
            npos_val.append(off)
            p += 40_000_000
        acc += ln
    npos_key.append(acc)
    npos_val.append(0)
    return lens, acc, chr_start, npos_key, npos_val


def make_genome(torch, dev, g, seed):
    """g uniform bases -> (uint8 codes [g], packed int32 words [ceil(g/16)+1])"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    codes = torch.empty(g, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for a in range(0, g, step):
        b = min(g, a + step)
        codes[a:b] = torch.randint(0, 4, (b - a,), dtype=torch.uint8, device=dev, generator=gen)
    return codes, repack(torch, dev, codes, g)


def repack(torch, dev, codes, g):
    words = (g + 15) // 16
    packed = torch.zeros(words + 1, dtype=torch.int32, device=dev)
    shifts = (2 * torch.arange(16, device=dev, dtype=torch.int32))
    wstep = 1 << 24
    for a in range(0, words, wstep):
        b = min(words, a + wstep)
        chunk = torch.zeros((b - a) * 16, dtype=torch.int32, device=dev)
        hi = min(g, b * 16)
        chunk[: hi - a * 16] = codes[a * 16: hi].to(torch.int32)
        packed[a:b] = (chunk.view(-1, 16) << shifts).sum(dim=1, dtype=torch.int32)
    return packed


def plant_repeats(torch, dev, codes, g, seed, n_families=2000, max_copies=5000):
    """config 5: n_families distinct 30-mers, family f planted copies[f] times (2..max_copies, log-uniform) at
    random 64-aligned slots; returns the slot starts (device tensor) so reads can be drawn from them"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    u = torch.rand(n_families, device=dev, generator=gen)
    copies = torch.exp(u * float(np.log(max_copies / 2.0))).mul(2.0).long().clamp(2, max_copies)
    total = int(copies.sum().item())
    slots = torch.randperm(g // 64 - 2, device=dev, generator=gen)[:total] * 64 + 7
    fam = torch.randint(0, 4, (n_families, 30), dtype=torch.uint8, device=dev, generator=gen)
    owner = torch.repeat_interleave(torch.arange(n_families, device=dev), copies)
    idx = slots[:, None] + torch.arange(30, device=dev)[None, :]
    codes[idx.reshape(-1)] = fam[owner].reshape(-1)
    return slots


def make_reads(torch, dev, codes, g, n_reads, L, seed, wpr, min_len=None, slots=None, mutate=0.0, origins=None):
    """n_reads error-free reads, strand uniform -> packed int32 [n_reads, wpr], lens int32.  Length L, or uniform
    in min_len..L; with `slots`, half of the reads start inside a planted repeat"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    packed = torch.zeros((n_reads, wpr), dtype=torch.int32, device=dev)
    lens = torch.full((n_reads,), L, dtype=torch.int32, device=dev)
    shifts = (2 * torch.arange(16, device=dev, dtype=torch.int32))
    ar = torch.arange(L, device=dev, dtype=torch.int64)
    step = 1 << 22
    for a in range(0, n_reads, step):
        b = min(n_reads, a + step)
        pos = torch.randint(0, g - L, (b - a,), device=dev, generator=gen, dtype=torch.int64)
        if slots is not None:
            pick = slots[torch.randint(0, slots.numel(), (b - a,), device=dev, generator=gen)]
            use = torch.randint(0, 2, (b - a,), device=dev, generator=gen).bool()
            pos = torch.where(use, pick, pos)
        rev = torch.randint(0, 2, (b - a,), device=dev, generator=gen, dtype=torch.int64).bool()
        if origins is not None:
            origins.append((pos, rev))
        bases = codes[(pos[:, None] + ar[None, :])].to(torch.int32)          # forward strand
        rc = 3 - bases.flip(1)
        bases = torch.where(rev[:, None], rc, bases)
        if mutate > 0.0:
            # one substitution in a fraction of the reads (they die somewhere along the search)
            hit = torch.rand(b - a, device=dev, generator=gen) < mutate
            where = torch.randint(0, L, (b - a,), device=dev, generator=gen)
            delta = torch.randint(1, 4, (b - a,), device=dev, generator=gen, dtype=torch.int32)
            rows = torch.nonzero(hit).squeeze(1)
            bases[rows, where[rows]] = (bases[rows, where[rows]] + delta[rows]) & 3
        buf = torch.zeros((b - a, wpr * 16), dtype=torch.int32, device=dev)
        buf[:, :L] = bases
        if min_len is not None:
            ln = torch.randint(min_len, L + 1, (b - a,), device=dev, generator=gen, dtype=torch.int64)
            lens[a:b] = ln.to(torch.int32)
            # bases past a read's length are zero in its record, as sbwt_pack_reads leaves them
            buf *= (torch.arange(wpr * 16, device=dev)[None, :] < ln[:, None]).to(torch.int32)
        packed[a:b] = (buf.view(b - a, wpr, 16) << shifts).sum(dim=2, dtype=torch.int32)
    return packed, lens


def origin_check(torch, dev, results, hits, origins, L, chr_start, npos_key, npos_val):
    """Full-size property of the error-free workload (no oracle needed): a read cut from forward offset `pos` must
    come back with exactly that origin -- chromosome, 0-based position with the N offset added, strand flag
    (searcher.hpp:504-539: a forward read is found in the reverse-complement half, isRC) -- at hits[its own
    index], unless its first and last base lie in different chromosomes or N-free stretches (then NH = 0)."""
    cs = torch.tensor(chr_start, dtype=torch.int64, device=dev)
    nk = torch.tensor(npos_key, dtype=torch.int64, device=dev)
    nv = torch.tensor(npos_val, dtype=torch.int64, device=dev)
    bounds = torch.unique(torch.cat([cs, nk]))
    res = results.view(torch.int64).view(-1, 4)
    hit = hits.view(torch.int64).view(-1, 2)
    bad = 0
    spanning = 0
    a = 0
    for pos, rev in origins:
        b = a + pos.numel()
        seg0 = torch.searchsorted(bounds, pos, right=True)
        seg1 = torch.searchsorted(bounds, pos + (L - 1), right=True)
        span = seg0 != seg1
        c = torch.searchsorted(cs, pos, right=True) - 1
        k = torch.searchsorted(nk, pos, right=True) - 1
        pos0 = pos - cs[c] + nv[k]
        word = c | ((~rev).to(torch.int64) << 32)          # chr_id | is_rc << 32
        r = res[a:b]
        nh = r[:, 3] & 0xFFFFFFFF
        idx = torch.arange(a, b, device=dev, dtype=torch.int64)
        h = hit[a:b]
        ok_hit = (nh == 1) & (r[:, 1] - r[:, 0] == 1) & (r[:, 2] == idx) & (h[:, 0] == pos0) & (h[:, 1] == word)
        ok = torch.where(span, nh == 0, ok_hit)
        bad += int((~ok).sum())
        spanning += int(span.sum())
        a = b
    return {"reads": a, "mismatches": bad, "boundary_spanning_reads": spanning}


def unpack_reads(packed_np, lens_np):
    """packed uint32 [n, wpr], lens [n] -> (ASCII bytes of all reads back to back, offsets uint64 [n + 1])"""
    Lmax = int(lens_np.max())
    j = np.arange(Lmax)
    codes = (packed_np[:, j >> 4] >> (2 * (j & 15)).astype(np.uint32)) & 3
    keep = j[None, :] < lens_np[:, None]
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[codes][keep]
    offsets = np.concatenate([[0], np.cumsum(lens_np.astype(np.uint64))]).astype(np.uint64)
    return seqs, offsets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0, help="genome size as a fraction of GRCh38 (debug)")
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--min-len", type=int, default=None, help="reads of length min-len..read-len (config 5)")
    ap.add_argument("--mutate", type=float, default=0.0, help="fraction of reads with one substitution")
    ap.add_argument("--repeats", action="store_true", help="plant a repeat family, half of the reads inside it (config 5)")
    args = ap.parse_args()

    import torch
    import sbwt_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the library has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    t0 = time.time()
    L = args.read_len
    lens_chr, g, chr_start, npos_key, npos_val = synth_tables(args.scale)
    codes, packed_genome = make_genome(torch, dev, g, seed=38)
    slots = None
    if args.repeats:
        slots = plant_repeats(torch, dev, codes, g, seed=5)
        packed_genome = repack(torch, dev, codes, g)
    torch.cuda.synchronize()
    t_genome = time.time() - t0

    # index: suffix order + tables on the GPU, straight into the HBM layout; rank 0 also exports the
    # reference-format tables for the CPU baseline leg
    # the CPU baseline and the oracle spot check run on rank 0 of a single-GPU run only
    run_cpu = (world == 1 and not args.no_cpu)
    want_tables = run_cpu
    t0 = time.time()
    ix, tables = sbwt_amd.build_index_device(packed_genome.data_ptr(), g, chr_start, npos_key, npos_val, interval=64,
                                             device=local, want_index=True, want_tables=want_tables)
    t_index = time.time() - t0
    del packed_genome
    info = ix.info()

    wpr = sbwt_amd.words_per_read(L)
    n_reads = args.reads
    t0 = time.time()
    plain = args.min_len is None and slots is None and args.mutate == 0.0
    origins = [] if (plain and rank == 0) else None
    packed, lens = make_reads(torch, dev, codes, g, n_reads, L, seed=50 + rank, wpr=wpr, min_len=args.min_len, slots=slots,
                              mutate=args.mutate, origins=origins)
    del codes
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_reads = time.time() - t0

    res_bytes = n_reads * 32
    results = torch.empty(res_bytes, dtype=torch.uint8, device=dev)
    hits_cap = n_reads + (1 << 20) if not args.repeats else 80 * n_reads
    hits = torch.empty(hits_cap * 16, dtype=torch.uint8, device=dev)
    ctx = sbwt_amd.Context(ix, n_reads)

    def step():
        return ctx.map_batch_device(packed.data_ptr(), lens.data_ptr(), n_reads, wpr, results.data_ptr(),
                                    hits.data_ptr(), hits_cap, limit=1000, strand=2)

    from sbwt_amd import shard
    kernel_ms = []
    last = {"hits": 0}

    def timed_step():
        last["hits"] = step()                # synchronises its own stream before returning
        kernel_ms.append(ctx.last_timing())

    for _ in range(args.warmup):
        step()
    elapsed = shard.timed_steps(timed_step, args.steps, 0, dist=dist, sync=torch.cuda.synchronize, device=dev)
    n_hits = last["hits"]

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    origin = None
    if origins is not None:
        origin = origin_check(torch, dev, results, hits, origins, L, chr_start, npos_key, npos_val)
        del origins
    value = world * n_reads * args.steps / elapsed
    search_ms = float(np.mean([k[1] for k in kernel_ms]))
    multi_ms = float(np.mean([k[2] for k in kernel_ms]))
    res_np = np.frombuffer(results.cpu().numpy().tobytes(), dtype=sbwt_amd.api.RESULT_DTYPE)
    mapped = int((res_np["nh"] > 0).sum())
    multi = int(((res_np["sa_hi"] - res_np["sa_lo"]) > 1).sum())
    # Algorithmic bytes of ONE k_search launch: the search is a chain of random accesses
    # into index tables far larger than any cache, and HBM serves such an access as one 128-byte line whatever
    # part of it is used, so the unit is the line: 128 B x (index lines the reads of the batch need, counted by
    # the kernel itself: search-prefix entry, two-base blocks, memoised locate, text comparison, final row)
    # + the streamed bytes (read record + length in, result + hit out).
    stats = ctx.last_stats()
    accesses = int(stats[0])
    streamed = n_reads * (wpr * 4 + 4 + 32) + int(n_hits) * 16
    algo_bytes = 128 * accesses + streamed
    achieved = algo_bytes / (search_ms * 1e-3) / 1e9
    survey_per_read = 16 + 64 * (2 * (L - 12) + 32.5)   # reference-layout bytes per unique read
    traffic = None
    requests = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # from the PMC passes of tools/profile_final.sh
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("reads") == n_reads and t.get("read_len") == L and abs(t.get("scale", 1.0) - args.scale) < 1e-9:
            traffic = t.get("traffic_GB_per_launch")
            if t.get("TCC_EA0_RDREQ") and t.get("TCC_EA0_WRREQ"):
                requests = t["TCC_EA0_RDREQ"] + t["TCC_EA0_WRREQ"]
    roofline = {"bound": "hbm", "kernel": "k_search_sm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_unit": "GB per launch (PMC, profiles/traffic.json)",
                "launch_ms": round(search_ms, 3), "algorithmic_GB_per_launch": round(algo_bytes / 1e9, 2),
                "index_lines_per_read": round(accesses / n_reads, 3), "bytes_per_read": round(algo_bytes / n_reads, 1),
                "index_lines_per_s_G": round(accesses / (search_ms * 1e-3) / 1e9, 2),
                "gather_roofline_lines_per_s_G": 47.0,
                # HBM requests per launch (PMC, profiles/traffic.json) over this run's launch time, against the
                # measured random-access roofline of 47 G requests/s (profiles/r1/gather_probe*.txt)
                "hbm_requests_per_s_G": round(requests / (search_ms * 1e-3) / 1e9, 2) if requests else None,
                "gather_roofline_frac": round(requests / (search_ms * 1e-3) / 47e9, 3) if requests else None,
                "survey_formula_GBps": round(survey_per_read * n_reads / (search_ms * 1e-3) / 1e9, 1)}

    cpu = None
    parity = None
    if run_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        T = tables.contents
        L_ = oracle_lib.lib()
        c4 = (C.c_uint64 * 4)(*T.c_function4)
        cs = np.asarray(chr_start, dtype=np.uint64)
        nk = np.asarray(npos_key, dtype=np.uint64)
        nv = np.asarray(npos_val, dtype=np.uint64)
        h = L_.sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions, T.occ[0],
                              T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt, T.location_table, T.n_loc,
                              cs.size, cs.ctypes.data, nk.size, nk.ctypes.data, nv.ctypes.data)
        oix = oracle_lib.OracleIndex(h)
        cores = min(16, os.cpu_count() or 1)
        probe_n = 20000
        sample_max = 8_000_000
        host_packed = packed[:sample_max].cpu().numpy().view(np.uint32)
        host_lens = lens[:sample_max].cpu().numpy().astype(np.uint32)
        pseq, poff = unpack_reads(host_packed[:probe_n], host_lens[:probe_n])
        t1 = time.perf_counter()
        lo, hi, nh, first, ohits = oix.map_batch_raw(np.concatenate([pseq, np.zeros(1, np.uint8)]), poff, probe_n, nthreads=1)
        rate1 = probe_n / (time.perf_counter() - t1)
        # parity of the same reads through the GPU path
        gres = res_np[:probe_n]
        ok = bool(np.array_equal(gres["sa_lo"], lo) and np.array_equal(gres["sa_hi"], hi) and np.array_equal(gres["nh"], nh))
        if ok and int(nh.sum()):
            # ordered hit lists of the probe reads, gathered on the device (the lists of multi-hit reads lie anywhere)
            tot = int(nh.sum())
            within = np.arange(tot) - np.repeat(np.cumsum(nh) - nh, nh)
            gi = torch.from_numpy((np.repeat(gres["first_hit"], nh) + within).astype(np.int64)).to(dev)
            ghits = np.frombuffer(hits.view(-1, 16)[gi].cpu().numpy().tobytes(), dtype=sbwt_amd.api.HIT_DTYPE)
            oi = (np.repeat(first, nh) + within).astype(np.int64)
            ok = bool(np.array_equal(ghits, ohits[oi]))
        parity = {"reads_checked": probe_n, "identical": ok}
        sample_n = int(min(sample_max, max(probe_n, rate1 * cores * args.cpu_seconds * 0.7)))
        seqs, offsets = unpack_reads(host_packed[:sample_n], host_lens[:sample_n])
        t1 = time.perf_counter()
        oix.map_batch_raw(np.concatenate([seqs, np.zeros(1, np.uint8)]), offsets, sample_n, nthreads=cores)
        dt = time.perf_counter() - t1
        cpu = {"value": round(sample_n / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": "first %d reads of the same batch, same index (reference-format tables), %d threads; "
                         "single thread: %.0f reads/s on %d reads" % (sample_n, cores, rate1, probe_n)}
        oix.close()
        sbwt_amd.host_tables_free(tables)

    line = {
        "metric": "mapped reads/sec (50bp exact, GRCh38) at 1/8 GPUs; achieved HBM GB/s",
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "GRCh38-sized synthetic genome (%.3f Gbp non-N, 24 contigs, N runs%s), %d x %s reads per GPU, "
                               "both strands, -m 1000 -s 2, inputs resident in HBM"
                               % (g / 1e9, ", planted repeat family" if args.repeats else "", n_reads,
                                  ("%d-bp" % L if args.min_len is None else "%d..%d-nt" % (args.min_len, L))
                                  + (" (%.0f%% with one substitution)" % (100 * args.mutate) if args.mutate else " exact")),
                   "index_suffixes": int(info.n), "index_hbm_GB": round(info.device_bytes / 1e9, 2),
                   "reads_per_gpu": n_reads, "read_len": L, "parallelism": "replicated index, reads sharded x%d" % world},
        "roofline": roofline, "cpu_baseline": cpu,
        "detail": {"mapped_reads": mapped, "multi_hit_reads": multi, "hits": int(n_hits),
                   "k_search_ms": round(search_ms, 3), "k_multi_ms": round(multi_ms, 3),
                   "setup_s": {"genome": round(t_genome, 1), "index_build": round(t_index, 1), "reads": round(t_reads, 1)},
                   "parity_vs_oracle": parity, "origin_check": origin},
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
