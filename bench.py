#!/usr/bin/env python3
"""bench.py -- headline benchmark: mapped reads/s, 50-bp exact reads against a GRCh38-sized index.

Workload (BASELINE.json configs[2]): synthetic genome of 2.94 G non-N bases over 24 contigs with
GRCh38-like lengths and N runs, 100 M error-free 50-bp reads, both strands, resident in HBM.
One step = one pass of the hot path (sbwt_map_batch_device) over the whole read batch.
N > 1: one process per GPU, index replicated, each rank maps its own 100 M reads (weak scaling, no
data-path collective; torch.distributed/RCCL only for the barrier and the max-over-ranks).  `--gpus N` without
a launcher (no WORLD_SIZE in the environment) starts the N ranks itself as child processes, before this
process touches a GPU; under torchrun each process is one rank.

Other workloads (parity-checked side lines, never the headline): --rank-path (no memoised search prefix:
SBWT_HASH_ORDER=0 SBWT_KMER_ORDER=12, the backward search through rank queries that north_star names),
--genome repeats (40 % of the bases in diverged repeat families + exact segmental duplications),
--shuffle-ties (suffix order stopped at sort_len 256 with ties in arbitrary order, as upstream's sort leaves them).

Prints ONE JSON line on rank 0.  torch is plumbing here (device memory for the synthetic inputs,
barrier); the timed work is the library's HIP kernels behind the C ABI.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
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


def plant_repeat_families(torch, dev, codes, g, seed, frac=0.40):
    """--genome repeats: `frac` of the bases are overwritten by copies of repeat families, every copy carrying its own
    substitutions (interspersed-repeat-like: 150 families, consensus 300 .. 6,000 bp, divergence 1 .. 15 % per
    family), plus 40 exact duplications of 2 .. 20 kb in 2 .. 6 copies (ties beyond sort_len for the suffix order).
    Returns the fraction of bases covered."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    rng = np.random.default_rng(seed)
    covered = 0
    fams = []
    for f in range(150):
        fams.append((int(rng.integers(300, 6001)), float(rng.choice([0.01, 0.02, 0.05, 0.10, 0.15])), 1.0))
    for f in range(40):
        fams.append((int(rng.integers(2000, 20001)), 0.0, 0.0))
    budget = int(g * frac)
    share = budget // 150
    for (ln, div, inter) in fams:
        copies = max(2, share // ln) if inter else int(rng.integers(2, 7))
        cons = torch.randint(0, 4, (ln,), dtype=torch.uint8, device=dev, generator=gen)
        step = max(1, (1 << 26) // ln)
        for a in range(0, copies, step):
            k = min(step, copies - a)
            starts = torch.randint(0, g - ln, (k,), device=dev, generator=gen, dtype=torch.int64)
            block = cons[None, :].expand(k, ln).clone()
            if div > 0:
                mut = torch.rand((k, ln), device=dev, generator=gen) < div
                delta = torch.randint(1, 4, (k, ln), device=dev, generator=gen, dtype=torch.uint8)
                block = torch.where(mut, (block + delta) & 3, block)
            rev = torch.randint(0, 2, (k,), device=dev, generator=gen).bool()
            block = torch.where(rev[:, None], 3 - block.flip(1), block)
            idx = starts[:, None] + torch.arange(ln, device=dev)[None, :]
            codes[idx.reshape(-1)] = block.reshape(-1)
        covered += copies * ln
    return covered / g


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


def spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: N fresh child processes, one rank per GPU, started before this process
    touches a GPU (it never does); rank 0's JSON line is passed through."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()           # counts devices without initialising the GPU (see the task's notes)
    if have < n and not os.environ.get("BENCH_SHARE_DEVICE"):
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out.decode().splitlines():     # rank 0's JSON line; anything a backend printed beside it goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if any(codes):
        raise SystemExit("bench.py: rank exit codes %s" % codes)


def write_fastq(path, host_packed, host_lens, L):
    """FASTQ text of fixed-length reads, vectorised: @r0000000 / bases / + / IIII"""
    n = host_lens.size
    seqs, _ = unpack_reads(host_packed, host_lens)
    seqs = seqs.reshape(n, L)
    name_w = 2 + 8
    rec = np.empty((n, name_w + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(8):
        rec[:, 2 + 7 - d] = ord("0") + (idx // 10 ** d) % 10
    o = name_w
    rec[:, o] = 10; o += 1
    rec[:, o:o + L] = seqs; o += L
    rec[:, o] = 10; rec[:, o + 1] = ord("+"); rec[:, o + 2] = 10; o += 3
    rec[:, o:o + L] = ord("I"); o += L
    rec[:, o] = 10
    rec.tofile(path)
    return rec.nbytes


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_index_from_tables(oracle_lib, tables, chr_start, npos_key, npos_val):
    T = tables.contents
    c4 = (C.c_uint64 * 4)(*T.c_function4)
    cs = np.asarray(chr_start, dtype=np.uint64)
    nk = np.asarray(npos_key, dtype=np.uint64)
    nv = np.asarray(npos_val, dtype=np.uint64)
    h = oracle_lib.lib().sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions,
                                        T.occ[0], T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt,
                                        T.location_table, T.n_loc, cs.size, cs.ctypes.data, nk.size, nk.ctypes.data,
                                        nv.ctypes.data)
    return oracle_lib.OracleIndex(h, keep=(c4, cs, nk, nv))


def cpu_config1(torch, dev, sbwt_amd, oracle_lib, cores):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): 4,641,652-bp genome, 10^6 x 50-bp reads,
    the oracle with 1 thread and with all cores (what `sbwt map -n 1` / `-n <cores>` time upstream)."""
    g = 4_641_652
    codes, packed_genome = make_genome(torch, dev, g, seed=4600000)
    ix, tables = sbwt_amd.build_index_device(packed_genome.data_ptr(), g, [0], [0, g], [0, 0], interval=64, device=dev.index,
                                             want_index=True, want_tables=True)
    wpr = sbwt_amd.words_per_read(50)
    n = 1_000_000
    packed, lens = make_reads(torch, dev, codes, g, n, 50, seed=50, wpr=wpr)
    hp = packed.cpu().numpy().view(np.uint32)
    hl = lens.cpu().numpy().astype(np.uint32)
    seqs, offsets = unpack_reads(hp, hl)
    seqs = np.concatenate([seqs, np.zeros(1, np.uint8)])
    oix = oracle_index_from_tables(oracle_lib, tables, [0], [0, g], [0, 0])
    n1 = 200_000
    t1 = time.perf_counter()
    oix.map_batch_raw(seqs, offsets, n1, nthreads=1)
    r1 = n1 / (time.perf_counter() - t1)
    t1 = time.perf_counter()
    lo, hi, nh, first, ohits = oix.map_batch_raw(seqs, offsets, n, nthreads=cores)
    rn = n / (time.perf_counter() - t1)
    # the GPU on the same million reads (device pass only), and parity
    ctx = sbwt_amd.Context(ix, n)
    res = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    hits = torch.empty(2 * n * 16, dtype=torch.uint8, device=dev)
    ctx.map_batch_device(packed.data_ptr(), lens.data_ptr(), n, wpr, res.data_ptr(), hits.data_ptr(), 2 * n)
    ctx.map_batch_device(packed.data_ptr(), lens.data_ptr(), n, wpr, res.data_ptr(), hits.data_ptr(), 2 * n)
    ms = ctx.last_timing()[0]
    g_res = np.frombuffer(res.cpu().numpy().tobytes(), dtype=sbwt_amd.api.RESULT_DTYPE)
    same = bool(np.array_equal(g_res["sa_lo"], lo) and np.array_equal(g_res["sa_hi"], hi) and np.array_equal(g_res["nh"], nh))
    ctx.close(); oix.close(); ix.close()
    sbwt_amd.host_tables_free(tables)
    return {"workload": "configs[0]: 4,641,652-bp synthetic genome, 10^6 x 50-bp reads", "oracle_1_thread_reads_per_s": round(r1, 1),
            "oracle_1_thread_sample": n1, "oracle_all_cores_reads_per_s": round(rn, 1), "cores": cores,
            "gpu_device_pass_reads_per_s": round(n / (ms * 1e-3), 1), "gpu_identical_to_oracle": same,
            "reference_binary_reads_per_s": {"-n 1": "3.4e4 - 4.1e4", "-n 8": 2.73e5,
                                             "source": "BASELINE.md (survey container, other host; not measured here)"}}


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
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end and configs[0] legs")
    ap.add_argument("--min-len", type=int, default=None, help="reads of length min-len..read-len (config 5)")
    ap.add_argument("--mutate", type=float, default=0.0, help="fraction of reads with one substitution")
    ap.add_argument("--repeats", action="store_true", help="plant a repeat family, half of the reads inside it (config 5)")
    ap.add_argument("--genome", choices=["iid", "repeats"], default="iid", help="repeats: 40 %% of the bases in diverged repeat families")
    ap.add_argument("--rank-path", action="store_true", help="no memoised search prefix: backward search through rank queries")
    ap.add_argument("--shuffle-ties", action="store_true", help="suffix order stopped at sort_len 256, ties in arbitrary order")
    ap.add_argument("--layout", choices=["plain", "compact", "hits"], default="hits",
                    help="per-read output: plain = 32-byte record + hit slot per read (sbwt_map_batch_device); compact = 16-byte "
                         "records (sbwt_map_batch_device_compact); hits = compact with SBWT_MAP_HITS_ONLY (what SAM output needs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])
        return

    if args.rank_path:
        os.environ["SBWT_HASH_ORDER"] = "0"
        os.environ["SBWT_KMER_ORDER"] = "12"
    if args.shuffle_ties:
        os.environ["SBWT_DEBUG_SORT_DEPTH"] = "256"

    import torch
    import sbwt_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the library has no CPU fallback")
    if os.environ.get("BENCH_SHARE_DEVICE"):   # rehearsal of the N > 1 path on a one-GPU box: all ranks on device 0, gloo
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("BENCH_SHARE_DEVICE"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    t0 = time.time()
    L = args.read_len
    lens_chr, g, chr_start, npos_key, npos_val = synth_tables(args.scale)
    codes, packed_genome = make_genome(torch, dev, g, seed=38)
    slots = None
    repeat_cover = None
    if args.genome == "repeats":
        repeat_cover = plant_repeat_families(torch, dev, codes, g, seed=40)
        packed_genome = repack(torch, dev, codes, g)
    if args.repeats:
        slots = plant_repeats(torch, dev, codes, g, seed=5)
        packed_genome = repack(torch, dev, codes, g)
    torch.cuda.synchronize()
    t_genome = time.time() - t0

    # index: suffix order + tables on the GPU, straight into the HBM layout; rank 0 of a single-GPU run also exports
    # the reference-format tables for the CPU baseline leg and the oracle spot check
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
    plain = args.min_len is None and slots is None and args.mutate == 0.0 and args.genome == "iid"
    origins = [] if (plain and rank == 0) else None
    packed, lens = make_reads(torch, dev, codes, g, n_reads, L, seed=50 + rank, wpr=wpr, min_len=args.min_len, slots=slots,
                              mutate=args.mutate, origins=origins)
    del codes
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_reads = time.time() - t0

    many_hits = args.repeats or args.genome == "repeats"
    ctx = sbwt_amd.Context(ix, n_reads)
    compact = args.layout != "plain"
    mode = sbwt_amd.MAP_HITS_ONLY if args.layout == "hits" else 0
    wide_cap = 1 << 16
    free_b, _ = torch.cuda.mem_get_info()
    list_cap = int(min(80 * n_reads, max(1 << 20, (free_b - (6 << 30)) // 16)))       # room for the lists of multi-hit reads
    if compact:
        results = torch.empty(n_reads * 16, dtype=torch.uint8, device=dev)            # sbwt_compact_result[n_reads]
        hits_cap = (1 << 20) if not many_hits else list_cap                           # lists of multi-hit reads only
        wide = torch.empty(wide_cap * 48, dtype=torch.uint8, device=dev)
    else:
        results = torch.empty(n_reads * 32, dtype=torch.uint8, device=dev)            # sbwt_read_result[n_reads]
        hits_cap = n_reads + (1 << 20) if not many_hits else n_reads + list_cap
    hits = torch.empty(hits_cap * 16, dtype=torch.uint8, device=dev)

    def step():
        if compact:
            return ctx.map_batch_device_compact(packed.data_ptr(), lens.data_ptr(), n_reads, wpr, results.data_ptr(),
                                                hits.data_ptr(), hits_cap, wide.data_ptr(), wide_cap, limit=1000, strand=2,
                                                mode=mode)
        return (ctx.map_batch_device(packed.data_ptr(), lens.data_ptr(), n_reads, wpr, results.data_ptr(),
                                     hits.data_ptr(), hits_cap, limit=1000, strand=2), 0)

    from sbwt_amd import shard
    kernel_ms = []
    last = {"hits": (0, 0)}

    def timed_step():
        last["hits"] = step()                   # synchronises its own stream before returning
        kernel_ms.append(ctx.last_timing_ex())  # HIP events on the ctx stream, around the kernels of this step

    for _ in range(args.warmup):
        step()
    elapsed = shard.timed_steps(timed_step, args.steps, 0, dist=dist, sync=torch.cuda.synchronize, device=dev)
    n_hits, n_wide = last["hits"]

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if compact:
        # decode the compact records on the device into the plain layout (only for the checks below, after the timed region)
        rec = results.view(torch.int64).view(-1, 2)
        w0, w1 = rec[:, 0], rec[:, 1]
        meta = (w1 >> 56) & 0xFF
        m40 = (1 << 40) - 1
        n_no_sa = int(((meta & 128) != 0).sum())
        n_wide_rec = int(((meta & 64) != 0).sum())
        uniq = (meta & 1) != 0
        multi_m = (meta & 4) != 0
        plain_res = torch.zeros((n_reads, 4), dtype=torch.int64, device=dev)
        plain_res[:, 0] = w0 & m40
        plain_res[:, 1] = (w0 & m40) + ((w0 >> 40) & 0xFFFFFF)
        idx = torch.arange(n_reads, device=dev, dtype=torch.int64)
        nh_t = torch.where(uniq, torch.ones_like(idx), torch.where(multi_m, (w1 >> 40) & 0xFFFF, torch.zeros_like(idx)))
        plain_res[:, 2] = torch.where(uniq, idx, torch.where(multi_m & (nh_t > 0), n_reads + (w1 & m40), torch.zeros_like(idx)))
        plain_res[:, 3] = nh_t | (((meta >> 3) & 3) << 32)
        # unique hits at their read's index; the lists stay where they are (hits[]), addressed behind the first n_reads slots
        plain_hits = torch.zeros((n_reads, 2), dtype=torch.int64, device=dev)
        plain_hits[:, 0] = torch.where(uniq, w1 & m40, torch.zeros_like(idx))
        plain_hits[:, 1] = torch.where(uniq, ((w1 >> 40) & 0xFFFF) | (((meta >> 1) & 1) << 32), torch.zeros_like(idx))
        del rec, w0, w1, idx, nh_t
        if args.layout == "hits":
            # reads settled without an SA interval: the origin check below expects a size-1 interval for a hit
            no_sa = (meta & 128) != 0
            plain_res[:, 1] = torch.where(no_sa & uniq, plain_res[:, 0] + 1, plain_res[:, 1])
        results_plain = plain_res.view(-1).view(torch.uint8)
        hits_plain = plain_hits.view(-1).view(torch.uint8)
    else:
        n_no_sa = n_wide_rec = 0
        results_plain, hits_plain = results, hits.view(-1)[: n_reads * 16]
        n_hits = n_hits - n_reads if n_hits >= n_reads else 0      # lists only, as in the compact layouts
    origin = None
    if origins is not None:
        origin = origin_check(torch, dev, results_plain, hits_plain, origins, L, chr_start, npos_key, npos_val)
        del origins
    value = world * n_reads * args.steps / elapsed
    search_ms = float(np.mean([k[1] for k in kernel_ms]))
    multi_ms = float(np.mean([k[2] for k in kernel_ms]))
    res_np = np.frombuffer(results_plain.cpu().numpy().tobytes(), dtype=sbwt_amd.api.RESULT_DTYPE)
    mapped = int((res_np["nh"] > 0).sum())
    multi = int((res_np["nh"] > 1).sum())
    no_sa_np = None
    if args.layout == "hits":
        no_sa_np = ((results.view(torch.int64).view(-1, 2)[:20000, 1] >> 56) & 128).cpu().numpy() != 0
    # Algorithmic bytes of ONE k_search_sm launch: the search is a chain of random accesses
    # into index tables far larger than any cache, and HBM serves such an access as one 128-byte line whatever
    # part of it is used, so the unit is the line: 128 B x (index lines the reads of the batch need, counted by
    # the kernel itself: search-prefix entry, two-base blocks, memoised locate, text comparison, final row)
    # + the streamed bytes (read record + length in, result + hit out).
    stats = ctx.last_stats_ex()
    settle_ms = float(np.mean([k[3] for k in kernel_ms]))
    pending_ms = float(np.mean([k[4] for k in kernel_ms]))
    out_bytes = 16 if compact else 48     # per read: one 16-byte record, or a 32-byte record + the 16-byte hit slot
    if settle_ms > 0:
        # hits layout: the dominant kernel is k_settle -- one bucket (one line) per looked-up read, records in, records out
        dom_kernel, dom_ms = ("k_settle" if (args.layout == "hits" and wpr == 4) else "k_direct"), settle_ms
        accesses = int(stats[6])
        streamed = n_reads * (wpr * 4 + 4 + out_bytes) + int(stats[7]) * 4
    else:
        dom_kernel, dom_ms = "k_search_sm", search_ms
        accesses = int(stats[0])
        streamed = n_reads * (wpr * 4 + 4 + out_bytes) + int(n_hits) * 16
    algo_bytes = 128 * accesses + streamed
    requests = accesses + (streamed - n_reads * out_bytes) / 128.0 + n_reads * out_bytes / 64.0   # (PMC of the same kernel: within 1 %)
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
    all_accesses = int(stats[0])
    survey_per_read = 16 + 64 * (2 * (L - 12) + 32.5)   # reference-layout bytes per unique read
    # the random-access roofline of THIS device, measured in this process over the index's own largest table
    probe = {}
    for cpl in (1, 2, 4):
        rate, pbytes = ix.probe_random_access(chains_per_lane=cpl, chain_len=64, workgroups_per_cu=8)
        probe["%d_chains_per_lane" % cpl] = round(rate / 1e9, 2)
    gather_peak = max(probe.values())
    # ... and the same gather with what a mapping kernel cannot avoid: the address of every line comes from a streamed
    # 16-byte record and a 16-byte result is written per line (no dependent chains, no other work)
    stream_gather = None
    try:
        free_b2, _ = torch.cuda.mem_get_info()
        n_probe = int(min(n_reads, 100_000_000, max(0, free_b2 - (2 << 30)) // 32))
        if n_probe >= 1_000_000:
            stream_gather = ix.probe_gather_stream(n_probe, 8) / 1e9
    except Exception as e:  # (measurement only: an allocation failure here must not cost the bench line)
        sys.stderr.write("stream gather probe skipped: %s\n" % e)
    lines_per_s = accesses / (dom_ms * 1e-3) / 1e9
    stored = None
    tpath = os.path.join(ROOT, "profiles", "r2", "traffic_%s.json" % args.layout)   # PMC passes of an earlier run: context only
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("reads") == n_reads and t.get("read_len") == L and abs(t.get("scale", 1.0) - args.scale) < 1e-9 and plain \
                and not args.rank_path and t.get("kernel") == dom_kernel:
            stored = {"source": "profiles/r2/traffic_%s.json (rocprofv3 --pmc passes of the same command on another box of this "
                                "pool; NOT measured in this run)" % args.layout,
                      "traffic_GB_per_launch": t.get("traffic_GB_per_launch"),
                      "hbm_requests_per_launch": t.get("hbm_requests_per_launch")}
    roofline = {"bound": "hbm", "kernel": dom_kernel, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": stored["traffic_GB_per_launch"] if stored else None,
                "traffic_source": stored["source"] if stored else None,
                "launch_ms": round(dom_ms, 3), "algorithmic_GB_per_launch": round(algo_bytes / 1e9, 2),
                "algorithmic_bytes": "128 B per index line the kernel reads (a random access moves one 128-byte line of HBM whatever "
                                     "part of it is used: the probe below runs at the same rate for 16 .. 128-byte granules) + streamed "
                                     "bytes (record and length in, result out)",
                "index_lines_per_read": round(accesses / n_reads, 3), "bytes_per_read": round(algo_bytes / n_reads, 1),
                "index_lines_per_s_G": round(lines_per_s, 2),
                "gather_roofline": {"measured_in_this_run": True, "unit": "G dependent random 128-byte-line accesses/s",
                                    "table_GB": round(pbytes / 1e9, 1), "by_chains_per_lane": probe, "peak": gather_peak,
                                    "kernel_index_lines_frac": round(lines_per_s / gather_peak, 3),
                                    "streamed_gather_G_lines_per_s": round(stream_gather, 2) if stream_gather else None,
                                    "streamed_gather": "independent random lines whose addresses come from a streamed 16-byte record, "
                                                       "a 16-byte result written per line, nothing else (k_probe_stream): the "
                                                       "ceiling of a one-line-per-read mapping kernel on this device",
                                    "kernel_frac_of_streamed_gather": round(lines_per_s / stream_gather, 3) if stream_gather else None,
                                    # the probe issues nothing but random reads; the kernel also streams its records in and
                                    # out, and those requests share the same memory pipeline: index lines + 128-byte read
                                    # requests of the streamed input + 64-byte write requests of the output
                                    "kernel_hbm_requests_per_s_G": round(requests / (dom_ms * 1e-3) / 1e9, 2),
                                    "kernel_hbm_requests_frac": round(requests / (dom_ms * 1e-3) / 1e9 / gather_peak, 3)},
                "whole_step": {"device_ms": round(float(np.mean([k[0] for k in kernel_ms])), 3),
                               "index_lines_per_read": round(all_accesses / n_reads, 3),
                               "fast_pass_ms": round(settle_ms, 3), "pending_pass_ms": round(pending_ms, 3),
                               "reads_left_pending": int(stats[7])},
                "hash_lookups_per_read": round(stats[4] / n_reads, 4), "hash_hit_rate": round(stats[5] / stats[4], 4) if stats[4] else None,
                "stored_pmc": stored,
                "survey_formula_GBps": round(survey_per_read * n_reads / (dom_ms * 1e-3) / 1e9, 1)}

    cpu = None
    parity = None
    end_to_end = None
    config1 = None
    cores = min(16, os.cpu_count() or 1)
    if run_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        oix = oracle_index_from_tables(oracle_lib, tables, chr_start, npos_key, npos_val)
        probe_n = int(min(20000, n_reads))
        sample_max = 8_000_000
        host_packed = packed[:sample_max].cpu().numpy().view(np.uint32)
        host_lens = lens[:sample_max].cpu().numpy().astype(np.uint32)
        pseq, poff = unpack_reads(host_packed[:probe_n], host_lens[:probe_n])
        t1 = time.perf_counter()
        lo, hi, nh, first, ohits = oix.map_batch_raw(np.concatenate([pseq, np.zeros(1, np.uint8)]), poff, probe_n, nthreads=1)
        rate1 = probe_n / (time.perf_counter() - t1)
        # parity of the same reads through the GPU path
        gres = res_np[:probe_n]
        sa_ok = np.ones(probe_n, bool) if no_sa_np is None else ~no_sa_np[:probe_n]
        ok = bool(np.array_equal(gres["sa_lo"][sa_ok], lo[sa_ok]) and np.array_equal(gres["sa_hi"][sa_ok], hi[sa_ok])
                  and np.array_equal(gres["nh"], nh))
        if ok and int(nh.sum()):
            # ordered hit lists of the probe reads, gathered on the device (the lists of multi-hit reads lie anywhere)
            tot = int(nh.sum())
            within = np.arange(tot) - np.repeat(np.cumsum(nh) - nh, nh)
            gi = torch.from_numpy((np.repeat(gres["first_hit"], nh) + within).astype(np.int64)).to(dev)
            # slots below n_reads: the unique hit of that read; above: the lists (hits[] itself in the compact layouts)
            lists16 = hits.view(-1, 16)
            list_base = 0 if compact else n_reads
            in_list = gi >= n_reads
            got = hits_plain.view(-1, 16)[torch.where(in_list, torch.zeros_like(gi), gi)]
            got_l = lists16[torch.where(in_list, gi - n_reads + list_base, torch.zeros_like(gi))]
            ghits = np.frombuffer(torch.where(in_list[:, None], got_l, got).cpu().numpy().tobytes(), dtype=sbwt_amd.api.HIT_DTYPE)
            oi = (np.repeat(first, nh) + within).astype(np.int64)
            ok = bool(np.array_equal(ghits, ohits[oi]))
        parity = {"reads_checked": probe_n, "identical": ok,
                  "compared": "NH and ordered hit lists of every read; SA intervals of the %d reads that report one" % int(sa_ok.sum())}
        sample_n = int(min(sample_max, host_lens.size, max(probe_n, rate1 * cores * args.cpu_seconds * 0.7)))
        seqs, offsets = unpack_reads(host_packed[:sample_n], host_lens[:sample_n])
        t1 = time.perf_counter()
        oix.map_batch_raw(np.concatenate([seqs, np.zeros(1, np.uint8)]), offsets, sample_n, nthreads=cores)
        dt = time.perf_counter() - t1
        cpu = {"value": round(sample_n / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
               "host_cpus": os.cpu_count(),
               "sample": "first %d reads of the same batch, same index (reference-format tables), %d threads; "
                         "single thread: %.0f reads/s on %d reads" % (sample_n, cores, rate1, probe_n),
               "note": "the port is about 2x the reference binary per thread (7.9e4 vs 3.4-4.1e4 reads/s on configs[0] in the "
                       "survey container, SURVEY.md section 6): it is a baseline beside the kernel, not the target"}
        oix.close()
        sbwt_amd.host_tables_free(tables)
        if not args.no_extras and args.min_len is None:
            # end to end, measured in this run (never `value`): (a) host buffers in, host buffers out through
            # page-locked memory (PCIe both ways), (b) FASTQ file -> SAM file with 16 host threads
            from sbwt_amd import api
            n_e = int(min(n_reads, 16_000_000))
            pin = api.PinnedBuffer(n_e * (wpr * 4 + 4 + 32 + 2 * 16) + n_e * 16 + (1 << 20) * 16 + 1024 * 48)
            o = 0
            pk = pin.array(np.uint32, n_e * wpr, o); o += n_e * wpr * 4
            ln = pin.array(np.uint32, n_e, o); o += n_e * 4
            rs = pin.array(api.RESULT_DTYPE, n_e, o); o += 32 * n_e
            ht = pin.array(api.HIT_DTYPE, 2 * n_e, o); o += 32 * n_e
            cr = pin.array(api.COMPACT_DTYPE, n_e, o); o += 16 * n_e
            cl = pin.array(api.HIT_DTYPE, 1 << 20, o); o += 16 << 20
            cw = pin.array(api.WIDE_DTYPE, 1024, o)
            pk[:] = packed[:n_e].cpu().numpy().view(np.uint32).reshape(-1)
            ln[:] = lens[:n_e].cpu().numpy().astype(np.uint32)
            ctx_e = sbwt_amd.Context(ix, n_e)
            best = best_c = None
            if not many_hits:
                for _ in range(3):
                    t1 = time.perf_counter()
                    ctx_e.map_batch(pk.reshape(n_e, wpr), ln, out=(rs, ht))
                    dt = time.perf_counter() - t1
                    best = dt if best is None or dt < best else best
                for _ in range(3):
                    t1 = time.perf_counter()
                    ctx_e.map_batch_compact(pk.reshape(n_e, wpr), ln, mode=sbwt_amd.MAP_HITS_ONLY, out=(cr, cl, cw))
                    dt = time.perf_counter() - t1
                    best_c = dt if best_c is None or dt < best_c else best_c
            n_f = int(min(n_e, 8_000_000))
            tmpd = tempfile.mkdtemp(prefix="sbwt_bench_")
            fq, sam = os.path.join(tmpd, "r.fq"), os.path.join(tmpd, "o.sam")
            fq_bytes = write_fastq(fq, pk.reshape(n_e, wpr)[:n_f], ln[:n_f], L)
            fbest = None
            for _ in range(2):
                if os.path.exists(sam):
                    os.unlink(sam)                 # (dropping the previous run's gigabyte of page cache is not part of the run)
                t1 = time.perf_counter()
                st = ix.map_fastq(fq, sam, n_threads=cores)
                dt = time.perf_counter() - t1
                fbest = dt if fbest is None or dt < fbest else fbest
            sam_bytes = os.path.getsize(sam)
            del pk, ln, rs, ht, cr, cl, cw
            ctx_e.close()
            pin.close()
            shutil.rmtree(tmpd, ignore_errors=True)
            end_to_end = {"host_buffers_reads_per_s": round(n_e / best_c, 1) if best_c else None, "host_buffers_reads": n_e,
                          "host_buffers_note": "sbwt_map_batch_compact (hits only), page-locked host buffers: H2D of packed reads (20 B per "
                                               "read), kernels, D2H of the 16-byte records",
                          "host_buffers_plain_layout_reads_per_s": round(n_e / best, 1) if best else None,
                          "fastq_to_sam_reads_per_s": round(n_f / fbest, 1), "fastq_to_sam_reads": n_f, "host_threads": cores,
                          "fastq_MB": round(fq_bytes / 1e6), "sam_MB": round(sam_bytes / 1e6), "sam_records": int(st[4])}
        if not args.no_extras and plain and args.scale == 1.0:
            config1 = cpu_config1(torch, dev, sbwt_amd, oracle_lib, cores)

    workload_genome = "GRCh38-sized synthetic genome (%.3f Gbp non-N, 24 contigs, N runs%s%s)" % (
        g / 1e9, ", planted repeat family" if args.repeats else "",
        ", %.0f %% of the bases in diverged repeat families + exact duplications" % (100 * repeat_cover) if repeat_cover else "")
    line = {
        "metric": "mapped reads/sec (50bp exact, GRCh38) at 1/8 GPUs; achieved HBM GB/s",
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s, %d x %s reads per GPU, both strands, -m 1000 -s 2, inputs resident in HBM"
                               % (workload_genome, n_reads,
                                  ("%d-bp" % L if args.min_len is None else "%d..%d-nt" % (args.min_len, L))
                                  + (" (%.0f%% with one substitution)" % (100 * args.mutate) if args.mutate else " exact")),
                   "index_suffixes": int(info.n), "index_hbm_GB": round(info.device_bytes / 1e9, 2),
                   "search_prefix_tables": {"direct_order": int(info.kmer_order), "hash_order": int(info.hash_order)},
                   "suffix_order": "stopped at sort_len 256, ties shuffled" if args.shuffle_ties else "full",
                   "result_layout": {"plain": "32-byte record + 16-byte hit slot per read (sbwt_map_batch_device)",
                                     "compact": "16-byte record per read (sbwt_map_batch_device_compact)",
                                     "hits": "16-byte record per read, SBWT_MAP_HITS_ONLY: hits, NH and flags of every read; no SA interval "
                                             "for reads their located prefix settles (what the SAM output needs)"}[args.layout],
                   "reads_per_gpu": n_reads, "read_len": L, "parallelism": "replicated index, reads sharded x%d" % world},
        "roofline": roofline, "cpu_baseline": cpu, "end_to_end": end_to_end, "cpu_baseline_config1": config1,
        "detail": {"mapped_reads": mapped, "multi_hit_reads": multi, "list_hits": int(n_hits), "wide_records": n_wide_rec,
                   "reads_without_sa_interval": n_no_sa, "reads_to_multi_hit_locate": int(stats[1]),
                   "reads_sharing_a_located_list": int(stats[8]) if len(stats) > 8 else 0, "reads_to_depth_first_locate": int(stats[3]), "of_which_level_by_level": int(stats[9]) if len(stats) > 9 else 0, "reads_settled_by_a_miss": int(stats[10]) if len(stats) > 10 else 0,
                   "k_search_ms": round(search_ms, 3), "k_multi_ms": round(multi_ms, 3),
                   "fast_pass_ms": round(float(np.mean([k[3] for k in kernel_ms])), 3),
                   "pending_pass_ms": round(float(np.mean([k[4] for k in kernel_ms])), 3),
                   "setup_s": {"genome": round(t_genome, 1), "index_build": round(t_index, 1), "reads": round(t_reads, 1)},
                   "parity_vs_oracle": parity, "origin_check": origin},
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
