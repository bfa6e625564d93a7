#!/usr/bin/env python3
"""bench.py -- headline benchmark: mapped reads/s, 50-bp exact reads against a GRCh38-sized index.

Workload (BASELINE.json configs[2]): synthetic genome of 2.94 G non-N bases over 24 contigs with
GRCh38-like lengths and N runs, 100 M error-free 50-bp reads, both strands, resident in HBM.
One step = one pass of the hot path (sbwt_map_batch_device[_compact]) over the whole read batch.
N > 1: one process per GPU, index replicated, each rank maps its own 100 M reads (weak scaling, no
data-path collective; torch.distributed/RCCL only for the barrier and the max-over-ranks).  `--gpus N` without
a launcher (no WORLD_SIZE in the environment) starts the N ranks itself as child processes, before this
process touches a GPU; under torchrun each process is one rank.

The default single-GPU run also measures, in the same process and after the headline's timed region, the other
configurations of BASELINE.json and of DESIGN.md section 4 (`configs` in the JSON line: the interval-reporting
layouts, 100-bp reads, the dm6-sized index, the small-RNA workload, the repeat-rich genome on a fully sorted and on
a tie-shuffled index, the rank path, the index written to files and loaded back) -- each with its own index where it
needs one, each checked against the oracle.  `--no-configs` skips them; any workload flag selects ONE workload as the
headline instead (side lines for the profiling scripts, never the driver's number).

`--config 4` is BASELINE.json configs[3] as a STRONG-scaling run: a fixed total of 800 M x 100-bp reads cut into
contiguous shares by sbwt_shard_range, every rank mapping its share chunk by chunk from HBM.

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
    if total > g // 64 - 2:
        raise ValueError("plant_repeats: %d copies do not fit a genome of %d bases (a full-size workload)" % (total, g))
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
                                             "source": "BASELINE.md (other host)"}}


class Workload:
    """A synthetic genome and its index on one device.  The test knobs that shape an index (no memoised search prefix,
    suffix order stopped at sort_len) are environment variables the library reads while it BUILDS the index: set for
    the build, removed again."""

    def __init__(self, torch, dev, sb, scale=1.0, genome="iid", plant=False, rank_path=False, shuffle_ties=False,
                 want_tables=False):
        self.torch, self.dev, self.sb = torch, dev, sb
        self.scale, self.genome, self.plant, self.rank_path, self.shuffle_ties = scale, genome, plant, rank_path, shuffle_ties
        t0 = time.time()
        self.lens_chr, self.g, self.chr_start, self.npos_key, self.npos_val = synth_tables(scale)
        self.codes, packed_genome = make_genome(torch, dev, self.g, seed=38)
        self.slots = None
        self.repeat_cover = None
        if genome == "repeats":
            self.repeat_cover = plant_repeat_families(torch, dev, self.codes, self.g, seed=40)
            packed_genome = repack(torch, dev, self.codes, self.g)
        if plant:
            self.slots = plant_repeats(torch, dev, self.codes, self.g, seed=5)
            packed_genome = repack(torch, dev, self.codes, self.g)
        torch.cuda.synchronize()
        self.t_genome = time.time() - t0
        env = {}
        if rank_path:
            env.update(SBWT_HASH_ORDER="0", SBWT_KMER_ORDER="12")
        if shuffle_ties:
            env.update(SBWT_DEBUG_SORT_DEPTH="256")
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        t0 = time.time()
        try:
            self.ix, self.tables = sb.build_index_device(packed_genome.data_ptr(), self.g, self.chr_start, self.npos_key,
                                                         self.npos_val, interval=64, device=dev.index, want_index=True,
                                                         want_tables=want_tables)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.t_index = time.time() - t0
        del packed_genome
        self.info = self.ix.info()
        self.many_hits = plant or genome == "repeats"
        self._oix = None

    def describe(self):
        return "%s synthetic genome (%.3f Gbp non-N, 24 contigs, N runs%s%s)" % (
            "GRCh38-sized" if self.scale == 1.0 else "%.4g x GRCh38-sized" % self.scale, self.g / 1e9,
            ", planted repeat family" if self.plant else "",
            ", %.0f %% of the bases in diverged repeat families + exact duplications" % (100 * self.repeat_cover)
            if self.repeat_cover else "")

    def oracle(self, oracle_lib):
        if self._oix is None and self.tables is not None:
            self._oix = oracle_index_from_tables(oracle_lib, self.tables, self.chr_start, self.npos_key, self.npos_val)
        return self._oix

    def drop_genome(self):
        self.codes = None
        self.slots = None
        self.torch.cuda.empty_cache()

    def close(self):
        if self._oix is not None:
            self._oix.close()
            self._oix = None
        if self.tables is not None:
            self.sb.host_tables_free(self.tables)
            self.tables = None
        self.codes = None
        self.slots = None
        if self.ix is not None:
            self.ix.close()
            self.ix = None
        self.torch.cuda.empty_cache()


class Run:
    """One measured pass configuration over one batch: output buffers, the timed steps, and the checks on what came back."""

    def __init__(self, torch, dev, sb, wl, packed, lens, n_reads, L, wpr, layout, ctx=None):
        self.torch, self.dev, self.sb, self.wl = torch, dev, sb, wl
        self.packed, self.lens, self.n_reads, self.L, self.wpr, self.layout = packed, lens, n_reads, L, wpr, layout
        self.compact = layout != "plain"
        self.mode = sb.MAP_HITS_ONLY if layout == "hits" else 0
        self.own_ctx = ctx is None
        self.ctx = ctx if ctx is not None else sb.Context(wl.ix, n_reads)
        self.wide_cap = 1 << 16
        free_b, _ = torch.cuda.mem_get_info()
        list_cap = int(min(80 * n_reads, max(1 << 20, (free_b - (6 << 30)) // 16)))      # room for the lists of multi-hit reads
        if self.compact:
            self.results = torch.empty(n_reads * 16, dtype=torch.uint8, device=dev)      # sbwt_compact_result[n_reads]
            self.hits_cap = (1 << 20) if not wl.many_hits else list_cap                  # lists of multi-hit reads only
            self.wide = torch.empty(self.wide_cap * 48, dtype=torch.uint8, device=dev)
        else:
            self.results = torch.empty(n_reads * 32, dtype=torch.uint8, device=dev)      # sbwt_read_result[n_reads]
            self.hits_cap = n_reads + (1 << 20) if not wl.many_hits else n_reads + list_cap
            self.wide = None
        self.hits = torch.empty(self.hits_cap * 16, dtype=torch.uint8, device=dev)
        self.kernel_ms = []
        self.n_hits = self.n_wide = 0

    def step(self):
        if self.compact:
            return self.ctx.map_batch_device_compact(self.packed.data_ptr(), self.lens.data_ptr(), self.n_reads, self.wpr,
                                                     self.results.data_ptr(), self.hits.data_ptr(), self.hits_cap,
                                                     self.wide.data_ptr(), self.wide_cap, limit=1000, strand=2, mode=self.mode)
        return (self.ctx.map_batch_device(self.packed.data_ptr(), self.lens.data_ptr(), self.n_reads, self.wpr,
                                          self.results.data_ptr(), self.hits.data_ptr(), self.hits_cap, limit=1000, strand=2), 0)

    def timed(self, steps, warmup, dist=None):
        from sbwt_amd import shard

        def timed_step():
            self.n_hits, self.n_wide = self.step()           # synchronises its own stream before returning
            self.kernel_ms.append(self.ctx.last_timing_ex())  # HIP events on the ctx stream, around the kernels of this step

        for _ in range(warmup):
            self.step()
        self.elapsed = shard.timed_steps(timed_step, steps, 0, dist=dist, sync=self.torch.cuda.synchronize, device=self.dev)
        self.steps = steps
        self.stats = self.ctx.last_stats_ex()
        return self.elapsed

    def decode(self):
        """the records in the plain layout (device tensors), whatever layout the call wrote: only for the checks"""
        torch, dev, n_reads = self.torch, self.dev, self.n_reads
        if self.compact:
            rec = self.results.view(torch.int64).view(-1, 2)
            w0, w1 = rec[:, 0], rec[:, 1]
            meta = (w1 >> 56) & 0xFF
            m40 = (1 << 40) - 1
            self.n_no_sa = int(((meta & 128) != 0).sum())
            self.n_wide_rec = int(((meta & 64) != 0).sum())
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
            if self.layout == "hits":
                # reads settled without an SA interval: the origin check expects a size-1 interval for a hit
                no_sa = (meta & 128) != 0
                plain_res[:, 1] = torch.where(no_sa & uniq, plain_res[:, 0] + 1, plain_res[:, 1])
            del rec, w0, w1, idx, nh_t
            self.results_plain = plain_res.view(-1).view(torch.uint8)
            self.hits_plain = plain_hits.view(-1).view(torch.uint8)
            self.list_hits = self.n_hits
        else:
            self.n_no_sa = self.n_wide_rec = 0
            self.results_plain, self.hits_plain = self.results, self.hits.view(-1)[: n_reads * 16]
            self.list_hits = self.n_hits - n_reads if self.n_hits >= n_reads else 0      # lists only, as in the compact layouts
        self.res_np = np.frombuffer(self.results_plain.cpu().numpy().tobytes(), dtype=self.sb.api.RESULT_DTYPE)
        self.mapped = int((self.res_np["nh"] > 0).sum())
        self.multi = int((self.res_np["nh"] > 1).sum())

    def parity(self, oracle_lib, probe_n, nthreads):
        """the first probe_n reads through the oracle on the same tables: SA intervals (where the layout reports one), NH
        and the ordered hit lists"""
        torch, dev, n_reads = self.torch, self.dev, self.n_reads
        oix = self.wl.oracle(oracle_lib)
        if oix is None:
            return None, None
        probe_n = int(min(probe_n, n_reads))
        hp = self.packed[:probe_n].cpu().numpy().view(np.uint32)
        hl = self.lens[:probe_n].cpu().numpy().astype(np.uint32)
        pseq, poff = unpack_reads(hp, hl)
        t1 = time.perf_counter()
        lo, hi, nh, first, ohits = oix.map_batch_raw(np.concatenate([pseq, np.zeros(1, np.uint8)]), poff, probe_n, nthreads=nthreads)
        rate = probe_n / (time.perf_counter() - t1)
        gres = self.res_np[:probe_n]
        sa_ok = np.ones(probe_n, bool)
        if self.layout == "hits":
            sa_ok = ((self.results.view(torch.int64).view(-1, 2)[:probe_n, 1] >> 56) & 128).cpu().numpy() == 0
        ok = bool(np.array_equal(gres["sa_lo"][sa_ok], lo[sa_ok]) and np.array_equal(gres["sa_hi"][sa_ok], hi[sa_ok])
                  and np.array_equal(gres["nh"], nh))
        if ok and int(nh.sum()):
            # ordered hit lists of the probe reads, gathered on the device (the lists of multi-hit reads lie anywhere)
            tot = int(nh.sum())
            within = np.arange(tot) - np.repeat(np.cumsum(nh) - nh, nh)
            gi = torch.from_numpy((np.repeat(gres["first_hit"], nh) + within).astype(np.int64)).to(dev)
            # slots below n_reads: the unique hit of that read; above: the lists (hits[] itself in the compact layouts)
            lists16 = self.hits.view(-1, 16)
            list_base = 0 if self.compact else n_reads
            in_list = gi >= n_reads
            got = self.hits_plain.view(-1, 16)[torch.where(in_list, torch.zeros_like(gi), gi)]
            got_l = lists16[torch.where(in_list, gi - n_reads + list_base, torch.zeros_like(gi))]
            ghits = np.frombuffer(torch.where(in_list[:, None], got_l, got).cpu().numpy().tobytes(), dtype=self.sb.api.HIT_DTYPE)
            oi = (np.repeat(first, nh) + within).astype(np.int64)
            ok = bool(np.array_equal(ghits, ohits[oi]))
        return {"reads_checked": probe_n, "identical": ok, "sa_intervals_compared": int(sa_ok.sum())}, rate   # + NH and ordered hit lists of every read

    def accounting(self):
        """Algorithmic bytes of ONE launch of the dominant kernel: the search is a chain of random
        accesses into index tables far larger than any cache, and HBM serves such an access as one 128-byte line whatever
        part of it is used, so the unit is the line: 128 B x (index lines the reads of the batch need, counted by the
        kernels themselves) + the streamed bytes (read record + length in, result + hit out)."""
        n_reads, wpr, stats = self.n_reads, self.wpr, self.stats
        km = self.kernel_ms
        A = {"device_ms": float(np.mean([k[0] for k in km])), "search_ms": float(np.mean([k[1] for k in km])),
             "multi_ms": float(np.mean([k[2] for k in km])), "settle_ms": float(np.mean([k[3] for k in km])),
             "pending_ms": float(np.mean([k[4] for k in km]))}
        out_bytes = 16 if self.compact else 48     # per read: one 16-byte record, or a 32-byte record + the 16-byte hit slot
        if A["settle_ms"] > 0:
            # the dominant kernel is the fast pass -- one bucket (one line) per looked-up read, records in, records out
            fast = "k_rank" if getattr(self.wl, "rank_path", False) else "k_settle" if (self.layout == "hits" and wpr == 4) else "k_direct"
            A["kernel"], A["kernel_ms"] = fast, A["settle_ms"]
            A["accesses"] = int(stats[6])
            A["streamed"] = n_reads * (wpr * 4 + 4 + out_bytes) + int(stats[7]) * 4
        else:
            A["kernel"], A["kernel_ms"] = ("k_search_rounds" if wpr <= 8 else "k_search_sm"), A["search_ms"]
            A["accesses"] = int(stats[0])
            A["streamed"] = n_reads * (wpr * 4 + 4 + out_bytes) + int(self.list_hits) * 16
        if A["multi_ms"] > A["kernel_ms"]:
            # locate-bound workloads: the multi-hit kernels take most of the step; one text (or chain) line + the sa_full
            # word in and a 16-byte hit out per raw row, the search kernel's lines on top
            rows = int(stats[12]) if len(stats) > 12 else int(self.list_hits)   # raw rows located (a list several reads share: once)
            A["kernel"], A["kernel_ms"] = "k_multi_sort+k_multi_chain", A["multi_ms"]
            A["accesses"] = rows
            A["streamed"] = rows * (8 + 16)
            A["rows_located"] = rows
        A["out_bytes"] = out_bytes
        A["algo_bytes"] = 128 * A["accesses"] + A["streamed"]
        A["achieved_GBps"] = A["algo_bytes"] / (A["kernel_ms"] * 1e-3) / 1e9
        A["all_accesses"] = int(stats[0])
        return A

    def summary(self, name, workload, parity, world=1, short=None):
        """one short entry of the `configs` block"""
        A = self.accounting()
        st = self.stats
        e = {"name": name, "layout": self.layout,
             "reads_per_s": round(world * self.n_reads * self.steps / self.elapsed, 1),
             "ms_per_step": round(self.elapsed / self.steps * 1e3, 3),
             "kernel": A["kernel"], "kernel_ms": round(A["kernel_ms"], 3),
             "lines_per_read": round(A["all_accesses"] / self.n_reads, 3),
             "GBps": round(A["achieved_GBps"], 1), "frac": round(A["achieved_GBps"] / 8000.0, 4),
             "hash_hit_rate": round(st[5] / st[4], 4) if st[4] else None,
             "parity_vs_oracle": parity["identical"] if parity else None,
             "reads_checked": parity["reads_checked"] if parity else 0}
        if self.multi:     # locate-heavy workloads
            e.update({"multi_hit_reads": self.multi, "list_hits": int(self.list_hits),
                      "rows_located": int(st[12]), "reads_off_the_text": int(st[3]), "from_chain_tab": int(st[11])})
        if short is None:
            e["workload"] = workload
        return e

    def close(self):
        if self.own_ctx and self.ctx is not None:
            self.ctx.close()
        self.ctx = None
        self.results = self.hits = self.wide = None
        self.results_plain = self.hits_plain = None
        self.torch.cuda.empty_cache()


def side_case(torch, dev, sb, oracle_lib, wl, name, workload, n_reads, L, layouts, steps, cores, min_len=None, seed=50,
              probe_n=20000, packed=None, lens=None):
    """a configuration beside the headline: its reads, a few steps per layout, the first probe_n reads against the oracle"""
    wpr = sb.words_per_read(L)
    if packed is None:
        packed, lens = make_reads(torch, dev, wl.codes, wl.g, n_reads, L, seed=seed, wpr=wpr, min_len=min_len, slots=wl.slots)
    torch.cuda.synchronize()     # the library's stream does not wait for torch's: the reads must be there before a step starts
    out = []
    ctx = sb.Context(wl.ix, n_reads)
    try:
        for layout in layouts:
            run = Run(torch, dev, sb, wl, packed, lens, n_reads, L, wpr, layout, ctx=ctx)
            try:
                run.timed(steps, 1)
                run.decode()
                par, _ = run.parity(oracle_lib, probe_n, cores) if oracle_lib else (None, None)
                out.append(run.summary(name, workload, par, short=True if out else None))   # (the workload: once per name)
            finally:
                run.close()
    finally:
        ctx.close()
    return out


def file_path_case(torch, dev, sb, wl, cores, n_reads, L=50):
    """Searcher_impl::load_table (searcher.hpp:190-204) at full size: the index the headline ran on, written in the
    upstream file format (sbwt_host_tables_save + the three side files), loaded back (sbwt_index_load) and run on the
    same reads -- the records must be the same bytes.  The built index is closed first: two replicas do not fit."""
    wpr = sb.words_per_read(L)
    packed, lens = make_reads(torch, dev, wl.codes, wl.g, n_reads, L, seed=77, wpr=wpr)
    torch.cuda.synchronize()
    results = torch.empty(n_reads * 16, dtype=torch.uint8, device=dev)
    hits = torch.empty((1 << 20) * 16, dtype=torch.uint8, device=dev)
    wide = torch.empty((1 << 16) * 48, dtype=torch.uint8, device=dev)

    def once(ix):
        ctx = sb.Context(ix, n_reads)
        for _ in range(2):
            ctx.map_batch_device_compact(packed.data_ptr(), lens.data_ptr(), n_reads, wpr, results.data_ptr(), hits.data_ptr(),
                                         1 << 20, wide.data_ptr(), 1 << 16, limit=1000, strand=2, mode=0)
        ms = ctx.last_timing_ex()[0]
        ctx.close()
        return results.clone(), ms

    ref, ms_built = once(wl.ix)
    tmpd = tempfile.mkdtemp(prefix="sbwt_files_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        prefix = os.path.join(tmpd, "idx.")
        t0 = time.time()
        sb.host_tables_save(wl.tables, prefix + "t_table.bwt")
        names = ["chr%d" % (k + 1) for k in range(len(wl.chr_start))]
        with open(prefix + "chrStart", "w") as f:
            for nm, v in zip(names, wl.chr_start):
                f.write("%s\t%d\n" % (nm, v))
        with open(prefix + "chrLen", "w") as f:
            for nm, v in zip(names, wl.lens_chr):
                f.write("%s\t%d\n" % (nm, v))
        write_npos(prefix + "NposLen.z", wl.npos_key, wl.npos_val)
        t_save = time.time() - t0
        file_gb = os.path.getsize(prefix + "t_table.bwt") / 1e9
        wl.ix.close()
        wl.ix = None
        torch.cuda.empty_cache()
        t0 = time.time()
        ix2 = sb.Index.load(prefix, dev.index)
        t_load = time.time() - t0
        info2 = ix2.info()
        got, ms_loaded = once(ix2)
        # the same bytes -- except WHERE the list of a multi-hit read lies in hits[], which no call promises
        ra, rb = ref.view(torch.int64).view(-1, 2), got.view(torch.int64).view(-1, 2)
        multi = ((ra[:, 1] >> 56) & 4) != 0
        same = bool(torch.equal(ra[:, 0], rb[:, 0]) and torch.equal(torch.where(multi, ra[:, 1] >> 40, ra[:, 1]),
                                                                  torch.where(multi, rb[:, 1] >> 40, rb[:, 1])))
        ix2.close()
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)
    return {"name": "index_files", "workload": "the headline's index through upstream-format files (%.2f GB t_table.bwt), sbwt_index_load, %d x %d bp"
            % (file_gb, n_reads, L), "layout": "compact",
            "index_save_s": round(t_save, 1), "index_load_s": round(t_load, 1),
            "reads_per_s": round(n_reads / (ms_loaded * 1e-3), 1), "device_ms_loaded": round(ms_loaded, 3),
            "device_ms_built": round(ms_built, 3), "index_hbm_GB": round(info2.device_bytes / 1e9, 2),
            "search_prefix_tables": {"direct_order": int(info2.kmer_order), "hash_order": int(info2.hash_order)},
            "records_identical_to_built_index": same}


def write_npos(path, keys, vals):
    """prefix.NposLen.z in the upstream archive layout: header, map class-info, count, item version,
    pair class-info, then (key, value) pairs"""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", 22) + b"serialization::archive" + struct.pack("<H", 10) + bytes([4, 8, 4, 8]) + struct.pack("<I", 1))
        f.write(b"\0" * 5 + struct.pack("<Q", len(keys)) + struct.pack("<I", 0) + b"\0" * 5)
        for k, v in zip(keys, vals):
            f.write(struct.pack("<QQ", int(k), int(v)))


def strong_scaling(args, torch, dev, sb, dist, rank, world):
    """BASELINE.json configs[3]: a fixed total of 800 M x 100-bp reads over 1/2/4/8 ranks.  sbwt_shard_range cuts the
    input into contiguous shares (as Job_distributer::distribute_jobs, job_distributer.hpp:37-90, hands out groups of
    ONE input in reading order); a rank holds its share in HBM and maps it chunk by chunk (compact records, hits only,
    two result buffers in turn: what a consumer would drain while the next chunk maps)."""
    from sbwt_amd import shard
    L = args.read_len if args.read_len != 50 else 100
    total = args.reads if args.reads != 100_000_000 else 800_000_000
    wl = Workload(torch, dev, sb, scale=args.scale)
    a, b = shard.shard_range(total, rank, world)
    mine = b - a
    wpr = sb.words_per_read(L)
    chunk = int(min(args.chunk, max(mine, 1)))
    free_b, _ = torch.cuda.mem_get_info()
    need = mine * (wpr * 4 + 4) + 2 * chunk * 16 + (4 << 30)
    if need > free_b:
        raise SystemExit("bench.py --config 4: a share of %d reads (%.1f GB) does not fit the %.1f GB this GPU has left next to "
                         "the index; use more ranks or --reads" % (mine, need / 1e9, free_b / 1e9))
    t0 = time.time()
    packed = torch.empty((mine, wpr), dtype=torch.int32, device=dev)
    lens = torch.empty((mine,), dtype=torch.int32, device=dev)
    gen = 1 << 21   # (make_reads needs about 1.5 KB of temporaries per 100-bp read: 3 GB a piece, next to a share of up to 29 GB)
    for c in range(a // gen, (b - 1) // gen + 1 if mine else 0):
        # piece c of the whole input (reads [c * gen, (c + 1) * gen)) is generated from seed 1000 + c whoever needs a
        # part of it: read i of the input is the same read whatever the number of ranks
        p, l = make_reads(torch, dev, wl.codes, wl.g, min(gen, total - c * gen), L, seed=1000 + c, wpr=wpr)
        s0, s1 = max(a, c * gen), min(b, (c + 1) * gen)
        packed[s0 - a:s1 - a] = p[s0 - c * gen:s1 - c * gen]
        lens[s0 - a:s1 - a] = l[s0 - c * gen:s1 - c * gen]
        del p, l
    wl.drop_genome()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()   # (the ctx below allocates through the library, not through torch's cache)
    t_reads = time.time() - t0
    ctx = sb.Context(wl.ix, chunk)
    outs = [torch.empty(chunk * 16, dtype=torch.uint8, device=dev) for _ in range(2)]
    hits = torch.empty((1 << 20) * 16, dtype=torch.uint8, device=dev)
    wide = torch.empty((1 << 16) * 48, dtype=torch.uint8, device=dev)
    mapped = [0]
    dev_ms = []

    def step():
        k = 0
        ms = 0.0
        for c0 in range(0, mine, chunk):
            n = min(chunk, mine - c0)
            ctx.map_batch_device_compact(packed[c0:].data_ptr(), lens[c0:].data_ptr(), n, wpr, outs[k & 1].data_ptr(), hits.data_ptr(),
                                         1 << 20, wide.data_ptr(), 1 << 16, limit=1000, strand=2, mode=sb.MAP_HITS_ONLY)
            ms += ctx.last_timing_ex()[0]
            k += 1
        dev_ms.append(ms)

    elapsed = shard.timed_steps(step, args.steps, args.warmup, dist=dist, sync=torch.cuda.synchronize, device=dev)
    # mapped reads of this rank's last chunk pass (a property check: error-free reads map unless they span a boundary)
    n_last = mine - (mine - 1) // chunk * chunk if mine else 0
    kk = ((mine - 1) // chunk) & 1 if mine else 0
    if mine:
        w1 = outs[kk].view(torch.int64).view(-1, 2)[:n_last, 1]
        mapped[0] = int((((w1 >> 56) & 5) != 0).sum())
    cnt = shard.sum_counters([mine, mapped[0], n_last], dist=dist)
    if rank == 0:
        line = {"metric": "mapped reads/sec (50bp exact, GRCh38) at 1/8 GPUs; achieved HBM GB/s",
                "value": round(total * args.steps / elapsed, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": "BASELINE.json configs[3]: %s, a fixed total of %d x %d-bp exact reads cut into %d contiguous "
                                       "shares (sbwt_shard_range), each resident in its GPU's HBM and mapped in chunks of %d reads, "
                                       "both strands, -m 1000 -s 2, compact records, hits only" % (wl.describe(), total, L, world, chunk),
                           "index_suffixes": int(wl.info.n), "index_hbm_GB": round(wl.info.device_bytes / 1e9, 2),
                           "total_reads": total, "reads_this_rank": mine, "read_len": L, "chunk_reads": chunk,
                           "parallelism": "replicated index, one input cut into %d contiguous shares" % world},
                "roofline": None, "cpu_baseline": None,
                "detail": {"reads_all_ranks": cnt[0], "device_ms_per_step_rank0": round(float(np.mean(dev_ms[-args.steps:])), 3),
                           "mapped_in_last_chunks": cnt[1], "reads_in_last_chunks": cnt[2],
                           "setup_s": {"genome": round(wl.t_genome, 1), "index_build": round(wl.t_index, 1), "reads": round(t_reads, 1)}}}
        print(json.dumps(line), flush=True)
    ctx.close()
    wl.close()


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
    ap.add_argument("--no-configs", action="store_true", help="skip the other configurations of the default run")
    ap.add_argument("--force-configs", action="store_true", help="(debug) the configurations block at any --scale / --reads, sizes scaled along")
    ap.add_argument("--only-configs", default="", help="(debug) comma-separated names: only these entries of the configurations block")
    ap.add_argument("--configs-budget-s", type=float, default=330.0, help="no further configuration is started after this many seconds of run time")
    ap.add_argument("--min-len", type=int, default=None, help="reads of length min-len..read-len (config 5)")
    ap.add_argument("--mutate", type=float, default=0.0, help="fraction of reads with one substitution")
    ap.add_argument("--repeats", action="store_true", help="plant a repeat family, half of the reads inside it (config 5)")
    ap.add_argument("--genome", choices=["iid", "repeats"], default="iid", help="repeats: 40 %% of the bases in diverged repeat families")
    ap.add_argument("--rank-path", action="store_true", help="no memoised search prefix: backward search through rank queries")
    ap.add_argument("--shuffle-ties", action="store_true", help="suffix order stopped at sort_len 256, ties in arbitrary order")
    ap.add_argument("--layout", choices=["plain", "compact", "hits"], default="hits",
                    help="per-read output: plain = 32-byte record + hit slot per read (sbwt_map_batch_device); compact = 16-byte "
                         "records (sbwt_map_batch_device_compact); hits = compact with SBWT_MAP_HITS_ONLY (what SAM output needs)")
    ap.add_argument("--config", type=int, default=0, help="4: BASELINE.json configs[3] as a strong-scaling run (fixed total of 8e8 x 100-bp reads)")
    ap.add_argument("--chunk", type=int, default=50_000_000, help="--config 4: reads per resident chunk")
    args = ap.parse_args()
    t_start = time.time()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])
        return

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

    if args.config == 4:
        strong_scaling(args, torch, dev, sbwt_amd, dist, rank, world)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.config:
        raise SystemExit("bench.py: --config takes 4 (the other configurations are part of the default run)")

    L = args.read_len
    # index: suffix order + tables on the GPU, straight into the HBM layout; rank 0 of a single-GPU run also exports
    # the reference-format tables for the CPU baseline leg and the oracle spot check
    run_cpu = (world == 1 and not args.no_cpu)
    wl = Workload(torch, dev, sbwt_amd, scale=args.scale, genome=args.genome, plant=args.repeats, rank_path=args.rank_path,
                  shuffle_ties=args.shuffle_ties, want_tables=run_cpu)
    ix, info = wl.ix, wl.info
    g, chr_start, npos_key, npos_val = wl.g, wl.chr_start, wl.npos_key, wl.npos_val
    default_workload = (args.min_len is None and not args.repeats and args.mutate == 0.0 and args.genome == "iid"
                        and not args.rank_path and not args.shuffle_ties)

    wpr = sbwt_amd.words_per_read(L)
    n_reads = args.reads
    t0 = time.time()
    plain = args.min_len is None and wl.slots is None and args.mutate == 0.0 and args.genome == "iid"
    origins = [] if (plain and rank == 0) else None
    packed, lens = make_reads(torch, dev, wl.codes, g, n_reads, L, seed=50 + rank, wpr=wpr, min_len=args.min_len, slots=wl.slots,
                              mutate=args.mutate, origins=origins)
    do_configs = (world == 1 and default_workload and not args.no_configs and L == 50 and args.layout == "hits"
                  and ((args.scale == 1.0 and n_reads == 100_000_000) or args.force_configs))
    if not do_configs:
        wl.drop_genome()
    torch.cuda.synchronize()
    t_reads = time.time() - t0

    many_hits = wl.many_hits
    head = Run(torch, dev, sbwt_amd, wl, packed, lens, n_reads, L, wpr, args.layout)
    compact = head.compact
    elapsed = head.timed(args.steps, args.warmup, dist=dist)
    n_hits, n_wide = head.n_hits, head.n_wide

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    head.decode()
    n_hits = head.list_hits
    origin = None
    if origins is not None:
        origin = origin_check(torch, dev, head.results_plain, head.hits_plain, origins, L, chr_start, npos_key, npos_val)
        del origins
    value = world * n_reads * args.steps / elapsed
    A = head.accounting()
    stats = head.stats
    dom_kernel, dom_ms, accesses, algo_bytes = A["kernel"], A["kernel_ms"], A["accesses"], A["algo_bytes"]
    out_bytes = A["out_bytes"]
    requests = accesses + (A["streamed"] - n_reads * out_bytes) / 128.0 + n_reads * out_bytes / 64.0   # (PMC of the same kernel: within 1 %)
    achieved = A["achieved_GBps"]
    survey_per_read = 16 + 64 * (2 * (L - 12) + 32.5)   # reference-layout bytes per unique read
    # the random-access roofline of THIS device, measured in this process over the index's own largest table
    probe = {}
    for cpl in (1, 2, 4):
        rate, pbytes = ix.probe_random_access(chains_per_lane=cpl, chain_len=64, workgroups_per_cu=8)
        probe["%d" % cpl] = round(rate / 1e9, 2)
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
    for tdir in ("r3", "r2"):                      # PMC passes of an earlier run of the same command: context only
        tpath = os.path.join(ROOT, "profiles", tdir, "traffic_%s.json" % args.layout)
        if stored is None and os.path.exists(tpath):
            t = json.load(open(tpath))
            if t.get("reads") == n_reads and t.get("read_len") == L and abs(t.get("scale", 1.0) - args.scale) < 1e-9 and plain \
                    and not args.rank_path and t.get("kernel") == dom_kernel:
                stored = {"source": "profiles/%s/traffic_%s.json: PMC passes of the same command, another box, NOT this run" % (tdir, args.layout),
                          "traffic_GB_per_launch": t.get("traffic_GB_per_launch"),
                          "hbm_requests_per_launch": t.get("hbm_requests_per_launch")}
    roofline = {"bound": "hbm", "kernel": dom_kernel, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": stored["traffic_GB_per_launch"] if stored else None,
                "launch_ms": round(dom_ms, 3), "algorithmic_GB_per_launch": round(algo_bytes / 1e9, 2),
                "index_lines_per_read": round(accesses / n_reads, 3), "bytes_per_read": round(algo_bytes / n_reads, 1),
                "index_lines_per_s_G": round(lines_per_s, 2),
                "gather_roofline": {"measured_in_this_run": True, "unit": "G dependent random 128-B-line accesses/s",
                                    "table_GB": round(pbytes / 1e9, 1), "by_chains_per_lane": probe, "peak": gather_peak,
                                    "kernel_index_lines_frac": round(lines_per_s / gather_peak, 3),
                                    # independent random lines whose addresses come from a streamed 16-byte record, a 16-byte
                                    # result written per line (k_probe_stream): the ceiling of a one-line-per-read kernel
                                    "streamed_gather_G_lines_per_s": round(stream_gather, 2) if stream_gather else None,
                                    "kernel_frac_of_streamed_gather": round(lines_per_s / stream_gather, 3) if stream_gather else None,
                                    # index lines + 128-byte read requests of the streamed input + 64-byte write requests of the output
                                    "kernel_hbm_requests_per_s_G": round(requests / (dom_ms * 1e-3) / 1e9, 2),
                                    "kernel_hbm_requests_frac": round(requests / (dom_ms * 1e-3) / 1e9 / gather_peak, 3)},
                "whole_step": {"device_ms": round(A["device_ms"], 3),
                               "index_lines_per_read": round(A["all_accesses"] / n_reads, 3),
                               "fast_pass_ms": round(A["settle_ms"], 3), "pending_pass_ms": round(A["pending_ms"], 3),
                               "reads_left_pending": int(stats[7])},
                "hash_lookups_per_read": round(stats[4] / n_reads, 4), "hash_hit_rate": round(stats[5] / stats[4], 4) if stats[4] else None,
                "stored_pmc": stored,
                "survey_formula_GBps": round(survey_per_read * n_reads / (dom_ms * 1e-3) / 1e9, 1)}

    cpu = None
    parity = None
    end_to_end = None
    config1 = None
    oracle_lib = None
    cores = min(16, os.cpu_count() or 1)
    if run_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        oix = wl.oracle(oracle_lib)
        parity, rate1 = head.parity(oracle_lib, 20000, 1)
        probe_n = parity["reads_checked"]
        sample_max = 8_000_000
        host_packed = packed[:sample_max].cpu().numpy().view(np.uint32)
        host_lens = lens[:sample_max].cpu().numpy().astype(np.uint32)
        sample_n = int(min(sample_max, host_lens.size, max(probe_n, rate1 * cores * args.cpu_seconds * 0.7)))
        seqs, offsets = unpack_reads(host_packed[:sample_n], host_lens[:sample_n])
        t1 = time.perf_counter()
        oix.map_batch_raw(np.concatenate([seqs, np.zeros(1, np.uint8)]), offsets, sample_n, nthreads=cores)
        dt = time.perf_counter() - t1
        del seqs, offsets
        cpu = {"value": round(sample_n / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
               "sample": "first %d reads of the same batch, same index (reference-format tables), %d threads; "
                         "single thread: %.0f reads/s on %d reads" % (sample_n, cores, rate1, probe_n),
               "note": "about 2x the reference binary per thread"}
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
                          "host_buffers_plain_layout_reads_per_s": round(n_e / best, 1) if best else None,
                          "fastq_to_sam_reads_per_s": round(n_f / fbest, 1), "fastq_to_sam_reads": n_f, "host_threads": cores,
                          "fastq_MB": round(fq_bytes / 1e6), "sam_MB": round(sam_bytes / 1e6), "sam_records": int(st[4])}

    workload_genome = wl.describe()
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
                   "result_layout": {"plain": "32-byte records (sbwt_map_batch_device)", "compact": "16-byte records (sbwt_map_batch_device_compact)",
                                     "hits": "16-byte records, SBWT_MAP_HITS_ONLY (what the SAM output needs)"}[args.layout],
                   "reads_per_gpu": n_reads, "read_len": L, "parallelism": "replicated index, reads sharded x%d" % world},
        "roofline": roofline, "cpu_baseline": cpu, "end_to_end": end_to_end, "cpu_baseline_config1": None,
        "detail": {"mapped_reads": head.mapped, "multi_hit_reads": head.multi, "list_hits": int(n_hits), "wide_records": head.n_wide_rec,
                   "reads_without_sa_interval": head.n_no_sa, "reads_to_multi_hit_locate": int(stats[1]),
                   "reads_sharing_a_located_list": int(stats[8]), "reads_to_depth_first_locate": int(stats[3]),
                   "of_which_level_by_level": int(stats[9]), "reads_settled_by_a_miss": int(stats[10]),
                   "k_search_ms": round(A["search_ms"], 3), "k_multi_ms": round(A["multi_ms"], 3),
                   "fast_pass_ms": round(A["settle_ms"], 3), "pending_pass_ms": round(A["pending_ms"], 3),
                   "setup_s": {"genome": round(wl.t_genome, 1), "index_build": round(wl.t_index, 1), "reads": round(t_reads, 1)},
                   "parity_vs_oracle": parity, "origin_check": origin},
    }
    head.close()

    # ---- the other configurations (after the headline's timed region; each checked against the oracle) ----------
    if do_configs:
        configs = []
        skipped = []
        olib = oracle_lib if run_cpu else None

        only = [x for x in args.only_configs.split(",") if x]

        def guarded(name, fn):
            if only and name not in only:
                return
            if time.time() - t_start > args.configs_budget_s:
                skipped.append(name)
                return
            t1 = time.time()
            try:
                out = fn()
                out = out if isinstance(out, list) else [out]
                out[0]["s"] = round(time.time() - t1, 1)    # wall time of this name's entries, builds and checks included
                configs.extend(out)
            except Exception as e:   # a side line must never cost the headline
                configs.append({"name": name, "error": "%s: %s" % (type(e).__name__, e)})
                torch.cuda.empty_cache()

        steps_side = 3
        w3 = "configs[2], SA intervals reported"
        wl_n = int(wl.info.n)      # (entries on an index of another size say so)
        guarded("config3_intervals", lambda: side_case(torch, dev, sbwt_amd, olib, wl, "config3_intervals", w3, n_reads, L,
                                                       ["compact", "plain"], steps_side, cores, packed=packed, lens=lens))
        del packed, lens
        torch.cuda.empty_cache()
        guarded("config4_per_gpu", lambda: side_case(torch, dev, sbwt_amd, olib, wl, "config4_per_gpu",
                                                     "configs[3] per GPU: %d x 100 bp" % n_reads, n_reads, 100,
                                                     ["hits", "compact"], steps_side, cores, seed=51))
        if run_cpu and not args.no_extras:
            guarded("index_files", lambda: file_path_case(torch, dev, sbwt_amd, wl, cores, n_reads // 10))
        wl.close()

        def fresh(name, workload, n, rl, layouts, probe_n=20000, min_len=None, **kw):
            def fn():
                kw.setdefault("scale", args.scale)
                w2 = Workload(torch, dev, sbwt_amd, want_tables=run_cpu, **kw)
                try:
                    out = side_case(torch, dev, sbwt_amd, olib, w2, name, workload, n, rl, layouts, steps_side, cores,
                                    min_len=min_len, probe_n=probe_n)
                    for e in out:
                        if int(w2.info.n) != int(wl_n):
                            e["index_suffixes"] = int(w2.info.n)
                    return out
                finally:
                    w2.close()
            guarded(name, fn)

        n_mid, n_small = n_reads // 5, n_reads // 10     # 2 x 10^7 and 10^7 in the full-size run
        fresh("config2_dm6", "configs[1]: dm6-sized index, %d x 36 bp" % n_small, n_small, 36, ["hits", "compact"], scale=0.0486 * args.scale)
        fresh("repeats", "40 %% of the genome in diverged repeat families, %d x 50 bp" % n_mid, n_mid, 50, ["hits"], probe_n=5000, genome="repeats")
        fresh("repeats_shuffled_ties", "the same, suffix order stopped at sort_len 256, ties shuffled (upstream's builder), %d x 50 bp" % n_mid,
              n_mid, 50, ["hits"], probe_n=5000, genome="repeats", shuffle_ties=True)
        fresh("config5_smallrna", "configs[4] per GPU: planted 30-mer family, %d x 18..30 nt" % n_mid, n_mid, 30,
              ["hits"], probe_n=5000, min_len=18, plant=True)
        fresh("rank_path", "no memoised prefix (12-mer table + rank queries), %d x 50 bp" % n_reads, n_reads, 50,
              ["compact"], rank_path=True)
        if run_cpu and not args.no_extras:
            try:
                line["cpu_baseline_config1"] = cpu_config1(torch, dev, sbwt_amd, oracle_lib, cores)
            except Exception as e:
                line["cpu_baseline_config1"] = {"error": "%s: %s" % (type(e).__name__, e)}
        line["configs"] = configs
        if skipped:
            line["configs_skipped_for_time"] = skipped
        line["run_s"] = round(time.time() - t_start, 1)
    else:
        if run_cpu and not args.no_extras and plain and args.scale == 1.0:
            wl.close()
            line["cpu_baseline_config1"] = cpu_config1(torch, dev, sbwt_amd, oracle_lib, cores)
        wl.close()
    # (one line the driver keeps whole: counters that are zero on this workload say nothing)
    line["detail"] = {k: v for k, v in line["detail"].items() if not (isinstance(v, int) and not isinstance(v, bool) and v == 0)}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
