"""GPU parity fuzzer: random genomes, index shapes, pass knobs and read mixes, every record against the oracle.

Test infrastructure (it drives the oracle as the checker): `python tests/fuzz_parity.py --seeds 0:40` on a GPU box;
tests/test_gpu_fuzz.py runs a handful of seeds of it inside the GPU suite.  One seed = one genome (random base
composition; repeat families of random unit length, copy number, divergence and strand; tandem repeats; exact long
duplications, which are the tie groups of a depth-limited suffix order; random chromosome starts and N runs), one index
built on the GPU under a random choice of the build knobs (prefix-table orders, slot text, twins, chain_tab, suffix
order stopped at 27 / 54 / 256 bases, SA sampling interval), one context under a random choice of the pass knobs, one
batch of reads (substrings of both strands with and without a substitution, reads inside the families, reads across
chromosome and N-run boundaries, reads at the text ends, random and low-complexity reads; lengths 12 .. 200, and from
seed 100000 on up to 512) mapped
through the 32-byte records, the compact records and hits-only, for a random -m limit and strand choice (from seed
200000 on also through sbwt_map_batch_device and sbwt_map_batch_device_static, from a hits buffer that is too small at
first).  What is
compared is what start_sbwt_match (searcher.hpp:386-539) produces: SA interval, NH, and the ordered hit list of every
read.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

BUILD_KNOBS = {
    "SBWT_HASH_ORDER": [None, None, None, "0", "14", "16", "18", "20"],
    "SBWT_KMER_ORDER": [None, None, "12", "13", "14", "15"],
    "SBWT_HASH_TEXT": [None, None, None, "0"],
    "SBWT_HASH_GROUP": [None, None, None, "2"],
    "SBWT_NO_TWINS": [None, None, None, "1"],
    "SBWT_NO_CHAIN_TAB": [None, None, None, None, "1"],
    "SBWT_NO_HASH_COMPLETE": [None, None, None, None, "1"],
    "SBWT_DEBUG_SORT_DEPTH": [None, None, "27", "54", "108", "256"],
}
PASS_KNOBS = {
    "SBWT_NO_SETTLE": [None] * 6 + ["1"],
    "SBWT_NO_DIRECT": [None] * 6 + ["1"],
    "SBWT_NO_BFS": [None] * 6 + ["1"],
    "SBWT_NO_CHAIN": [None] * 5 + ["1"],
    "SBWT_NO_ROUNDS": [None] * 5 + ["1"],
    "SBWT_NO_RANK_PASS": [None] * 5 + ["1"],
    "SBWT_NO_SEGMENTS": [None] * 5 + ["1", "2"],
    "SBWT_NO_SHARED_LISTS": [None] * 6 + ["1"],
    "SBWT_NO_SPECULATIVE_PENDING": [None] * 6 + ["1"],
}


def pack_codes(codes):
    n = len(codes)
    words = (n + 15) // 16
    buf = np.zeros(words * 16, dtype=np.uint64)
    buf[:n] = codes
    sh = (2 * np.arange(16, dtype=np.uint64))
    return (buf.reshape(words, 16) << sh).sum(axis=1).astype(np.uint32)


def rc(a):
    return (3 - a[::-1]).astype(np.uint8)


def make_genome(rng):
    g = int(rng.choice([40_000, 250_000, 1_200_000, 2_400_000, 4_500_000]))
    p = rng.dirichlet(np.full(4, float(rng.choice([0.7, 3.0, 50.0]))))
    p = np.maximum(p, 0.02)
    p /= p.sum()
    codes = rng.choice(4, size=g, p=p).astype(np.uint8)
    fams = []
    cursor = 0

    plan = []
    for _ in range(int(rng.integers(0, 24))):
        unit = int(rng.choice([14, 20, 21, 24, 30, 32, 40, 64, 100, 300, 600]))
        copies = int(rng.choice([2, 2, 3, 5, 8, 20, 63, 64, 65, 130, 256, 257, 400, 512, 513, 700, 1000, 1001, 1500]))
        if unit * copies > g // 6:
            copies = max(2, (g // 6) // unit)
        div = float(rng.choice([0.0, 0.0, 0.01, 0.05]))
        plan.append((rng.choice(4, size=unit, p=p).astype(np.uint8), copies, div))
    for _ in range(int(rng.integers(0, 5))):    # exact long duplications: ties beyond any sort depth
        unit = int(rng.integers(300, 3000))
        plan.append((rng.choice(4, size=unit, p=p).astype(np.uint8), int(rng.integers(2, 7)), 0.0))
    jobs = [(f, d) for (f, c, d) in plan for _ in range(c)]
    order = rng.permutation(len(jobs))
    scale = max(1, (g - 2000) // max(1, sum(len(jobs[j][0]) for j in order) + len(jobs)))
    for j in order:
        f, d = jobs[j]
        cursor += int(rng.integers(0, 2 * scale))
        if cursor + len(f) >= g - 64:
            break
        u = f.copy()
        if d:
            mut = rng.random(len(u)) < d
            u[mut] = (u[mut] + rng.integers(1, 4, mut.sum())) & 3
        if rng.random() < 0.3:
            u = rc(u)
        codes[cursor: cursor + len(u)] = u
        cursor += len(u) + 1
    for f, c, d in plan:
        fams.append(f)
    # tandem repeats anywhere (they may overwrite family copies: one more source of divergence)
    for _ in range(int(rng.integers(0, 4))):
        unit = rng.integers(0, 4, int(rng.choice([1, 1, 2, 3, 5]))).astype(np.uint8)
        length = int(rng.choice([40, 200, 1500]))
        s = int(rng.integers(0, g - length - 1))
        codes[s: s + length] = np.resize(unit, length)
        fams.append(np.resize(unit, min(length, 120)).astype(np.uint8))
    # chromosomes and N runs (non-N coordinates, as the index stores them)
    n_chr = int(rng.integers(1, 6))
    chr_start = np.concatenate([[0], np.sort(rng.choice(np.arange(1000, g - 1000), n_chr - 1, replace=False))]).astype(np.uint64)
    n_runs = int(rng.integers(0, 6))
    keys = np.concatenate([[0], np.sort(rng.choice(np.arange(500, g - 500), n_runs, replace=False))]).astype(np.uint64)
    vals = np.concatenate([[int(rng.choice([0, 0, 17, 1000]))], rng.integers(0, 60000, n_runs)]).astype(np.uint64)
    return codes, fams, chr_start, keys, vals


def make_reads(rng, codes, fams, chr_start, keys, n, l_max=512, wide=False):
    g = len(codes)
    lens_pool = [list(range(12, 34)), [18, 22, 26, 30], [36], [50], [36, 50, 51, 52, 53, 64], [65, 100, 128], [100],
                 [129, 150, 192, 200], list(range(12, 201, 7))]
    if wide:  # (seeds from 100000 on: the draws of the earlier seeds stay what they were)
        lens_pool += [[12, 13, 14, 15, 16, 40], [200, 256, 257, 300, 384, 385, 512], [30, 64, 65, 128, 129, 192, 193, 320]]
    lens_ = lens_pool[int(rng.integers(0, len(lens_pool)))]
    lens_ = sorted(set(min(x, l_max) for x in lens_))
    mutate = float(rng.choice([0.0, 0.2, 0.5]))
    out = []

    def emit(r):
        if rng.random() < 0.5:
            r = rc(r)
        if rng.random() < mutate and len(r):
            r = r.copy()
            q = int(rng.integers(0, len(r)))
            r[q] = (r[q] + 1 + rng.integers(0, 3)) & 3
        out.append(bytes(b"ACGT"[c] for c in r))

    for _ in range(n):
        L = int(rng.choice(lens_))
        kind = rng.random()
        if kind < 0.6 or not fams:
            p = int(rng.integers(0, g - L))
            emit(codes[p:p + L])
        elif kind < 0.85:
            f = fams[int(rng.integers(0, len(fams)))]
            if len(f) >= 12:
                Lf = min(L, len(f))
                a = int(rng.integers(0, len(f) - Lf + 1))
                emit(f[a:a + Lf])
            else:
                emit(np.resize(f, L).astype(np.uint8))
        elif kind < 0.93:
            edges = np.concatenate([chr_start, keys, [0, g]])
            e = int(edges[int(rng.integers(0, len(edges)))])
            p = min(max(e - int(rng.integers(0, L + 1)), 0), g - L)
            emit(codes[p:p + L])
        else:
            emit(rng.integers(0, 4, L).astype(np.uint8))
    return out


# the rare paths (tests/test_gpu_rare_paths.py), drawn from seed 300000 on: results that do not fit a 16-byte entry
# (wide records, the ring bypass), the filter tables in global memory instead of LDS, host batches cut into sub-batches
RARE_KNOBS = {
    "SBWT_DEBUG_RING_WIDTH": [None, None, None, "0", "2", "40"],
    "SBWT_DEBUG_RING_CHR": [None, None, None, "0"],
    "SBWT_DEBUG_LDS_KEYS": [None, None, None, "0"],
    "SBWT_SUBBATCH": [None, None, "128", "1024"],
}


def pick(rng, table):
    env = {}
    for k, choices in table.items():
        v = choices[int(rng.integers(0, len(choices)))]
        if v is not None:
            env[k] = v
    return env


class Env:
    def __init__(self, env):
        self.env = env

    def __enter__(self):
        self.saved = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def first_bad(a, b):
    bad = np.nonzero(a != b)[0]
    return int(bad[0]) if bad.size else -1


def compare(tag, res, hits, lo, hi, nh, first, ohits, sa_mask=None):
    """-> list of problems (strings)"""
    out = []
    if not np.array_equal(res["nh"], nh):
        r = first_bad(res["nh"], nh)
        out.append("%s: nh of read %d: %d [%d, %d), oracle %d [%d, %d) (%d reads differ: %s)" % (
            tag, r, res["nh"][r], res["sa_lo"][r], res["sa_hi"][r], nh[r], lo[r], hi[r], (res["nh"] != nh).sum(),
            np.nonzero(res["nh"] != nh)[0][:8]))
        return out
    m = np.ones(len(nh), bool) if sa_mask is None else sa_mask
    for name, want in (("sa_lo", lo), ("sa_hi", hi)):
        if not np.array_equal(res[name][m], want[m]):
            r = int(np.nonzero(m)[0][first_bad(res[name][m], want[m])])
            out.append("%s: %s of read %d: %d, oracle %d" % (tag, name, r, res[name][r], want[r]))
    k = np.maximum(nh, 0)
    tot = int(k.sum())
    gi = np.repeat(res["first_hit"], k) + (np.arange(tot) - np.repeat(np.cumsum(k) - k, k))
    oi = np.repeat(first, k) + (np.arange(tot) - np.repeat(np.cumsum(k) - k, k))
    a, b = hits[gi.astype(np.int64)], ohits[oi.astype(np.int64)]
    if not np.array_equal(a, b):
        j = first_bad(a, b)
        r = int(np.searchsorted(np.cumsum(k), j, side="right"))
        out.append("%s: hit list of read %d (nh %d) differs at entry %d: %s, oracle %s"
                   % (tag, r, nh[r], j - int((np.cumsum(k) - k)[r]), a[j], b[j]))
    return out


def run_seed(seed, sb, oracle, n_reads=12000, verbose=True, override=None, beyond=False):
    """-> (problems, description).  override: knobs (name -> value, None = unset) that replace the seed's own draw"""
    rng = np.random.default_rng(1_000_003 * seed + 17)
    codes, fams, chr_start, keys, vals = make_genome(rng)
    g = len(codes)
    build_env = pick(rng, BUILD_KNOBS)
    pass_env = pick(rng, PASS_KNOBS)
    if seed >= 300000:
        pass_env.update(pick(rng, RARE_KNOBS))
    for k_, v_ in (override or {}).items():
        for env_ in (build_env, pass_env):
            env_.pop(k_, None)
        if v_ is not None:
            (build_env if k_ in BUILD_KNOBS else pass_env)[k_] = v_
    interval = int(rng.choice([64, 64, 64, 64, 16, 32, 128, 256]))
    if interval == 256 and pass_env.get("SBWT_SUBBATCH") == "128":
        # (every multi-hit read of such an index takes the depth-first kernel, one lane per read: launches of 128 reads
        # make a seed last minutes -- seed 301013: 196 s, green)
        pass_env["SBWT_SUBBATCH"] = "1024"
    # The k-ordering contract: a suffix order stopped at k bases answers patterns of up to k bases exactly, and the
    # locate extends a read of L bases by up to interval - 1 more: L + interval - 1 <= k.  (Beyond it upstream's walk
    # reports rows twice or not at all; the library follows it up to the raw interval size and flags the rest
    # SBWT_RF_TRUNCATED, include/sbwt_amd.h.)  `beyond` lifts the restriction.
    l_max = 512
    if "SBWT_DEBUG_SORT_DEPTH" in build_env and not beyond:
        k_ord = (int(build_env["SBWT_DEBUG_SORT_DEPTH"]) + 26) // 27 * 27
        l_max = min(512, k_ord - interval + 1)
        if l_max < 12:
            build_env.pop("SBWT_DEBUG_SORT_DEPTH")
            l_max = 512
    limit = int(rng.choice([1000, 1000, 1000, 0, 1, 7, 300, 5000] + ([-1] if seed >= 100000 else [])))
    strand = int(rng.choice([2, 2, 2, 0, 1]))
    desc = "seed %d: g %d, interval %d, -m %d, strand %d, reads <= %d, build %s, pass %s" % (seed, g, interval, limit, strand, l_max, build_env, pass_env)
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    host = np.concatenate([pack_codes(codes), np.zeros(4, np.uint32)])
    d_fwd = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_fwd), host.nbytes) == 0
    assert hip.hipMemcpy(d_fwd, host.ctypes.data, host.nbytes, 1) == 0
    t0 = time.time()
    with Env({**build_env, **pass_env}):
        ix, tables = sb.build_index_device(d_fwd.value, g, chr_start, keys, vals, interval=interval, device=0,
                                           want_index=True, want_tables=True)
    hip.hipFree(d_fwd)
    t_build = time.time() - t0
    T = tables.contents
    c4 = (C.c_uint64 * 4)(*T.c_function4)
    h = oracle.lib().sbo_index_wrap(T.interval, T.n, T.first_location, T.real_size, C.addressof(c4), T.c_functions,
                                    T.occ[0], T.occ[1], T.occ[2], T.occ[3], T.occ_len, T.jbwt_seq, T.fbwt,
                                    T.location_table, T.n_loc, chr_start.size, chr_start.ctypes.data, keys.size,
                                    keys.ctypes.data, vals.ctypes.data)
    oix = oracle.OracleIndex(h, keep=(c4, chr_start, keys, vals))
    problems = []
    try:
        reads = make_reads(rng, codes, fams, chr_start, keys, n_reads, l_max, wide=seed >= 100000)
        packed, lens = sb.pack_reads(reads)
        t0 = time.time()
        lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand, nthreads=8)
        t_oracle = time.time() - t0
        over = int((nh.astype(np.int64) > (hi.astype(np.int64) - lo.astype(np.int64))).sum())
        with Env({**build_env, **pass_env}):
            ctx = sb.Context(ix, len(reads))
        try:
            res, hits = ctx.map_batch(packed, lens, limit=limit, strand=strand)
            st = ctx.last_stats_ex()
            problems += compare("plain", res, hits, lo, hi, nh, first, ohits)
            for mode, tag in ((0, "compact"), (sb.MAP_HITS_ONLY, "hits-only")):
                rec, lists, wide = ctx.map_batch_compact(packed, lens, limit=limit, strand=strand, mode=mode)
                no_sa = ((rec["w1"] >> np.uint64(56)).astype(np.uint32) & sb.CR_NO_SA) != 0
                if mode == 0 and no_sa.any():
                    problems.append("compact: %d records without an SA interval" % no_sa.sum())
                res, hits = sb.compact_expand(rec, lists, wide)
                problems += compare(tag, res, hits, lo, hi, nh, first, ohits, sa_mask=~no_sa)
            if seed >= 200000:
                # the device-pointer entry points on the same batch: sbwt_map_batch_device (the ctx's) and
                # sbwt_map_batch_device_static (no ctx, allocates nothing), twice each with a capacity retry in between
                res0, hits0 = ctx.map_batch(packed, lens, limit=limit, strand=strand)
                n = len(reads)
                wpr = packed.shape[1] if packed.ndim == 2 else packed.size // n
                bufs = []

                def dev(nbytes):
                    p_ = C.c_void_p()
                    assert hip.hipMalloc(C.byref(p_), max(int(nbytes), 16)) == 0
                    bufs.append(p_)
                    return p_

                d_p, d_l = dev(packed.nbytes), dev(lens.nbytes)
                assert hip.hipMemcpy(d_p, packed.ctypes.data, packed.nbytes, 1) == 0
                assert hip.hipMemcpy(d_l, lens.ctypes.data, lens.nbytes, 1) == 0
                d_r = dev(n * sb.api.RESULT_DTYPE.itemsize)
                L_ = sb.api.lib()
                for tag in ("device", "static"):
                    cap = n  # too small whenever a list exists: the capacity report says what to come back with
                    used = C.c_uint64(0)
                    for _ in range(6):
                        d_h = dev(cap * sb.api.HIT_DTYPE.itemsize)
                        if tag == "device":
                            rc_ = L_.sbwt_map_batch_device(ctx._h, d_p.value, d_l.value, n, wpr, limit, strand, d_r.value, d_h.value, cap, C.byref(used))
                        else:
                            rc_ = L_.sbwt_map_batch_device_static(ix._h, None, d_p.value, d_l.value, n, wpr, limit, strand, d_r.value, d_h.value, cap, C.byref(used))
                        if rc_ != -7:
                            break
                        cap = int(used.value)
                    if rc_ != 0:
                        problems.append("%s: rc %d after capacity retries (%s)" % (tag, rc_, L_.sbwt_last_error().decode()))
                        continue
                    used = used.value
                    res1 = np.zeros(n, dtype=sb.api.RESULT_DTYPE)
                    hits1 = np.zeros(max(int(used), 1), dtype=sb.api.HIT_DTYPE)
                    assert hip.hipMemcpy(res1.ctypes.data, d_r, res1.nbytes, 2) == 0
                    assert hip.hipMemcpy(hits1.ctypes.data, d_h, int(used) * sb.api.HIT_DTYPE.itemsize, 2) == 0
                    problems += compare(tag, res1, hits1, lo, hi, nh, first, ohits)
                for b_ in bufs:
                    hip.hipFree(b_)
        finally:
            ctx.close()
        if seed >= 100000:
            # the host pipeline on the same reads: FASTQ in, SAM out (sbwt_map_fastq; position2sam, sam.hpp), bytes and
            # counters against the oracle's; random batch size and formatter threads
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                fq, sam_a, sam_b = os.path.join(td, "r.fq"), os.path.join(td, "a.sam"), os.path.join(td, "b.sam")
                sub = reads[: int(rng.choice([1, 700, 4000]))]
                with open(fq, "wb") as f:
                    for k, s_ in enumerate(sub):
                        q = bytes(33 + (k + j) % 41 for j in range(len(s_)))
                        f.write(b"@r%d\n%s\n+\n%s\n" % (k, s_, q))
                env = {}
                b_ = rng.choice(["", "64", "1000"])
                if b_:
                    env["SBWT_FASTQ_BATCH"] = str(b_)
                nt = int(rng.choice([1, 2, 5, 16]))
                with Env({**build_env, **pass_env, **env}):
                    st = ix.map_fastq(fq, sam_a, limit=limit, strand=strand, n_threads=nt)
                rc, ost = oix.map_fastq(fq, sam_b, limit=limit, strand=strand)
                if rc != 0 or list(st) != list(ost):
                    problems.append("map_fastq: counters %s, oracle rc %d %s" % (list(st), rc, list(ost)))
                elif open(sam_a, "rb").read() != open(sam_b, "rb").read():
                    la, lb = open(sam_a, "rb").read().split(b"\n"), open(sam_b, "rb").read().split(b"\n")
                    k = next((i for i, (x, y) in enumerate(zip(la, lb)) if x != y), min(len(la), len(lb)))
                    problems.append("map_fastq (%d reads, %d threads, batch %s): SAM line %d differs: %r, oracle %r"
                                    % (len(sub), nt, b_ or "default", k, la[k][:200] if k < len(la) else None, lb[k][:200] if k < len(lb) else None))
        if verbose:
            print("%s | hash %d text %d | nh>1 %d, nh>64 %d, max nh %d, dead %d, lists longer than the interval %d | build %.1f s, oracle %.1f s | %s" % (
                desc, ix.info().hash_order, ix.info().hash_text, (nh > 1).sum(), (nh > 64).sum(), nh.max(), (nh == 0).sum(),
                over, t_build, t_oracle, "ok" if not problems else "MISMATCH"), flush=True)
            for p_ in problems:
                print("    " + p_, flush=True)
    finally:
        oix.close()
        ix.close()
        sb.host_tables_free(tables)
    return problems, desc + " | lists longer than the interval: %d" % over


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:20", help="a:b (half-open) or a comma-separated list")
    ap.add_argument("--reads", type=int, default=12000)
    ap.add_argument("--seconds", type=float, default=1e9, help="stop starting new seeds after this long")
    ap.add_argument("--beyond", action="store_true", help="reads longer than the suffix order's depth allows (see run_seed)")
    ap.add_argument("--set", default="", help="NAME=VALUE,NAME= ... : knobs forced (empty value: unset) for every seed")
    a = ap.parse_args()
    if ":" in a.seeds:
        lo_, hi_ = a.seeds.split(":")
        seeds = range(int(lo_), int(hi_))
    else:
        seeds = [int(s) for s in a.seeds.split(",")]
    import oracle_lib
    import sbwt_amd as sb
    oracle_lib.build()
    sb.device_count()
    t0 = time.time()
    bad = []
    done = 0
    for s in seeds:
        if time.time() - t0 > a.seconds:
            break
        ov = {kv.split("=")[0]: (kv.split("=")[1] or None) for kv in a.set.split(",") if kv}
        problems, desc = run_seed(s, sb, oracle_lib, a.reads, override=ov, beyond=a.beyond)
        done += 1
        if problems:
            bad.append(s)
    print("fuzz: %d seeds, %d with mismatches %s, %.0f s" % (done, len(bad), bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
