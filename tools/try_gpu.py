"""First-contact GPU check: product library vs oracle on the committed golden sets."""
import hashlib, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import sbwt_amd

def fq_reads(path):
    lines = open(path).read().split("\n")
    return [lines[i + 1].strip() for i in range(0, len(lines) - 3, 4)]

def compare(name, ix, oix, reads, limit=1000, strand=2):
    packed, lens = sbwt_amd.pack_reads(reads)
    ctx = sbwt_amd.Context(ix, max(len(reads), 1))
    res, hits = ctx.map_batch(packed, lens, limit=limit, strand=strand)
    lo, hi, nh, first, ohits = oix.map_batch(reads, limit=limit, strand=strand)
    bad = 0
    for r in range(len(reads)):
        ok = res["sa_lo"][r] == lo[r] and res["sa_hi"][r] == hi[r] and res["nh"][r] == nh[r]
        if ok and nh[r]:
            a = hits[int(res["first_hit"][r]): int(res["first_hit"][r]) + int(nh[r])]
            b = ohits[int(first[r]): int(first[r]) + int(nh[r])]
            ok = bool(np.array_equal(a, b))
        if not ok:
            bad += 1
            if bad <= 5:
                print("  MISMATCH read", r, reads[r][:30], "gpu", res[r], "oracle", lo[r], hi[r], nh[r])
    print("%s: %d reads, %d mismatches, timing %s" % (name, len(reads), bad, ctx.last_timing()))
    ctx.close()
    return bad

tmp = tempfile.mkdtemp(prefix="sbwt_try_")
total = 0
for g in ("g1", "g2"):
    fa = os.path.join(ROOT, "tests/golden/%s.fa" % g); fq = os.path.join(ROOT, "tests/golden/%s.fq" % g)
    prefix = os.path.join(tmp, g)
    oracle_lib.build_index(fa, prefix)
    t = time.time(); ix = sbwt_amd.Index.load(prefix); print("index load %.2fs, HBM %.1f MB" % (time.time() - t, ix.info().device_bytes / 1e6))
    oix = oracle_lib.OracleIndex.load(prefix)
    reads = [r for r in fq_reads(fq) if len(r) >= 12]
    for strand in (2, 0, 1):
        total += compare("%s strand=%d" % (g, strand), ix, oix, reads, strand=strand)
    total += compare(g + " limit=2", ix, oix, reads, limit=2)
    # random substrings of the genome (both strands), several lengths, plus mutated reads
    seq = "".join(l.strip() for l in open(fa) if not l.startswith(">")).upper()
    rng = np.random.default_rng(1)
    comp = str.maketrans("ACGT", "TGCA")
    rs = []
    for L in (12, 13, 14, 15, 20, 31, 50, 63, 64, 65, 100, 129):
        for _ in range(300):
            p = int(rng.integers(0, len(seq) - L)); s = seq[p:p + L]
            if "N" in s: continue
            if rng.random() < 0.5: s = s.translate(comp)[::-1]
            if rng.random() < 0.3:
                k = int(rng.integers(0, L)); s = s[:k] + "ACGT"[int(rng.integers(0, 4))] + s[k + 1:]
            rs.append(s)
    total += compare(g + " random", ix, oix, rs)
    sam = os.path.join(tmp, g + ".sam")
    st = ix.map_fastq(fq, sam)
    got = open(sam, "rb").read(); want = open(os.path.join(ROOT, "tests/golden/%s.sam" % g), "rb").read()
    print(g, "map_fastq stats", st, "SAM identical:", got == want)
    total += got != want
    ix.close(); oix.close()
print("TOTAL MISMATCHES", total)
sys.exit(1 if total else 0)
