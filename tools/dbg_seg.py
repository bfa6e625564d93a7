import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sbwt_amd as sb
import oracle_lib
g = os.path.join(ROOT, "tests", "golden")
prefix = "/tmp/dbg_g2"
sb.build_index(os.path.join(g, "g2.fa"), prefix)
ix = sb.Index.load(prefix)
oix = oracle_lib.OracleIndex.load(prefix)
reads = [l.strip().encode() for i, l in enumerate(open(os.path.join(g, "g2.fq"))) if i % 4 == 1]
packed, lens = sb.pack_reads(reads)
lo, hi, nh, first, ohits = oix.map_batch(reads)
ctx = sb.Context(ix, len(reads))
for mode in ("plain", "compact", "hits"):
    if mode == "plain":
        res, hits = ctx.map_batch(packed, lens)
    else:
        rec, lists, wide = ctx.map_batch_compact(packed, lens, mode=sb.MAP_HITS_ONLY if mode == "hits" else 0)
        res, hits = sb.compact_expand(rec, lists, wide)
    for r in range(len(reads)):
        na, nb = int(res["nh"][r]), int(nh[r])
        fa, fb = int(res["first_hit"][r]), int(first[r])
        la = hits[fa: fa + max(na, 0)]
        lb = ohits[fb: fb + max(nb, 0)]
        if na != nb or not np.array_equal(la, lb):
            print(mode, "read", r, "len", lens[r], "sa", lo[r], hi[r], "gpu:", na, la.tolist(), "oracle:", nb, lb.tolist())
print("done", len(reads))
