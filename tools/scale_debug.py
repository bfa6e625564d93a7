import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, sbwt_amd
os.environ["SBWT_DEBUG"] = "1"
dev = torch.device("cuda", 0)
for scale in [float(x) for x in sys.argv[1:]]:
    lens_chr, g, cs, nk, nv = bench.synth_tables(scale)
    codes, packed = bench.make_genome(torch, dev, g, 38)
    torch.cuda.synchronize()
    # verify packing of a few random positions
    idx = torch.randint(0, g, (1000,), device=dev)
    w = packed[idx >> 4].to(torch.int64) & 0xffffffff
    got = (w >> (2 * (idx & 15))) & 3
    print("scale", scale, "g", g, "packing ok:", bool((got == codes[idx].to(torch.int64)).all()), flush=True)
    t = time.time()
    ix, _ = sbwt_amd.build_index_device(packed.data_ptr(), g, cs, nk, nv, want_index=True)
    print("build %.1fs" % (time.time() - t), "first_location", ix.info().first_location, flush=True)
    L = 50; wpr = 4; n = 1000000
    pk, ln = bench.make_reads(torch, dev, codes, g, n, L, 50, wpr)
    res = torch.empty(n * 32, dtype=torch.uint8, device=dev); hits = torch.empty((n + 1024) * 16, dtype=torch.uint8, device=dev)
    ctx = sbwt_amd.Context(ix, n)
    nh = ctx.map_batch_device(pk.data_ptr(), ln.data_ptr(), n, wpr, res.data_ptr(), hits.data_ptr(), n + 1024)
    r = np.frombuffer(res.cpu().numpy().tobytes(), dtype=sbwt_amd.api.RESULT_DTYPE)
    print("hits", nh, "nonempty intervals", int((r["sa_hi"] > r["sa_lo"]).sum()), "timing", ctx.last_timing(), flush=True)
    ctx.close(); ix.close(); del codes, packed, pk, ln, res, hits
    torch.cuda.empty_cache()
