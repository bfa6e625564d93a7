import sys, os; sys.path.insert(0,'/root/repo')
import torch, bench, sbwt_amd
dev=torch.device('cuda',0); torch.cuda.set_device(0)
lens_chr, g, chr_start, npos_key, npos_val = bench.synth_tables(1.0)
codes, packed_genome = bench.make_genome(torch, dev, g, seed=38)
ix,_ = sbwt_amd.build_index_device(packed_genome.data_ptr(), g, chr_start, npos_key, npos_val, interval=64, device=0, want_index=True, want_tables=False)
del packed_genome, codes; torch.cuda.empty_cache()
print('chained probe', ix.probe_random_access(chains_per_lane=2)[0]/1e9)
for v, what in ((0,'read 16 B + line + write 16 B'),(1,'read 16 B + line'),(2,'line + write 16 B'),(3,'line only (independent)'),(4,'read 16 B + line + write 8 B'),(5,'as 0 with non-temporal record loads and result stores')):
    os.environ['SBWT_PROBE_VARIANT']=str(v)
    print(v, what, [round(ix.probe_gather_stream(100_000_000, w)/1e9,2) for w in (4,8)], flush=True)
