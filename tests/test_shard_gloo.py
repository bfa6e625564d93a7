"""CPU, world_size 2 over gloo: the N > 1 path of the mapper (sbwt_amd/shard.py).

Reads shard in contiguous ranges, every rank maps its own range against its own copy of the index, the only
exchange is a host-side gather of SAM bytes and counters in rank order.  There is no GPU here, so the
per-shard mapper is the CPU oracle (as the checker standing in for a rank's device); what is under test is
the sharding, the gather order, the counter reduction and the max-over-ranks timing that bench.py uses.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from sbwt_amd import shard
import oracle_lib
prefix, fq, outdir = sys.argv[2], sys.argv[3], sys.argv[4]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lines = open(fq).read().split("\n")
recs = ["\n".join(lines[i:i + 4]) + "\n" for i in range(0, len(lines) - 3, 4)]
a, b = shard.shard_range(len(recs), rank, world)
part = os.path.join(outdir, "part%d.fq" % rank)
open(part, "w").write("".join(recs[a:b]))
oix = oracle_lib.OracleIndex.load(prefix)
sam = os.path.join(outdir, "part%d.sam" % rank)
calls = []
def step():
    calls.append(1)
    if rank == 1:
        time.sleep(0.05)          # the slow rank must set the reported time
    return oix.map_fastq(part, sam)
elapsed = shard.timed_steps(step, steps=2, warmup=1, dist=dist)
rc, st = oix.map_fastq(part, sam)
assert rc == 0
total = shard.sum_counters(st, dist)
parts = shard.gather_bytes(open(sam, "rb").read(), dist, dst=0)
if rank == 0:
    open(os.path.join(outdir, "all.sam"), "wb").write(b"".join(parts))
    json.dump({"stats": total, "elapsed": elapsed, "calls": len(calls), "ranges": [shard.shard_range(len(recs), r, world) for r in range(world)]},
              open(os.path.join(outdir, "out.json"), "w"))
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_tiles_input():
    from sbwt_amd import shard
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            rs = [shard.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[k][1] == rs[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


@pytest.mark.parametrize("g", ["g1", "g2"])
def test_two_ranks_reproduce_the_unsharded_sam(g, golden_dir, golden_indexes, tmp_path):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(worker), ROOT, golden_indexes[g],
                                       os.path.join(golden_dir, g + ".fq"), str(tmp_path)], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    out = json.load(open(tmp_path / "out.json"))
    gold = open(os.path.join(golden_dir, g + ".sam"), "rb").read()
    assert open(tmp_path / "all.sam", "rb").read() == gold          # rank order == input order
    n_reads = 20 if g == "g1" else 8
    assert out["stats"][0] == n_reads and out["stats"][4] == gold.count(b"\n")
    assert out["ranges"] == [[0, n_reads // 2], [n_reads // 2, n_reads]]
    assert out["calls"] == 3                                          # warmup 1 + steps 2
    assert out["elapsed"] >= 0.1                                      # max over ranks: rank 1 sleeps 2 x 50 ms
