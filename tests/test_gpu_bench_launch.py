"""GPU: bench.py's own launch paths, rehearsed on the one device of the box so that the driver's multi-GPU runs work
the first time a node exists: `--gpus 2` started plainly (bench.py spawns the ranks itself, fresh child processes,
before it touches the GPU; BENCH_SHARE_DEVICE=1 puts every rank on device 0 with gloo for the barrier), in the default
weak-scaling mode and in the strong-scaling mode of BASELINE.json configs[3] (`--config 4`: one input cut by
sbwt_shard_range)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, timeout=420):
    env = dict(os.environ, BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_launch():
    d = run_bench(["--gpus", "2", "--scale", "0.01", "--reads", "200000", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["value"] > 0 and d["config"]["reads_per_gpu"] == 200000
    assert d["detail"]["origin_check"]["mismatches"] == 0


def test_two_ranks_strong_scaling_config4_launch():
    one = run_bench(["--config", "4", "--gpus", "1", "--scale", "0.01", "--reads", "600001", "--chunk", "250000", "--steps", "2"])
    two = run_bench(["--config", "4", "--gpus", "2", "--scale", "0.01", "--reads", "600001", "--chunk", "250000", "--steps", "2"])
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["value"] > 0
        assert d["config"]["total_reads"] == 600001 and d["detail"]["reads_all_ranks"] == 600001
        # error-free reads: all of a last chunk map, the few that span a contig or an N gap excepted
        assert d["detail"]["mapped_in_last_chunks"] >= 0.99 * d["detail"]["reads_in_last_chunks"]
    assert two["config"]["reads_this_rank"] == 300001      # rank 0's contiguous share (sbwt_shard_range)
