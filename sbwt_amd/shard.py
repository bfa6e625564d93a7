"""Read sharding across ranks (one process per GPU) for the mapping path.

Upstream parallelises over independent 1000-read groups on host threads (job_distributer.hpp:37-90) and
serialises the output under a mutex; reads never interact.  Across GPUs the same independence holds: the
index is replicated, every rank maps a contiguous range of the input, and the only exchange is a host-side
gather of the per-shard SAM bytes and counters, in rank order (= input order).  There is no data-path
collective; torch.distributed (RCCL on GPUs, gloo in the CPU tests) carries the barrier, the max-over-ranks
of the elapsed time and the small host gathers.
"""
import time


def shard_range(n_items, rank, world):
    """Contiguous range [a, b) of rank `rank`; ranges tile [0, n_items) in rank order, sizes differ by <= 1.
    The arithmetic is the library's own (sbwt_shard_range, the one sbwt_map_fastq_multi cuts its batches with)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    from . import api
    return api.shard_range(n_items, world, rank)


def barrier(dist, sync=None):
    if dist is not None:
        dist.barrier()
    if sync is not None:
        sync()


def timed_steps(step, steps, warmup, dist=None, sync=None, device=None):
    """bench.py contract: `warmup` untimed steps, then exactly `steps` timed ones bracketed by a barrier and a
    device synchronise on both sides; returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    barrier(dist, sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(dist, sync)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        on_cpu = device is None or dist.get_backend() == "gloo"
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_cpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def sum_counters(counters, dist=None):
    """Element-wise sum over ranks of the upstream statistics tuple (total, mapped, single, multi, records)."""
    if dist is None:
        return [int(x) for x in counters]
    import torch
    t = torch.tensor([int(x) for x in counters], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.cpu().tolist()]


def gather_bytes(payload, dist=None, dst=0):
    """Host-side gather of per-shard byte strings (SAM records) to rank `dst`, returned in rank order there,
    None elsewhere."""
    if dist is None:
        return [payload]
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(payload, out, dst=dst)
    return out
