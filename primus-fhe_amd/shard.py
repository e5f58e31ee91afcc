"""Batch sharding across the GPUs of one node.

Every polynomial / ciphertext of a batch is independent: no step of the NTT, the pointwise ops or
the external product mixes batch elements, so the multi-GPU path has NO data-path collective —
each rank (one process per GPU) owns a contiguous range of the batch, tables and a shared GGSW are
replicated per device.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for
the timing barrier and for reducing the per-rank wall time to its maximum.
"""
from __future__ import annotations

import time


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [begin, end) of `total` units owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def timed_steps(step, steps: int, warmup: int, sync, dist=None, device=None):
    """Contract of bench.py: W untimed steps, then exactly K steps bracketed by barrier + sync on
    both sides; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def strong_scaling_leg(total: int, world: int, rank: int, run_shard, steps: int, warmup: int, sync, dist=None,
                       device=None):
    """BASELINE config 5: a FIXED job of `total` independent units (ciphertexts) split over the ranks in contiguous
    ranges (shard_range) with no data-path collective.  `run_shard(begin, end)` processes this rank's range once; one
    step = every rank doing that.  Returns a dict with the whole-job rate (total * steps / max-over-ranks time)."""
    begin, end = shard_range(total, world, rank)
    dt = timed_steps(lambda: run_shard(begin, end), steps, warmup, sync, dist, device)
    return {"scaling": "strong", "batch_total": total, "n_gpus": world, "range_of_rank0": [begin, end] if rank == 0 else None,
            "units_this_rank": end - begin, "steps": steps, "seconds": dt, "value": total * steps / dt}
