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


_SPLITMIX_GAMMA = 0x9E3779B97F4A7C15


def job_seed(seed: int, word_offset: int) -> int:
    """Seed under which pfhe_fill_uniform_dev writes, at word 0 of a buffer, what it would have written at word
    `word_offset` of a buffer filled with `seed`: the fill is counter-based (word i = splitmix64(seed, i) scaled into
    [0, q_r)), so a shard of a job's synthetic input depends only on its position in the job, not on how the job is
    split over the ranks.  `word_offset` must be a multiple of one RNS polynomial (L * N words)."""
    return (seed + word_offset * _SPLITMIX_GAMMA) & 0xFFFFFFFFFFFFFFFF


def fill_job_shard(lib, device: int, dst_ptr: int, begin_unit: int, units: int, unit_words: int, moduli, poly_len: int,
                   seed: int, stream=None) -> None:
    """Synthetic input of units [begin_unit, begin_unit + units) of a job whose unit (ciphertext, RNS polynomial) is
    `unit_words` words, written to device memory at dst_ptr.  Identical to the same units of the job filled in one go."""
    import ctypes as C

    import numpy as np
    mods = np.ascontiguousarray(moduli, dtype=np.uint64)
    if unit_words % (len(mods) * poly_len):
        raise ValueError("a unit must be whole RNS polynomials")
    rc = lib.pfhe_fill_uniform_dev(device, C.c_void_p(dst_ptr), units * unit_words,
                                   mods.ctypes.data_as(C.POINTER(C.c_uint64)), len(mods), poly_len,
                                   job_seed(seed, begin_unit * unit_words), stream)
    if rc != 0:
        raise RuntimeError("pfhe_fill_uniform_dev failed: %d" % rc)


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


def device_identity(index=None) -> dict:
    """What tells two GPUs apart: PCI domain:bus:device and the UUID of visible device `index` (None: a rank without a
    GPU — the CPU tests — reports its host and process id instead)."""
    import os
    import socket
    if index is None:
        return {"device": None, "host": socket.gethostname(), "pid": os.getpid()}
    import torch
    pr = torch.cuda.get_device_properties(index)
    pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    return {"device": index, "pci": pci, "uuid": str(getattr(pr, "uuid", "")), "name": pr.name, "host": socket.gethostname(),
            "pid": os.getpid()}


def dist_evidence(dist, identity: dict) -> dict:
    """What the process group ACTUALLY saw, for the bench line: backend and world size as torch.distributed reports them
    (not the launcher's environment) and the identity of every rank's device, gathered over the group — "N ranks on N
    distinct devices over RCCL" is then checkable from the line alone.  Collective: call on every rank."""
    world = dist.get_world_size()
    seen = [None] * world
    dist.all_gather_object(seen, dict(identity, rank=dist.get_rank()))
    keys = [(d.get("host"), d.get("pci") or d.get("pid")) for d in seen]
    return {"backend": dist.get_backend(), "world_size": world, "devices": seen,
            "distinct_devices": len(set(keys))}


def strong_scaling_leg(total: int, world: int, rank: int, run_shard, steps: int, warmup: int, sync, dist=None,
                       device=None):
    """BASELINE config 5: a FIXED job of `total` independent units (ciphertexts) split over the ranks in contiguous
    ranges (shard_range) with no data-path collective.  `run_shard(begin, end)` processes this rank's range once; one
    step = every rank doing that.  Returns a dict with the whole-job rate (total * steps / max-over-ranks time)."""
    begin, end = shard_range(total, world, rank)
    dt = timed_steps(lambda: run_shard(begin, end), steps, warmup, sync, dist, device)
    return {"scaling": "strong", "batch_total": total, "n_gpus": world, "range_of_rank0": [begin, end] if rank == 0 else None,
            "units_this_rank": end - begin, "steps": steps, "seconds": dt, "value": total * steps / dt}


def run_on_devices(devices, total: int, make_worker):
    """The other arrangement SURVEY.md §8e sketches: ONE process, one host thread + stream per device.  `devices` lists a
    device index per shard (a device may appear more than once); shard r gets units shard_range(total, len(devices), r).
    `make_worker(rank, device, begin, end)` is called INSIDE shard r's thread and returns that shard's result; every
    handle a worker creates carries its device, and every C-ABI entry point switches to the handle's device for the
    call, so the workers need no device bookkeeping of their own.  Returns the results in rank order; the first
    exception of any worker is re-raised.  (The repository's measured path is one process per GPU — bench.py; this helper
    and tests/test_gpu_multi_device.py exist so that the single-process form is exercised wherever two devices are.)"""
    import threading

    world = len(devices)
    results, errors = [None] * world, []

    def body(r):
        try:
            b, e = shard_range(total, world, r)
            results[r] = make_worker(r, devices[r], b, e)
        except BaseException as exc:  # noqa: BLE001 - re-raised below
            errors.append(exc)

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
