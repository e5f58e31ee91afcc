"""Helpers shared by the -m gpu parity tests (torch is only plumbing for device memory)."""
import numpy as np


def to_dev(a: np.ndarray):
    import torch
    assert a.dtype == np.uint64
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def to_host(t) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint64)


def rand_mod(rng, q, n):
    return rng.integers(0, q, n, dtype=np.uint64)


def rand_rns(rng, moduli, n, batch=1):
    """batch RNS polynomials, modulus-major inside each element."""
    return np.concatenate([rand_mod(rng, q, n) for _ in range(batch) for q in moduli])


def usable_cores() -> int:
    """Cores this process may use (affinity mask and cgroup quota)."""
    import os
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def oracle_map(fn, host: np.ndarray, unit: int, threads: int | None = None):
    """Apply fn (an in-place oracle method; ctypes releases the GIL) to every contiguous run of `unit` words of
    `host`, spread over all usable cores."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or usable_cores()
    units = host.size // unit
    per = max(1, (units + threads - 1) // threads)
    chunks = [host[i * unit:min(units, i + per) * unit] for i in range(0, units, per)]
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fn, chunks))
