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


def isolated(fn):
    """Run a GPU test in a child pytest process of its own.

    For tests that hipHostRegister heap memory (numpy arrays) on purpose.  The pages of such a block go back to malloc when
    the array dies and are handed out again — to arrays that torch then copies as PAGEABLE memory — and the runtime / driver
    defect located in round 5 (profiles/r05_experiments.txt item 7, tools/microbench12_register_hazard.hip: pages that have
    been both registered and the source of pageable copies) can then surface anywhere later in the same process: once in
    about twenty full-suite runs a later, unrelated test hung on a device-to-host copy (round 6 soak,
    profiles/r06_experiments.txt item 7).  In a child process the registrations die with it and the main test process never
    registers heap memory."""
    import functools
    import inspect
    import os
    import subprocess
    import sys

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("PFHE_TEST_ISOLATED") == "1":
            return fn(*args, **kwargs)
        path = inspect.getsourcefile(fn)
        env = dict(os.environ, PFHE_TEST_ISOLATED="1")
        r = subprocess.run([sys.executable, "-m", "pytest", f"{path}::{fn.__name__}", "-q", "-x", "-m", "gpu",
                            "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(path))))
        assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-4000:] + r.stderr[-2000:])

    return wrapper
