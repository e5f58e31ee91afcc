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
