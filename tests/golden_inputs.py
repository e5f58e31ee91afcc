"""Deterministic synthetic inputs for the golden fixtures and benches (test infrastructure).

SplitMix64 stream seeded 0x5EED_0000_0000_0000 + case id; each word is masked to the bit length of
q and kept when < q (rejection), matching the reference's per-modulus Uniform::new(0, q) sampling
in distribution (primus_distr/src/common.rs:244-263) -- SURVEY.md §8d.
"""
import hashlib

import numpy as np

SEED_BASE = 0x5EED_0000_0000_0000
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix_words(seed: int, start: int, count: int) -> np.ndarray:
    """Words start .. start+count of the SplitMix64 stream with the given seed."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + _GAMMA * np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def splitmix_uniform(case_id: int, q: int, count: int) -> np.ndarray:
    """First `count` accepted draws in [0, q) of stream SEED_BASE + case_id."""
    mask = np.uint64((1 << int(q).bit_length()) - 1)
    out = np.empty(count, np.uint64)
    have, pos = 0, 0
    while have < count:
        chunk = max(1024, 2 * (count - have))
        w = splitmix_words(SEED_BASE + case_id, pos, chunk) & mask
        pos += chunk
        w = w[w < np.uint64(q)][:count - have]
        out[have:have + w.size] = w
        have += w.size
    return out


def splitmix_rns(case_id: int, moduli, n: int, batch: int = 1) -> np.ndarray:
    """batch RNS polynomials, modulus-major inside each element; limb (e, r) uses its own stream
    SEED_BASE + (case_id << 20) + e*len(moduli) + r so any element can be regenerated alone."""
    L = len(moduli)
    return np.concatenate([splitmix_uniform((case_id << 20) + e * L + r, q, n)
                           for e in range(batch) for r, q in enumerate(moduli)])


def digest(words: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(words, dtype="<u8").tobytes()).hexdigest()
