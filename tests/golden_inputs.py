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


def _mulhi64(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    m = np.uint64(0xFFFFFFFF)
    s = np.uint64(32)
    a0, a1, b0, b1 = a & m, a >> s, b & m, b >> s
    with np.errstate(over="ignore"):
        mid = (a0 * b0 >> s) + (a1 * b0 & m) + (a0 * b1 & m)
        return a1 * b1 + (a1 * b0 >> s) + (a0 * b1 >> s) + (mid >> s)


def fill_uniform_words(seed: int, start: int, count: int, moduli, poly_len: int) -> np.ndarray:
    """Host model of pfhe_fill_uniform_dev (include/pfhe.h): words start .. start+count of a buffer filled with `seed`,
    word i = floor(splitmix64(seed, i) * q / 2^64) with q = moduli[(i / poly_len) % len(moduli)]."""
    i = np.arange(start, start + count, dtype=np.uint64)
    q = np.asarray(moduli, dtype=np.uint64)[(i // np.uint64(poly_len)) % np.uint64(len(moduli))]
    return _mulhi64(splitmix_words(seed, start, count), q)
