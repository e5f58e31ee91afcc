"""Deterministic prime helpers shared by the parity tests (no third-party number theory)."""


def is_prime(n: int) -> bool:
    """Miller-Rabin with the first twelve prime bases: deterministic below 3.3e24."""
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def random_ntt_primes(rng, count, bits, log_n):
    """`count` distinct primes of exactly `bits` bits with 2^(log_n+1) | q - 1."""
    step = 1 << (log_n + 1)
    lo, hi = (1 << (bits - 1)) // step + 1, (1 << bits) // step
    assert hi - lo >= 64 * count, "not enough candidates: raise `bits`"
    out = []
    for _ in range(200000):
        k = int(rng.integers(lo, hi))
        q = k * step + 1
        if q.bit_length() == bits and q not in out and is_prime(q):
            out.append(q)
            if len(out) == count:
                return out
    raise AssertionError("prime search exhausted")


def ntt_primes_below(count, bits, log_n):
    """The `count` largest primes below 2^bits with 2^(log_n+1) | q - 1, descending (a fixed, seed-free list)."""
    step = 1 << (log_n + 1)
    q = ((1 << bits) - 1) // step * step + 1
    out = []
    while len(out) < count:
        if q < (1 << bits) and is_prime(q):
            out.append(q)
        q -= step
        assert q > (1 << (bits - 1)), "ran out of candidates"
    return out
