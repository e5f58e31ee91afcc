"""Pure-Python big-integer ground truth for small cases (test infrastructure).

Independent of oracle/ (no shared code): used to pin the C restatement and, through it, the
HIP path.  Everything here is textbook mathematics on Python ints:
  * minimal primitive 2N-th root (definition used by primus_ntt/src/root.rs:103-125),
  * direct evaluation of the negacyclic NTT output ordering
    out[i] = a(psi^(2*brv(i)+1))  (primus_ntt/src/ntt/prime64/table.rs:580-589),
  * schoolbook product mod (X^N + 1, q)  (primus_poly/src/poly/mul.rs:107-134),
  * CRT composition, balanced gadget digits (primus_decompose/src/big_integer/*),
  * the RNS gadget external product as sum_i sum_j digit_ij (*) key_ij.
"""
from __future__ import annotations

import numpy as np

# Moduli used by the reference's own tests + the survey's N=2^16-capable primes.
REF_TEST_PRIMES = [132120577, 536813569, 562949953392641, 1152921504606830593,
                   1073692673, 1125899906826241, 1125899906629633]
Q62 = 4611686018425815041
Q61 = [2305843009211596801, 2305843009210023937, 2305843009208713217]
Q60 = 1152921504606584833
# SURVEY.md Appendix A.1 (computed there with sympy, independently of this repo's code)
SURVEY_MIN_ROOTS = {
    (132120577, 10): 73993,
    (1125899906826241, 12): 46909545429,
    (4611686018425815041, 10): 1205379444979587,
    (2305843009211596801, 14): 117297622845463,
    (2305843009211596801, 16): 25740574174379,
    (2305843009210023937, 16): 11864589261338,
    (2305843009208713217, 16): 14354131908784,
}


def brv(i: int, bits: int) -> int:
    r = 0
    for b in range(bits):
        r |= ((i >> b) & 1) << (bits - 1 - b)
    return r


def minimal_primitive_root(log_degree: int, q: int) -> int:
    """Smallest element of multiplicative order exactly 2^log_degree modulo prime q."""
    degree = 1 << log_degree
    if (q - 1) % degree:
        raise ValueError("no primitive root")
    g = None
    for r in range(2, 2000):
        w = pow(r, (q - 1) // degree, q)
        if pow(w, degree // 2, q) == q - 1:
            g = w
            break
    assert g is not None
    best, cur, g2 = g, g, g * g % q
    for _ in range(degree // 2):
        best = min(best, cur)
        cur = cur * g2 % q
    return best


def ntt_direct(a, q: int, log_n: int, psi: int | None = None):
    """out[i] = sum_j a[j] * psi^((2*brv(i)+1)*j) mod q — O(N^2), small N only."""
    n = 1 << log_n
    psi = psi or minimal_primitive_root(log_n + 1, q)
    a = [int(x) for x in a]
    out = []
    for i in range(n):
        e = 2 * brv(i, log_n) + 1
        w = pow(psi, e, q)
        acc, p = 0, 1
        for j in range(n):
            acc = (acc + a[j] * p) % q
            p = p * w % q
        out.append(acc)
    return np.array(out, dtype=np.uint64)


def negacyclic_mul(a, b, q: int):
    n = len(a)
    a = [int(x) for x in a]
    b = [int(x) for x in b]
    c = [0] * n
    for i in range(n):
        if a[i] == 0:
            continue
        for j in range(n):
            k = i + j
            if k < n:
                c[k] += a[i] * b[j]
            else:
                c[k - n] -= a[i] * b[j]
    return [x % q for x in c]


def limbs_to_int(limbs) -> int:
    v = 0
    for i, x in enumerate(limbs):
        v |= int(x) << (64 * i)
    return v


def int_to_limbs(v: int, n: int):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def crt_compose(residues, moduli) -> int:
    Q = 1
    for m in moduli:
        Q *= m
    v = 0
    for r, m in zip(residues, moduli):
        P = Q // m
        v += int(r) * pow(P, -1, m) % m * P
    return v % Q


class Gadget:
    """Balanced base-2^log_basis digits of v in [0,Q), least significant first.

    Mathematical statement of primus_decompose's approximate signed decomposition
    (basis.rs:40-211, common.rs:275-285): with bits = bitlen(Q), ell = bits // log_basis (or the
    requested prefix), drop = bits - ell*log_basis, a value v is first mapped to
    v' = v + (2^bits - Q) when v >= threshold, then digit_j is the balanced digit of
    round-half-up(v' / 2^drop) in base B, the final carry being discarded.
    """

    def __init__(self, moduli, log_basis: int, reverse_length: int | None = None):
        self.moduli = list(moduli)
        self.Q = 1
        for m in self.moduli:
            self.Q *= m
        self.bits = self.Q.bit_length()
        self.log_basis = log_basis
        self.B = 1 << log_basis
        self.ell = self.bits // log_basis if reverse_length is None else reverse_length
        self.drop = self.bits - self.ell * log_basis
        # threshold (basis.rs:87-131)
        if log_basis == 1:
            if self.drop == 0:
                t = None
            else:
                t = 0
                for _ in range(self.ell):
                    t = (t << 1) | 1
                t = (t << 1) | 1
                t <<= self.drop - 1
        else:
            t = 0
            for _ in range(self.ell):
                t = (t << log_basis) | ((self.B - 1) >> 1)
            if self.drop > 0:
                t = ((t << 1) | 1) << (self.drop - 1)
            else:
                t += 1
        if t is not None and t >= self.Q:
            t = None
        self.threshold = t
        self.add = (1 << self.bits) - self.Q

    def unsigned_digits(self, v: int):
        if self.threshold is not None and v >= self.threshold:
            v += self.add
        carry = (v >> (self.drop - 1)) & 1 if self.drop > 0 else 0
        cm = 2 if self.log_basis == 1 else (self.B | (self.B >> 1))
        out = []
        for j in range(self.ell):
            t = ((v >> (self.drop + j * self.log_basis)) & (self.B - 1)) + carry
            carry = 1 if (t & cm) else 0
            out.append(t & (self.B - 1))
        return out

    def signed_digits(self, v: int):
        if self.B == 2:
            return self.unsigned_digits(v)
        half = (self.B + 1) // 2
        return [u if u < half else u - self.B for u in self.unsigned_digits(v)]

    def scalar(self, j: int) -> int:
        return 1 << (self.drop + j * self.log_basis)


def external_product_coeff(moduli, n, k, gadget: Gadget, crt_glwe, key_coeff):
    """sum_i sum_j digit_ij(X) (*) key[i][j][c](X) mod (X^n+1, q_r) in coefficient form.

    crt_glwe: (k+1, L, n) ints; key_coeff: (k+1, ell, k+1, L, n) ints (coefficient domain).
    Returns (k+1, L, n) ints.
    """
    L = len(moduli)
    out = [[[0] * n for _ in range(L)] for _ in range(k + 1)]
    for i in range(k + 1):
        vals = [crt_compose([crt_glwe[i][r][t] for r in range(L)], moduli) for t in range(n)]
        digs = [gadget.signed_digits(v) for v in vals]
        for j in range(gadget.ell):
            d = [digs[t][j] for t in range(n)]
            for c in range(k + 1):
                for r in range(L):
                    q = moduli[r]
                    prod = negacyclic_mul([x % q for x in d], key_coeff[i][j][c][r], q)
                    out[c][r] = [(x + y) % q for x, y in zip(out[c][r], prod)]
    return out


def ntt_fast(a, q: int, log_n: int, psi: int | None = None):
    """Same output as ntt_direct in O(N log N): recursive splitting of Z_q[X]/(X^m - psi^e) into
    (X^(m/2) - psi^(e/2)) and (X^(m/2) + psi^(e/2)), left factor first.  Python ints throughout."""
    n = 1 << log_n
    psi = psi or minimal_primitive_root(log_n + 1, q)
    two_n = 2 * n

    def rec(c, e):
        m = len(c)
        if m == 1:
            return c
        h = m // 2
        w = pow(psi, e // 2, q)
        t = [w * c[j + h] % q for j in range(h)]
        lo = [(c[j] + t[j]) % q for j in range(h)]
        hi = [(c[j] - t[j]) % q for j in range(h)]
        return rec(lo, e // 2) + rec(hi, (e // 2 + n) % two_n)

    return rec([int(x) % q for x in a], n)


def intt_fast(v, q: int, log_n: int, psi: int | None = None):
    """Inverse of ntt_fast (undoes each split: lo = (L+R)/2, hi = (L-R)/(2w))."""
    n = 1 << log_n
    psi = psi or minimal_primitive_root(log_n + 1, q)
    two_n = 2 * n
    half = pow(2, -1, q)

    def rec(c, e):
        m = len(c)
        if m == 1:
            return c
        h = m // 2
        L, R = rec(c[:h], e // 2), rec(c[h:], (e // 2 + n) % two_n)
        winv = pow(psi, -(e // 2), q) * half % q
        return [(x + y) * half % q for x, y in zip(L, R)] + [(x - y) * winv % q for x, y in zip(L, R)]

    return rec([int(x) for x in v], n)


def negacyclic_mul_fast(a, b, q: int, log_n: int):
    psi = minimal_primitive_root(log_n + 1, q)
    fa, fb = ntt_fast(a, q, log_n, psi), ntt_fast(b, q, log_n, psi)
    return intt_fast([x * y % q for x, y in zip(fa, fb)], q, log_n, psi)
