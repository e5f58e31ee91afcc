"""Pin the oracle's U32NttTable restatement (prime32/table.rs, prime32/scalar/*.rs).

Re-creates primus_ntt/src/ntt/prime32/tests.rs (lazy/canonical ranges, lazy == canonical mod q,
round trips over N = 8..1024, cross-check against a second table implementation, monomials) and
adds the independent big-integer evaluation of tests/pyref.py.
"""
import numpy as np
import pytest

import pyref

Q27 = 132120577          # prime32/tests.rs:5 -- 27-bit, 1 mod 2048
Q30 = [1073479681, 1071513601, 1070727169]   # largest primes below 2^30 that are 1 mod 2^17
Q29 = 536813569          # prime64/tests.rs's 30-bit-path prime


def rand32(rng, hi, n):
    return rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)


def test_q30_primes_are_what_they_claim():
    for q in Q30:
        assert q < 1 << 30 and (q - 1) % (1 << 17) == 0
        assert all(q % p for p in range(3, 40000, 2))


def test_mul_mod_lazy_is_barrett32(orc):
    rng = np.random.default_rng(1)
    for q in [Q27, Q29] + Q30:
        for _ in range(200):
            w = int(rng.integers(0, q))
            y = int(rng.integers(0, 1 << 32))
            wp = (w << 32) // q
            r = orc.lib().orc_u32_mul_mod_lazy(y, w, wp, q)
            assert r < 2 * q and r % q == w * y % q


def test_table_constants(orc):
    for log_n, q in [(10, Q27), (3, Q27), (16, Q30[0]), (1, 17), (0, 17)]:
        t = orc.U32NttTable(log_n, q)
        n = 1 << log_n
        psi = pyref.minimal_primitive_root(log_n + 1, q)
        assert t.root == psi and t.inv_root == pow(psi, -1, q)
        assert t.inv_n == pow(n, -1, q)
        roots, inv_roots = t.roots, t.inv_roots
        for k in range(min(n, 64)):
            assert roots[pyref.brv(k, log_n)] == pow(psi, k, q)
        for k in range(min(n - 1, 64)):
            assert inv_roots[pyref.brv(k, log_n) + 1] == pow(psi, 2 * n - 1 - k, q)
        assert t.inv_n_w == t.inv_n * int(inv_roots[n - 1]) % q
        # the u64 table built for the same (N, q) holds the same roots (same minimal psi)
        t64 = orc.U64NttTable(log_n, q)
        assert np.array_equal(t64.roots.astype(np.uint32), roots)


def test_errors(orc):
    q_big = next(q for q in range((1 << 30) + 1, (1 << 30) + (1 << 22), 2048)
                 if all(q % p for p in range(3, 33000, 2)))
    with pytest.raises(orc.OracleError) as e:
        orc.U32NttTable(10, q_big)
    assert e.value.code == 5  # ModulusTooLarge { max_bits: 30 } (table.rs:195-200)
    with pytest.raises(orc.OracleError) as e:
        orc.U32NttTable(21, Q27)  # q - 1 = 63 * 2^21: no root of order 2^22
    assert e.value.code == 1


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("q", [Q27, Q30[1]])
def test_forward_matches_direct_evaluation(orc, log_n, q):
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    t = orc.U32NttTable(log_n, q)
    a = rand32(rng, q, n)
    exp = pyref.ntt_direct(a.tolist(), q, log_n) if log_n else np.array(a)
    x = a.copy()
    t.transform_slice(x)
    assert x.tolist() == [int(v) for v in exp]
    t.inverse_transform_slice(x)
    assert np.array_equal(x, a)


@pytest.mark.parametrize("log_n,q", [(10, Q27), (10, Q29), (12, Q30[0]), (16, Q30[2])])
def test_cross_check_against_u64_table(orc, log_n, q):
    """prime32/tests.rs:133-236 with U64NttTable as the second implementation."""
    rng = np.random.default_rng(log_n + q % 97)
    n = 1 << log_n
    t32, t64 = orc.U32NttTable(log_n, q), orc.U64NttTable(log_n, q)
    a = rand32(rng, q, n)
    x, y = a.copy(), a.astype(np.uint64)
    t32.transform_slice(x); t64.transform_slice(y)
    assert np.array_equal(x.astype(np.uint64), y)
    if log_n >= 12:
        assert x.tolist() == pyref.ntt_fast(a.tolist(), q, log_n)
    lz = a.copy(); t32.lazy_transform_slice(lz)
    assert lz.max() < 4 * q and np.array_equal(lz % np.uint32(q), x)
    x2, y2 = a.copy(), a.astype(np.uint64)
    t32.inverse_transform_slice(x2); t64.inverse_transform_slice(y2)
    assert np.array_equal(x2.astype(np.uint64), y2)
    lzi = a.copy(); t32.lazy_inverse_transform_slice(lzi)
    assert lzi.max() < 2 * q and np.array_equal(lzi % np.uint32(q), x2)
    coeff, degree = int(rng.integers(1, q)), int(rng.integers(1, n))
    assert np.array_equal(t32.transform_monomial(coeff, degree).astype(np.uint64), t64.transform_monomial(coeff, degree))


def test_lazy_ranges_and_lazy_vs_canonical(orc):
    """prime32/tests.rs:13-112: inputs up to 4q (forward) / 2q (inverse)."""
    rng = np.random.default_rng(7)
    t = orc.U32NttTable(10, Q27)
    a = rand32(rng, 4 * Q27, 1024)
    lz = a.copy(); t.lazy_transform_slice(lz)
    assert lz.max() < 4 * Q27
    can = (a % np.uint32(Q27)).copy(); t.transform_slice(can)
    assert can.max() < Q27 and np.array_equal(lz % np.uint32(Q27), can)
    b = rand32(rng, 2 * Q27, 1024)
    lzi = b.copy(); t.lazy_inverse_transform_slice(lzi)
    assert lzi.max() < 2 * Q27
    cani = (b % np.uint32(Q27)).copy(); t.inverse_transform_slice(cani)
    assert cani.max() < Q27 and np.array_equal(lzi % np.uint32(Q27), cani)


def test_round_trip_all_sizes(orc):
    """prime32/tests.rs:114-137."""
    rng = np.random.default_rng(9)
    for log_n in range(3, 11):
        t = orc.U32NttTable(log_n, Q27)
        a = rand32(rng, Q27, 1 << log_n)
        x = a.copy(); t.transform_slice(x); t.inverse_transform_slice(x)
        assert np.array_equal(x, a)


@pytest.mark.parametrize("log_n", [1, 4, 10])
def test_monomials(orc, log_n):
    """transform_monomial == transform of the explicit monomial (table.rs:376-470)."""
    q, n = Q27, 1 << log_n
    t = orc.U32NttTable(log_n, q)
    for coeff, degree in [(0, 3), (5, 0), (1, 1), (q - 1, n - 1), (12345, n // 2), (777, 1)]:
        degree %= n
        m = np.zeros(n, np.uint32); m[degree] = coeff
        t.transform_slice(m)
        assert np.array_equal(t.transform_monomial(coeff, degree), m)
    for degree in (0, 1, n - 1):
        one = np.zeros(n, np.uint32); one[degree] = 1; t.transform_slice(one)
        assert np.array_equal(t.transform_coeff_one_monomial(degree), one)
        neg = np.zeros(n, np.uint32); neg[degree] = q - 1; t.transform_slice(neg)
        assert np.array_equal(t.transform_coeff_minus_one_monomial(degree), neg)


def test_dcrt32_polymul_equals_schoolbook(orc):
    log_n, n = 6, 64
    rng = np.random.default_rng(3)
    t = orc.U32DcrtTable(log_n, Q30)
    a = np.concatenate([rand32(rng, q, n) for q in Q30])
    b = np.concatenate([rand32(rng, q, n) for q in Q30])
    fa, fb = a.copy(), b.copy()
    t.transform_slice(fa); t.transform_slice(fb)
    t.mul_assign(fa, fb)
    t.inverse_transform_slice(fa)
    for r, q in enumerate(Q30):
        s = slice(r * n, (r + 1) * n)
        assert fa[s].tolist() == pyref.negacyclic_mul(a[s].tolist(), b[s].tolist(), q)
