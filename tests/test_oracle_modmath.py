"""Oracle pin #1: scalar modular arithmetic against Python integers.

Mirrors primus_modulus/tests/barrett_modulus.rs:104-131 (mul vs `%`),
primus_factor/tests/shoup_factor.rs:21-39 and primus_factor/src/mul_factor/mod.rs:95-147
(Shoup / MultiplyFactor vs u128 %), primus_modulus/tests/barrett_modulus.rs:26-56 (add/sub).
"""
import random

import numpy as np
import pytest

from pyref import Q61, Q62, REF_TEST_PRIMES

MODULI = REF_TEST_PRIMES + Q61 + [Q62, 3, 97, (1 << 62) - 57]
M64 = (1 << 64) - 1


@pytest.mark.parametrize("q", MODULI)
def test_barrett_matches_python(orc, q):
    rnd = random.Random(q)
    m = orc.Barrett(q)
    assert (m.ratio[1] << 64 | m.ratio[0]) == (1 << 128) // q
    edge = [0, 1, q - 1, q // 2]
    for a in edge + [rnd.randrange(q) for _ in range(200)]:
        for b in edge + [rnd.randrange(q) for _ in range(5)]:
            c = rnd.randrange(q)
            assert m.reduce_mul(a, b) == a * b % q
            assert m.reduce_mul_add(a, b, c) == (a * b + c) % q
            lazy = m.lazy_reduce_wide((a * b) & M64, (a * b) >> 64)
            assert lazy < 2 * q and lazy % q == a * b % q
            assert orc.lib().orc_reduce_add(q, a, b) == (a + b) % q
            assert orc.lib().orc_reduce_sub(q, a, b) == (a - b) % q
    for v in [0, 1, q, 2 * q - 1, M64, rnd.getrandbits(64)]:
        assert m.reduce(v) == v % q


def test_barrett_rejects_bad_moduli(orc):
    for bad in (0, 1):
        with pytest.raises(orc.OracleError):
            orc.Barrett(bad)
    with pytest.raises(orc.OracleError):
        orc.Barrett(1 << 62)  # needs >= 2 leading zero bits (barrett/mod.rs:41-42)


@pytest.mark.parametrize("q", MODULI)
def test_shoup_matches_python(orc, q):
    rnd = random.Random(q + 1)
    L = orc.lib()
    for _ in range(300):
        w, b = rnd.randrange(q), rnd.randrange(q)
        wp = L.orc_shoup_quotient(w, q)
        assert wp == (w << 64) // q
        lazy = L.orc_mul_mod_lazy(b, w, wp, q)
        assert lazy < 2 * q and lazy % q == w * b % q
        assert L.orc_shoup_mul(w, wp, b, q) == w * b % q
        # the lazy multiply tolerates any 64-bit y because q < 2^62 (arithmetic.rs:32-35)
        y = rnd.getrandbits(64)
        lazy = L.orc_mul_mod_lazy(y, w, wp, q)
        assert lazy < 2 * q and lazy % q == w * y % q


@pytest.mark.parametrize("q", [132120577, 536813569, 268369921])
def test_barrett32_lazy_multiply(orc, q):
    """q < 2^30 path (arithmetic.rs:23-28)."""
    rnd = random.Random(7)
    L = orc.lib()
    for _ in range(300):
        w = rnd.randrange(q)
        y = rnd.randrange(4 * q)
        wp32 = (w << 32) // q
        lazy = L.orc_mul_mod_lazy32(y, w, wp32, q)
        assert lazy < 2 * q and lazy % q == w * y % q


def test_slice_kernels(orc):
    q = Q61[0]
    rng = np.random.default_rng(1)
    for n in list(range(0, 66)):  # lengths 0..65 as primus_modulus/tests/barrett_modulus.rs:58
        a = rng.integers(0, q, n, dtype=np.uint64)
        b = rng.integers(0, q, n, dtype=np.uint64)
        acc = rng.integers(0, q, n, dtype=np.uint64)
        exp_mul = np.array([int(x) * int(y) % q for x, y in zip(a, b)], np.uint64)
        exp_fma = np.array([(int(x) * int(y) + int(z)) % q for x, y, z in zip(a, b, acc)], np.uint64)
        a2 = a.copy()
        import ctypes as C
        p = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint64))
        orc.lib().orc_reduce_mul_slice_assign(q, p(a2), p(b), n)
        assert np.array_equal(a2, exp_mul)
        acc2 = acc.copy()
        orc.lib().orc_reduce_add_mul_slice_assign(q, p(acc2), p(a), p(b), n)
        assert np.array_equal(acc2, exp_fma)
