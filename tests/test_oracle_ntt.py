"""Oracle pin #2: NTT tables and transforms.

Re-creates the reference's own NTT tests against the C restatement
(primus_ntt/src/ntt/prime64/tests.rs, primus_ntt/tests/ntt.rs, tests/root.rs) and adds an
independent Python big-integer evaluation (tests/pyref.py).
"""
import numpy as np
import pytest

import pyref
from pyref import Q61, Q62, SURVEY_MIN_ROOTS


def rand_poly(rng, q, n, bound=1):
    return rng.integers(0, bound * q, n, dtype=np.uint64)


@pytest.mark.parametrize("key", sorted(SURVEY_MIN_ROOTS))
def test_minimal_root_matches_survey_table(orc, key):
    q, log_n = key
    assert orc.minimal_primitive_root(log_n + 1, q) == SURVEY_MIN_ROOTS[key]


@pytest.mark.parametrize("q,log_n", [(132120577, 3), (132120577, 10), (Q62, 10), (Q61[1], 8),
                                     (1125899906826241, 6)])
def test_minimal_root_matches_python(orc, q, log_n):
    assert orc.minimal_primitive_root(log_n + 1, q) == pyref.minimal_primitive_root(log_n + 1, q)


def test_table_errors(orc):
    # 2N does not divide q-1 -> NoPrimitiveRoot (root.rs:76-81)
    with pytest.raises(orc.OracleError) as e:
        orc.U64NttTable(20, 1125899906826241)
    assert e.value.code == 1
    # q >= 2^62 -> ModulusTooLarge (table.rs:318-323); 2^62 + 2^17*k + 1 style prime-free check:
    # the root search runs first, so use a modulus that has a root: 4611686018427387904+... skip
    # root existence by picking q = 2^62 + 134217729? not prime -> may fail root; accept either.
    with pytest.raises(orc.OracleError):
        orc.U64NttTable(4, (1 << 62) + 33)


@pytest.mark.parametrize("q,log_n", [(132120577, 3), (132120577, 5), (Q62, 4), (Q61[0], 6),
                                     (1125899906826241, 5), (536813569, 7)])
def test_forward_matches_direct_evaluation(orc, q, log_n):
    """Output index i holds a(psi^(2*brv(i)+1)) (table.rs:580-589)."""
    rng = np.random.default_rng(log_n)
    t = orc.U64NttTable(log_n, q)
    a = rand_poly(rng, q, t.n)
    expect = pyref.ntt_direct(a, q, log_n, t.root)
    got = a.copy()
    t.transform_slice(got)
    assert np.array_equal(got, expect)
    t.inverse_transform_slice(got)
    assert np.array_equal(got, a)


@pytest.mark.parametrize("q", [536813569, 562949953392641, 1152921504606830593, Q62] + Q61)
@pytest.mark.parametrize("log_n", [3, 4, 7, 10, 11])
def test_u64_table_equals_uint_table(orc, q, log_n):
    """prime64/tests.rs:100-237 and tests/ntt.rs:16-127: canonical outputs are identical,
    lazy outputs agree mod q and stay in their documented ranges."""
    rng = np.random.default_rng(q % 1000 + log_n)
    t, u = orc.U64NttTable(log_n, q), orc.UintNttTable(log_n, q)
    a = rand_poly(rng, q, t.n)
    x, y = a.copy(), a.copy()
    t.transform_slice(x); u.transform_slice(y)
    assert np.array_equal(x, y)
    xl, yl = a.copy(), a.copy()
    t.lazy_transform_slice(xl); u.lazy_transform_slice(yl)
    assert xl.max() < 4 * q and np.array_equal(xl % np.uint64(q), x) and np.array_equal(yl % np.uint64(q), x)
    xi, yi = x.copy(), x.copy()
    t.inverse_transform_slice(xi); u.inverse_transform_slice(yi)
    assert np.array_equal(xi, a) and np.array_equal(yi, a)
    xli = x.copy()
    t.lazy_inverse_transform_slice(xli)
    assert xli.max() < 2 * q and np.array_equal(xli % np.uint64(q), a)


@pytest.mark.parametrize("q", [132120577, 536813569])
def test_barrett32_equals_barrett64(orc, q):
    """prime64/tests.rs:243-272: scalar BIT_SHIFT 32 and 64 agree on canonical output."""
    rng = np.random.default_rng(3)
    t = orc.U64NttTable(10, q)
    a = rand_poly(rng, q, t.n)
    x, y = a.copy(), a.copy()
    t.scalar_forward(x, 32, 1); t.scalar_forward(y, 64, 1)
    assert np.array_equal(x, y)
    t.scalar_inverse(x, 32, 1); t.scalar_inverse(y, 64, 1)
    assert np.array_equal(x, a) and np.array_equal(y, a)


def test_lazy_forward_accepts_4q_inputs(orc):
    """ntt/mod.rs:52-60: lazy_transform_slice takes inputs in [0,4q)."""
    q = Q62
    rng = np.random.default_rng(5)
    t = orc.U64NttTable(8, q)
    a = rand_poly(rng, q, t.n, bound=4)
    x = a.copy(); t.lazy_transform_slice(x)
    ref = (a % np.uint64(q)).copy(); t.transform_slice(ref)
    assert x.max() < 4 * q and np.array_equal(x % np.uint64(q), ref)


@pytest.mark.parametrize("q,log_n", [(132120577, 10), (Q61[2], 6), (1125899906826241, 11)])
def test_monomial_transforms(orc, q, log_n):
    """prime64/tests.rs:160-173 + closed form table.rs:580-589."""
    t, u = orc.U64NttTable(log_n, q), orc.UintNttTable(log_n, q)
    n = t.n
    rng = np.random.default_rng(9)
    for degree in [0, 1, 2, n // 2, n - 1, n, n + 3, 2 * n - 1]:
        for coeff in [0, 1, q - 1, int(rng.integers(2, q - 1))]:
            got = t.transform_monomial(coeff, degree)
            assert np.array_equal(got, u.transform_monomial(coeff, degree))
            # NTT of coeff * X^degree (degree >= n wraps negacyclically)
            poly = np.zeros(n, np.uint64)
            if degree < n:
                poly[degree] = coeff
            else:
                poly[degree - n] = (q - coeff) % q
            t.transform_slice(poly)
            assert np.array_equal(got, poly)
        assert np.array_equal(t.transform_coeff_one_monomial(degree), t.transform_monomial(1, degree))
        assert np.array_equal(t.transform_coeff_minus_one_monomial(degree), t.transform_monomial(q - 1, degree))


@pytest.mark.parametrize("log_n", [3, 5, 6])
def test_ntt_product_equals_schoolbook(orc, log_n):
    q = Q61[0]
    rng = np.random.default_rng(log_n)
    t = orc.U64NttTable(log_n, q)
    a, b = rand_poly(rng, q, t.n), rand_poly(rng, q, t.n)
    expect = np.array(pyref.negacyclic_mul(a, b, q), np.uint64)
    assert np.array_equal(orc.naive_negacyclic_mul(q, a, b), expect)
    x, y = a.copy(), b.copy()
    t.transform_slice(x); t.transform_slice(y)
    import ctypes as C
    orc.lib().orc_reduce_mul_slice_assign(q, x.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          y.ctypes.data_as(C.POINTER(C.c_uint64)), t.n)
    t.inverse_transform_slice(x)
    assert np.array_equal(x, expect)


def test_table_closed_forms(orc):
    """roots[brv(i)] = psi^i, inv_roots[brv(i)+1] = psi^(2N-1-i),
    inv_n_w = inv_n * inv_roots[N-1], Shoup precon = floor(w 2^64 / q)."""
    q, log_n = Q61[1], 9
    t = orc.U64NttTable(log_n, q)
    n, psi = t.n, t.root
    roots, inv_roots, rp, irp = t.roots, t.inv_roots, t.roots_precon64, t.inv_roots_precon64
    assert psi * t.inv_root % q == 1 and t.inv_n * n % q == 1
    for i in range(n):
        assert int(roots[pyref.brv(i, log_n)]) == pow(psi, i, q)
        assert int(rp[i]) == (int(roots[i]) << 64) // q
        assert int(irp[i]) == (int(inv_roots[i]) << 64) // q
    assert int(inv_roots[0]) == 1
    for i in range(n - 1):
        assert int(inv_roots[pyref.brv(i, log_n) + 1]) == pow(psi, 2 * n - 1 - i, q)
    assert t.inv_n_w == t.inv_n * int(inv_roots[n - 1]) % q


def test_dcrt_table_is_per_limb(orc):
    """dcrt/prime64.rs:106-127: modulus-major chunks, one table per limb."""
    log_n = 8
    d = orc.U64DcrtTable(log_n, Q61)
    rng = np.random.default_rng(11)
    a = np.concatenate([rand_poly(rng, q, d.n) for q in Q61])
    x = a.copy(); d.transform_slice(x)
    for i, q in enumerate(Q61):
        ref = a[i * d.n:(i + 1) * d.n].copy()
        orc.U64NttTable(log_n, q).transform_slice(ref)
        assert np.array_equal(x[i * d.n:(i + 1) * d.n], ref)
    d.inverse_transform_slice(x)
    assert np.array_equal(x, a)
