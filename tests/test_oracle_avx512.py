"""The oracle's AVX-512 forward NTT (restating prime64/avx512/) against its scalar path: canonical
outputs identical, lazy outputs equal mod q and < 4q.  Skipped on hosts without AVX-512 DQ."""
import numpy as np
import pytest

from pyref import Q61, Q62


@pytest.mark.parametrize("log_n,q", [(4, Q61[0]), (5, 1125899906826241), (8, Q62), (10, Q61[1]), (11, Q61[2]),
                                     (12, 1152921504606830593), (14, Q61[0]), (16, Q61[2]), (17, Q62)])
def test_avx512_forward_equals_scalar(orc, log_n, q):
    if not orc.lib().orc_avx512_available():
        pytest.skip("host has no AVX-512 DQ")
    rng = np.random.default_rng(log_n)
    t = orc.U64NttTable(log_n, q)
    a = rng.integers(0, q, 3 << log_n, dtype=np.uint64)
    a[:4] = [0, q - 1, 1, q // 2]
    ref = a.copy(); t.transform_slice(ref)
    got = a.copy(); t.transform_slice_avx512(got)
    assert np.array_equal(got, ref)
    lz = rng.integers(0, 4 * q, 1 << log_n, dtype=np.uint64)
    can = (lz % np.uint64(q)).copy(); t.transform_slice(can)
    t.transform_slice_avx512(lz, lazy=True)
    assert lz.max() < 4 * q and np.array_equal(lz % np.uint64(q), can)


@pytest.mark.parametrize("log_n,q", [(4, Q61[0]), (5, 1125899906826241), (6, Q62), (8, Q62), (10, Q61[1]), (11, Q61[2]),
                                     (12, 1152921504606830593), (14, Q61[0]), (16, Q61[2]), (17, Q62)])
def test_avx512_inverse_equals_scalar(orc, log_n, q):
    """prime64/avx512/transform.rs:205 restated: canonical outputs identical to the scalar inverse, lazy outputs below
    2q and equal mod q; forward then inverse through the vector backend is the identity."""
    if not orc.lib().orc_avx512_available():
        pytest.skip("host has no AVX-512 DQ")
    rng = np.random.default_rng(100 + log_n)
    t = orc.U64NttTable(log_n, q)
    a = rng.integers(0, q, 3 << log_n, dtype=np.uint64)
    a[:4] = [0, q - 1, 1, q // 2]
    ref = a.copy(); t.inverse_transform_slice(ref)
    got = a.copy(); t.inverse_transform_slice_avx512(got)
    assert np.array_equal(got, ref)
    lz = a.copy(); t.inverse_transform_slice_avx512(lz, lazy=True)
    assert lz.max() < 2 * q and np.array_equal(lz % np.uint64(q), ref)
    rt = a.copy(); t.transform_slice_avx512(rt); t.inverse_transform_slice_avx512(rt)
    assert np.array_equal(rt, a)


def test_vector_backend_switch_covers_both_directions(orc):
    """orc_set_vector_backend routes transform_slice AND inverse_transform_slice (bench.py's cpu_baseline)."""
    if not orc.lib().orc_avx512_available():
        pytest.skip("host has no AVX-512 DQ")
    rng = np.random.default_rng(3)
    t = orc.U64NttTable(12, Q61[0])
    a = rng.integers(0, Q61[0], 1 << 12, dtype=np.uint64)
    f0, i0 = a.copy(), a.copy()
    t.transform_slice(f0); t.inverse_transform_slice(i0)
    orc.lib().orc_set_vector_backend(1)
    try:
        f1, i1 = a.copy(), a.copy()
        t.transform_slice(f1); t.inverse_transform_slice(i1)
    finally:
        orc.lib().orc_set_vector_backend(0)
    assert np.array_equal(f0, f1) and np.array_equal(i0, i1)


def test_avx512_rejects_tiny_transforms(orc):
    if not orc.lib().orc_avx512_available():
        pytest.skip("host has no AVX-512 DQ")
    t = orc.U64NttTable(3, 97)
    with pytest.raises(orc.OracleError):
        t.transform_slice_avx512(np.zeros(8, np.uint64))


Q50 = 1125899906826241  # the reference's own bench prime (benches/bench_u64.rs:8), below 2^50: the IFMA rung


@pytest.mark.parametrize("log_n,q", [(4, Q50), (5, 1125899906629633), (10, 562949953392641), (12, Q50), (13, 132120577),
                                     (16, 1125899903827969)])
def test_avx512_ifma_rung_equals_scalar(orc, log_n, q):
    """BIT_SHIFT = 52 (prime64/avx512 with vpmadd52, q < 2^50: internal.rs:12,24,28; table.rs:166-186,236-256): canonical
    outputs identical to the scalar path and to the DQ rung, lazy outputs in range and equal mod q, round trip."""
    if not orc.lib().orc_avx512_ifma_available():
        pytest.skip("host has no AVX-512 IFMA")
    rng = np.random.default_rng(200 + log_n)
    t = orc.U64NttTable(log_n, q)
    a = rng.integers(0, q, 3 << log_n, dtype=np.uint64)
    a[:4] = [0, q - 1, 1, q // 2]
    ref = a.copy(); t.transform_slice(ref)
    for shift in (52, 64, 0):
        got = a.copy(); t.transform_slice_avx512(got, shift=shift)
        assert np.array_equal(got, ref), shift
    lz = rng.integers(0, 4 * q, 1 << log_n, dtype=np.uint64)
    can = (lz % np.uint64(q)).copy(); t.transform_slice(can)
    t.transform_slice_avx512(lz, lazy=True, shift=52)
    assert lz.max() < 4 * q and np.array_equal(lz % np.uint64(q), can)
    iref = a.copy(); t.inverse_transform_slice(iref)
    for shift in (52, 64, 0):
        got = a.copy(); t.inverse_transform_slice_avx512(got, shift=shift)
        assert np.array_equal(got, iref), shift
    il = a.copy(); t.inverse_transform_slice_avx512(il, lazy=True, shift=52)
    assert il.max() < 2 * q and np.array_equal(il % np.uint64(q), iref)
    rt = a.copy(); t.transform_slice_avx512(rt, shift=52); t.inverse_transform_slice_avx512(rt, shift=52)
    assert np.array_equal(rt, a)


def test_avx512_ifma_rejects_wide_moduli(orc):
    """The IFMA rung is for q < 2^50 (MAX_FWD_IFMA_MODULUS); a 61-bit prime must not take it."""
    if not orc.lib().orc_avx512_ifma_available():
        pytest.skip("host has no AVX-512 IFMA")
    t = orc.U64NttTable(8, Q61[0])
    with pytest.raises(orc.OracleError):
        t.transform_slice_avx512(np.zeros(256, np.uint64), shift=52)
    with pytest.raises(orc.OracleError):
        t.inverse_transform_slice_avx512(np.zeros(256, np.uint64), shift=52)


@pytest.mark.parametrize("log_n,q", [(4, 132120577), (8, 536813569), (10, 1073479681), (13, 132120577), (16, 1071513601)])
def test_avx512_32bit_rung_equals_scalar(orc, log_n, q):
    """BIT_SHIFT = 32 (q < 2^30: prime64/avx512 butterfly.rs:23-29,90-96, transform.rs:375-387; table.rs:188-200): canonical
    outputs identical to the scalar path (which takes its own Barrett-32 butterflies for these primes) and to the other
    rungs, lazy outputs in range and equal mod q."""
    if not orc.lib().orc_avx512_available():
        pytest.skip("host has no AVX-512 DQ")
    rng = np.random.default_rng(300 + log_n)
    t = orc.U64NttTable(log_n, q)
    a = rng.integers(0, q, 3 << log_n, dtype=np.uint64)
    a[:4] = [0, q - 1, 1, q // 2]
    ref = a.copy(); t.transform_slice(ref)
    iref = a.copy(); t.inverse_transform_slice(iref)
    shifts = (32, 64, 0) + ((52,) if orc.lib().orc_avx512_ifma_available() else ())
    for shift in shifts:
        got = a.copy(); t.transform_slice_avx512(got, shift=shift)
        assert np.array_equal(got, ref), shift
        got = a.copy(); t.inverse_transform_slice_avx512(got, shift=shift)
        assert np.array_equal(got, iref), shift
    lz = rng.integers(0, 4 * q, 1 << log_n, dtype=np.uint64)
    can = (lz % np.uint64(q)).copy(); t.transform_slice(can)
    t.transform_slice_avx512(lz, lazy=True, shift=32)
    assert lz.max() < 4 * q and np.array_equal(lz % np.uint64(q), can)
    il = a.copy(); t.inverse_transform_slice_avx512(il, lazy=True, shift=32)
    assert il.max() < 2 * q and np.array_equal(il % np.uint64(q), iref)
    with pytest.raises(orc.OracleError):
        orc.U64NttTable(8, Q61[0]).transform_slice_avx512(np.zeros(256, np.uint64), shift=32)
