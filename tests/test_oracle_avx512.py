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
