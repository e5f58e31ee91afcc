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


def test_avx512_rejects_tiny_transforms(orc):
    if not orc.lib().orc_avx512_available():
        pytest.skip("host has no AVX-512 DQ")
    t = orc.U64NttTable(3, 97)
    with pytest.raises(orc.OracleError):
        t.transform_slice_avx512(np.zeros(8, np.uint64))
