"""RNSBase::add_wrapping_decompose_small_values_scaled / add_decompose_small_values_scaled (primus_rns/src/base.rs:
326-443): the reference's own test case (tests/rns.rs:200-275) re-created, the formulas on Python integers, and the
HIP kernels against the restatement (-m gpu)."""
import numpy as np
import pytest

from pyref import Q61


def shoup(values, moduli):
    return [w for v, q in zip(values, moduli) for w in (v, (v << 64) // q)]


def lift(v, m, q):
    return v if (m == 2 or v < (m + 1) // 2) else q - m + v


def expected(moduli, small, acc, m, fvals, centred):
    n = len(small)
    out = [int(x) for x in acc]
    for i, q in enumerate(moduli):
        for c, v in enumerate(small):
            x = lift(int(v), m, q) if centred else int(v)
            out[i * n + c] = (out[i * n + c] + fvals[i] * x) % q
    return out


def reference_case():
    """tests/rns.rs:200-233: base (97, 101, 103), small modulus 7, values (3i+1) mod 7, factors (3, 5, 7),
    acc[i][c] = (11 + 7c + i) mod q_i."""
    moduli, m = [97, 101, 103], 7
    small = np.array([(i * 3 + 1) % m for i in range(17)], np.uint64)
    fvals = [3, 5, 7]
    acc = np.array([(11 + c * 7 + i) % q for i, q in enumerate(moduli) for c in range(17)], np.uint64)
    return moduli, m, small, fvals, acc


def test_oracle_reference_case(orc):
    moduli, m, small, fvals, acc = reference_case()
    base = orc.RNSBase(moduli)
    a = acc.copy()
    base.add_wrapping_decompose_small_values_scaled(small, a, m, shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, True)
    a = acc.copy()
    base.add_decompose_small_values_scaled(small, a, shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, False)


@pytest.mark.parametrize("moduli,m", [(Q61, 1 << 30), (Q61[:2], 2), ([97, 101, 103], 96), ([1125899906826241], 3),
                                      (Q61, (1 << 31) - 1)])
def test_oracle_matches_integers(orc, moduli, m):
    rng = np.random.default_rng(m % 1000)
    n = 257
    small = rng.integers(0, m, n, dtype=np.uint64)
    small[:3] = [0, m - 1, (m + 1) // 2 - 1 if m > 2 else 1]
    fvals = [int(rng.integers(0, q)) for q in moduli]
    acc = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli])
    base = orc.RNSBase(moduli)
    a = acc.copy(); base.add_wrapping_decompose_small_values_scaled(small, a, m, shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, True)
    a = acc.copy(); base.add_decompose_small_values_scaled(small, a, shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, False)
    # consistency with the unfused pair: wrapping_decompose_small_values_to then a factor multiply-add
    lifted = base.wrapping_decompose_small_values_to(small, m)
    a = acc.copy(); orc.CrtPolyOps(moduli, n).add_mul_factor_assign(a, lifted, shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, True)


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


@pytest.mark.gpu
def test_gpu_reference_case(pf):
    moduli, m, small, fvals, acc = reference_case()
    base = pf.RNSBase(moduli)
    a = acc.copy()
    base.add_wrapping_decompose_small_values_scaled(small, a, len(small), m, shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, True)
    a = acc.copy()
    base.add_decompose_small_values_scaled(small, a, len(small), shoup(fvals, moduli))
    assert a.tolist() == expected(moduli, small, acc, m, fvals, False)


@pytest.mark.gpu
@pytest.mark.parametrize("moduli,m,n", [(Q61, 1 << 30, 1 << 16), (Q61[:2], 2, 1000), ([97, 101, 103], 96, 1),
                                        ([1125899906826241], 3, 4097), (Q61, (1 << 31) - 1, 3 << 18)])
def test_gpu_matches_oracle(pf, orc, moduli, m, n):
    from gpu_util import to_dev, to_host
    rng = np.random.default_rng(n)
    small = rng.integers(0, m, n, dtype=np.uint64)
    fvals = [int(rng.integers(0, q)) for q in moduli]
    f = shoup(fvals, moduli)
    acc = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli])
    obase, base = orc.RNSBase(moduli), pf.RNSBase(moduli)
    e1 = acc.copy(); obase.add_wrapping_decompose_small_values_scaled(small, e1, m, f)
    e2 = acc.copy(); obase.add_decompose_small_values_scaled(small, e2, f)
    ds = to_dev(small)
    d1 = to_dev(acc); base.add_wrapping_decompose_small_values_scaled_dev(ds, d1, n, m, f)
    d2 = to_dev(acc); base.add_decompose_small_values_scaled_dev(ds, d2, n, f)
    assert np.array_equal(to_host(d1), e1) and np.array_equal(to_host(d2), e2)
    h = acc.copy(); base.add_wrapping_decompose_small_values_scaled(small, h, n, m, f)
    assert np.array_equal(h, e1)


@pytest.mark.gpu
def test_gpu_errors(pf):
    base = pf.RNSBase([97, 101, 103])
    small, acc = np.zeros(4, np.uint64), np.zeros(12, np.uint64)
    f = shoup([1, 2, 3], [97, 101, 103])
    with pytest.raises(pf.PfheError) as e:
        base.add_wrapping_decompose_small_values_scaled(small, acc[:8], 4, 7, f)
    assert e.value.kind == "BadLength"
    with pytest.raises(pf.PfheError) as e:  # base.rs:337-341: the small modulus must be below every RNS modulus
        base.add_wrapping_decompose_small_values_scaled(small, acc, 4, 97, f)
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:
        base.add_decompose_small_values_scaled(small, acc, 4, [97, 0] + f[2:])  # factor value not reduced
    assert e.value.kind == "BadArgument"
    base.add_decompose_small_values_scaled(small[:0], acc[:0], 0, f)  # empty: no-op
