"""GPU parity of BaseConverter (fast / exact) and RNSBase::decompose_big_uint_values_to against the
oracle, through the C ABI.  Mirrors primus_rns/tests/rns.rs:281-343."""
import numpy as np
import pytest

import pyref
from gpu_util import to_dev, to_host
from pyref import Q61
from test_oracle_converter import Q60S, pack_modulus_major

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def test_reference_case(pf, orc):
    inp, outp = pf.RNSBase([17, 19, 23]), pf.RNSBase([29, 31])
    conv = pf.BaseConverter(inp, outp)
    oconv = orc.BaseConverter(orc.RNSBase([17, 19, 23]), orc.RNSBase([29, 31]))
    assert (conv.input_moduli_count(), conv.output_moduli_count()) == (3, 2)
    assert np.array_equal(conv.base_change_matrix(), oconv.base_change_matrix)
    rows = [[0, 0, 0], [1, 2, 3], [16, 18, 22], [7, 11, 13], [4, 0, 19]]
    crt_in = pack_modulus_major(rows, 3)
    out = np.full(2 * len(rows), 2 ** 64 - 1, np.uint64)
    conv.fast_convert_array(crt_in, out, len(rows))
    assert np.array_equal(out, oconv.fast_convert_array(crt_in, len(rows)))
    exact = pf.BaseConverter(inp, pf.RNSBase([37]))
    vals = [0, 1, 2, 7, 16]
    eo = np.empty(len(vals), np.uint64)
    exact.exact_convert_array(pack_modulus_major([[v] * 3 for v in vals], 3), eo, len(vals))
    assert eo.tolist() == [v % 37 for v in vals]
    with pytest.raises(pf.PfheError) as e:
        conv.exact_convert_array(crt_in, out, len(rows))
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:
        conv.fast_convert_array(crt_in, out[:-1].copy(), len(rows))
    assert e.value.kind == "BadLength"


@pytest.mark.parametrize("mod_in,mod_out", [(Q61, Q60S[:2]), (Q61[:2], Q60S), ([97, 101, 103, 107], [109, 113]),
                                            (Q61, [Q60S[0]]), ([1125899906826241, 1125899906629633], Q61),
                                            (Q61 + Q60S, [1125899906826241, 1125899906629633])])
@pytest.mark.parametrize("n", [1, 5, 4096, 65536 + 3])
def test_fast_and_exact_match_oracle(pf, orc, mod_in, mod_out, n):
    rng = np.random.default_rng(n + len(mod_in))
    conv = pf.BaseConverter(pf.RNSBase(mod_in), pf.RNSBase(mod_out))
    oin, oout = orc.RNSBase(mod_in), orc.RNSBase(mod_out)
    oconv = orc.BaseConverter(oin, oout)
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in mod_in])
    x[0] = 0
    out = np.empty(len(mod_out) * n, np.uint64)
    conv.fast_convert_array(x, out, n)
    assert np.array_equal(out, oconv.fast_convert_array(x, n))
    dout = to_dev(np.zeros_like(out))
    conv.fast_convert_array_dev(to_dev(x), dout, n)
    assert np.array_equal(to_host(dout), out)
    # exact conversion to the first output modulus: identical f64 correction term
    e = pf.BaseConverter(pf.RNSBase(mod_in), pf.RNSBase(mod_out[:1]))
    oe = orc.BaseConverter(oin, orc.RNSBase(mod_out[:1]))
    eo = np.empty(n, np.uint64)
    e.exact_convert_array(x, eo, n)
    assert np.array_equal(eo, oe.exact_convert_array(x, n))


def test_exact_convert_rounding_boundary(pf, orc):
    """Values around Q/2, where (sum + 0.5) decides between x and x - Q: GPU and oracle must take
    the same branch for every input (same IEEE operations in the same order)."""
    mod_in, p = Q61, Q60S[0]
    Q = Q61[0] * Q61[1] * Q61[2]
    rng = np.random.default_rng(99)
    vals = [Q // 2 + d for d in range(-40, 41)] + [Q // 2 + int(rng.integers(-2 ** 40, 2 ** 40)) for _ in range(200)]
    n = len(vals)
    x = np.array([v % q for q in mod_in for v in vals], np.uint64)
    e = pf.BaseConverter(pf.RNSBase(mod_in), pf.RNSBase([p]))
    oe = orc.BaseConverter(orc.RNSBase(mod_in), orc.RNSBase([p]))
    eo = np.empty(n, np.uint64)
    e.exact_convert_array(x, eo, n)
    ref = oe.exact_convert_array(x, n)
    assert np.array_equal(eo, ref)
    assert {int(r) for r in ref} <= {v % p for v in vals} | {(v - Q) % p for v in vals}


@pytest.mark.parametrize("moduli", [Q61, Q61[:1], [97, 101, 103], Q61 + Q60S[:2]])
def test_decompose_big_uint_values(pf, orc, moduli):
    rng = np.random.default_rng(len(moduli))
    base, obase = pf.RNSBase(moduli), orc.RNSBase(moduli)
    Q = 1
    for q in moduli:
        Q *= q
    W = base.big_uint_value_len()
    count = 1000
    vals = [0, 1, Q - 1, Q // 2] + [int.from_bytes(rng.bytes(8 * W), "little") % Q for _ in range(count - 4)]
    big = np.concatenate([pyref.int_to_limbs(v, W) for v in vals])
    res = np.empty(len(moduli) * count, np.uint64)
    base.decompose_big_uint_values_to(big, res, count)
    assert np.array_equal(res, obase.decompose_big_uint_values_to(big, count))
    dres = to_dev(np.zeros_like(res))
    base.decompose_big_uint_values_to_dev(to_dev(big), dres, count)
    assert np.array_equal(to_host(dres), res)
    # compose(decompose(v)) == v
    back = np.empty_like(big)
    base.compose_multiple_values_to(res, back, count)
    assert np.array_equal(back, big)


def test_fast_convert_to_pairs(pf, orc):
    """fast_convert_array_to_pair_iter (converter.rs:233-272, rns.rs:325-332): (mod p_0, mod p_1) pairs equal the
    two output rows of fast_convert_array."""
    import torch
    rng = np.random.default_rng(17)
    n = 5000
    conv = pf.BaseConverter(pf.RNSBase(Q61), pf.RNSBase(Q60S[:2]))
    oconv = orc.BaseConverter(orc.RNSBase(Q61), orc.RNSBase(Q60S[:2]))
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in Q61])
    exp = oconv.fast_convert_array(x, n)
    pairs = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    conv.fast_convert_array_to_pairs_dev(to_dev(x), pairs, n)
    got = to_host(pairs)
    assert np.array_equal(got[0::2], exp[:n]) and np.array_equal(got[1::2], exp[n:])
    three = pf.BaseConverter(pf.RNSBase(Q61), pf.RNSBase(Q60S))
    with pytest.raises(pf.PfheError) as e:
        three.fast_convert_array_to_pairs_dev(to_dev(x), pairs, n)
    assert e.value.kind == "BadArgument"
