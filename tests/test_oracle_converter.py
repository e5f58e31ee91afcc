"""Pin the oracle's BaseConverter restatement (primus_rns/src/converter.rs) and
RNSBase::decompose_big_uint_values_to (base.rs:457-481).

Re-creates primus_rns/tests/rns.rs:281-343 and adds big-integer ground truth (tests/pyref.py).
"""
import numpy as np
import pytest

import pyref
from pyref import Q61, crt_compose

Q60S = [1152921504606584833, 1152921504598720513, 1152921504597016577]  # 60-bit pairwise-coprime moduli


def pack_modulus_major(rows, L):
    return np.array([r[i] for i in range(L) for r in rows], np.uint64)


def test_reference_case_fast_and_exact(orc):
    """rns.rs:281-343."""
    inp, outp = orc.RNSBase([17, 19, 23]), orc.RNSBase([29, 31])
    conv = orc.BaseConverter(inp, outp)
    rows = [[0, 0, 0], [1, 2, 3], [16, 18, 22], [7, 11, 13], [4, 0, 19]]
    crt_in = pack_modulus_major(rows, 3)
    exp = np.empty(2 * len(rows), np.uint64)
    for c, r in enumerate(rows):
        o = conv.fast_convert(r)
        exp[c], exp[len(rows) + c] = o[0], o[1]
    assert np.array_equal(conv.fast_convert_array(crt_in, len(rows)), exp)
    # scalar fast_convert against the definition: sum_i [x_i * (Q/q_i)^-1]_{q_i} * (Q/q_i) mod p_j
    Q = 17 * 19 * 23
    for c, r in enumerate(rows):
        s = sum((x * pow(Q // q, -1, q) % q) * (Q // q) for x, q in zip(r, [17, 19, 23]))
        assert [int(exp[c]), int(exp[len(rows) + c])] == [s % 29, s % 31]
    exact = orc.BaseConverter(inp, orc.RNSBase([37]))
    vals = [0, 1, 2, 7, 16]
    got = exact.exact_convert_array(pack_modulus_major([[v] * 3 for v in vals], 3), len(vals))
    assert got.tolist() == [v % 37 for v in vals]
    with pytest.raises(orc.OracleError):
        conv.exact_convert_array(crt_in, len(rows))  # output base must hold exactly one modulus


@pytest.mark.parametrize("mod_in,mod_out", [(Q61, Q60S[:2]), (Q61[:2], Q60S), ([97, 101, 103, 107], [109, 113]),
                                            (Q61, [Q60S[0]]), ([1125899906826241, 1125899906629633], Q61)])
def test_fast_convert_definition(orc, mod_in, mod_out):
    rng = np.random.default_rng(len(mod_in) * 10 + len(mod_out))
    n = 50
    inp, outp = orc.RNSBase(mod_in), orc.RNSBase(mod_out)
    conv = orc.BaseConverter(inp, outp)
    Q = 1
    for q in mod_in:
        Q *= q
    M = conv.base_change_matrix.reshape(len(mod_out), len(mod_in))
    for j, p in enumerate(mod_out):
        for i, q in enumerate(mod_in):
            assert int(M[j, i]) == (Q // q) % p
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in mod_in])
    out = conv.fast_convert_array(x, n)
    for t in range(n):
        s = sum((int(x[i * n + t]) * pow(Q // q, -1, q) % q) * (Q // q) for i, q in enumerate(mod_in))
        assert [int(out[j * n + t]) for j in range(len(mod_out))] == [s % p for p in mod_out]


@pytest.mark.parametrize("mod_in,p", [(Q61, Q60S[0]), ([17, 19, 23], 37), (Q61[:2], 1125899906826241)])
def test_exact_convert_is_the_centred_value(orc, mod_in, p):
    """exact_convert_array returns [x]_Q's representative nearest to zero, reduced mod p, whenever
    x/Q is not within float error of 1/2 (the (sum + 0.5) rounding of converter.rs:331-338)."""
    rng = np.random.default_rng(p % 1000)
    n = 64
    inp = orc.RNSBase(mod_in)
    conv = orc.BaseConverter(inp, orc.RNSBase([p]))
    Q = 1
    for q in mod_in:
        Q *= q
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in mod_in])
    x[0::n] = 0  # coefficient 0 is the value 0
    out = conv.exact_convert_array(x, n)
    checked = 0
    for t in range(n):
        v = crt_compose([int(x[i * n + t]) for i in range(len(mod_in))], mod_in)
        frac = v / Q
        if abs(frac - 0.5) < 1e-9:
            continue
        exp = v % p if frac < 0.5 else (v - Q) % p
        assert int(out[t]) == exp
        checked += 1
    assert checked >= n - 2


def test_decompose_big_uint_values(orc):
    base = orc.RNSBase(Q61)
    rng = np.random.default_rng(5)
    Q = Q61[0] * Q61[1] * Q61[2]
    vals = [0, 1, Q - 1, Q // 2] + [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 59)) % Q
                                    for _ in range(20)]
    W = base.value_len
    big = np.concatenate([pyref.int_to_limbs(v, W) for v in vals])
    res = base.decompose_big_uint_values_to(big, len(vals))
    for i, q in enumerate(Q61):
        assert [int(r) for r in res[i * len(vals):(i + 1) * len(vals)]] == [v % q for v in vals]
    # compose o decompose = identity
    assert np.array_equal(base.compose_multiple_values_to(res, len(vals)), big)
