"""The oracle at RNS bases of more than eight moduli (RNSBase::new takes any number, primus_rns/src/base.rs:79-117)
against Python big integers: the GPU tests of tests/test_gpu_wide_base.py lean on it at L = 9 ... 32."""
import numpy as np
import pytest

import pyref
from primes import ntt_primes_below
from pyref import crt_compose, int_to_limbs, limbs_to_int


def rand_rns(rng, moduli, n, batch=1):
    return np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for _ in range(batch) for q in moduli])


@pytest.mark.parametrize("L,bits", [(9, 61), (12, 45), (16, 61), (24, 30), (32, 61)])
def test_compose_decompose_and_converter(orc, L, bits):
    moduli = ntt_primes_below(L, bits, 4)
    rng = np.random.default_rng(L)
    ob = orc.RNSBase(moduli)
    n, vl = 40, ob.value_len
    Q = 1
    for q in moduli:
        Q *= q
    assert limbs_to_int(ob.moduli_product) == Q and vl == (Q.bit_length() + 63) // 64
    res = rand_rns(rng, moduli, n)
    out = ob.compose_multiple_values_to(res, n)
    for c in range(n):
        assert limbs_to_int(out[c * vl:(c + 1) * vl]) == crt_compose([int(res[i * n + c]) for i in range(L)], moduli)
    assert np.array_equal(ob.decompose_big_uint_values_to(out, n), res)
    # base conversion into a second wide base: sum_i t_i (Q/q_i) mod p_j, and the exact form x mod p_0
    mod_out = ntt_primes_below(L, bits - 1, 4)
    conv = orc.BaseConverter(ob, orc.RNSBase(mod_out))
    fast = conv.fast_convert_array(res, n)
    exact = orc.BaseConverter(ob, orc.RNSBase(mod_out[:1])).exact_convert_array(res, n)
    for c in range(n):
        t = [(int(res[i * n + c]) * pow(Q // q, -1, q)) % q for i, q in enumerate(moduli)]
        s = sum(ti * (Q // q) for ti, q in zip(t, moduli))
        assert [int(fast[j * n + c]) for j in range(L)] == [s % p for p in mod_out]
        x = s % Q
        if min(x, Q - x) > Q >> 40:  # away from the rounding boundary the f64 correction is exact: centred value mod p_0
            assert int(exact[c]) == (x if 2 * x < Q else x - Q) % mod_out[0]


@pytest.mark.parametrize("L,bits,log_basis,rev", [(9, 61, 30, None), (12, 45, 13, 7), (16, 61, 45, None), (24, 30, 7, 20)])
def test_gadget_digits(orc, L, bits, log_basis, rev):
    moduli = ntt_primes_below(L, bits, 4)
    rng = np.random.default_rng(L + log_basis)
    ob = orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(ob, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    assert (obasis.decompose_length, obasis.drop_bits) == (g.ell, g.drop)
    vl, n = ob.value_len, 64
    vals = [int.from_bytes(rng.bytes(8 * vl + 8), "little") % g.Q for _ in range(n)]
    vals[:4] = [0, g.Q - 1, g.Q // 2, g.threshold or 1]
    v = np.concatenate([int_to_limbs(x, vl) for x in vals])
    carries = obasis.init_value_carry_slice_inplace(v, n)
    digits = [obasis.unsigned_decompose_slice_to(j, v, carries, n) for j in range(g.ell)]
    for c, x in enumerate(vals):
        assert [int(d[c]) for d in digits] == g.unsigned_digits(x)
        rec = sum(sd * g.scalar(j) for j, sd in enumerate(g.signed_digits(x)))
        err = (rec - x) % g.Q
        assert min(err, g.Q - err) <= (1 << max(g.drop - 1, 0))  # big_uint.rs:154,195-200


def test_external_product_schoolbook(orc):
    log_n, k, L, log_basis, rev = 4, 1, 9, 30, 5
    moduli = ntt_primes_below(L, 61, log_n)
    rng = np.random.default_rng(7)
    n = 1 << log_n
    table, ob = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(ob, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    ell = g.ell
    glwe = rand_rns(rng, moduli, n, k + 1)
    key_coeff = rand_rns(rng, moduli, n, (k + 1) * ell * (k + 1))
    ggsw = key_coeff.copy()
    table.transform_slice(ggsw)
    out = orc.mul_dcrt_ggsw_to(table, ob, obasis, k, glwe, ggsw)
    table.inverse_transform_slice(out)
    exp = pyref.external_product_coeff(moduli, n, k, g, glwe.reshape(k + 1, L, n).tolist(),
                                       key_coeff.reshape(k + 1, ell, k + 1, L, n).tolist())
    assert out.reshape(k + 1, L, n).tolist() == exp
