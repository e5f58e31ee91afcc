"""The <u32> instantiations in the oracle (oracle/pfhe_oracle_rns32.c: RNSBase<u32>, BigUintApproxSignedBasis<u32>, external
product over U32DcrtTable) against Python big integers, against the 64-bit restatement on the same integers, and on the
reference's own u32 test case (primus_decompose/tests/big_uint.rs:17-135: moduli 134215681, 134176769, log B = 7)."""
import numpy as np
import pytest

import pyref
from primes import ntt_primes_below
from pyref import crt_compose

Q30 = [1073479681, 1071513601, 1070727169]
REF_U32 = [134215681, 134176769]  # big_uint.rs:21


def limbs32_to_int(a) -> int:
    return sum(int(x) << (32 * i) for i, x in enumerate(a))


def int_to_limbs32(v: int, n: int):
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)], np.uint32)


def rand32(rng, moduli, n, batch=1):
    return np.concatenate([rng.integers(0, q, n, dtype=np.uint64).astype(np.uint32) for _ in range(batch) for q in moduli])


def shoup32(v, q):
    return (v, (v << 32) // q)


BASES = [REF_U32, Q30, Q30[:1], [97, 101, 103], ntt_primes_below(5, 30, 4), ntt_primes_below(9, 28, 4),
         ntt_primes_below(16, 30, 4), [3, 5, 7]]


@pytest.mark.parametrize("moduli", BASES, ids=lambda m: f"L{len(m)}_{m[0]}")
def test_rns32_against_integers_and_the_64_bit_restatement(orc, moduli):
    rng = np.random.default_rng(len(moduli))
    b32, b64 = orc.RNSBase32(moduli), orc.RNSBase(moduli)
    Q = 1
    for q in moduli:
        Q *= q
    vl = b32.value_len
    assert vl == (Q.bit_length() + 31) // 32 and limbs32_to_int(b32.moduli_product) == Q   # big_integer.rs:675-686
    for i, q in enumerate(moduli):
        assert limbs32_to_int(b32.punctured_product[i * vl:(i + 1) * vl]) == Q // q
    n = 64
    res = rand32(rng, moduli, n)
    res[0] = 0
    for i, q in enumerate(moduli):
        res[i * n + n - 1] = q - 1
    out = b32.compose_multiple_values_to(res, n)
    out64 = b64.compose_multiple_values_to(res.astype(np.uint64), n)
    for c in range(n):
        v = crt_compose([int(res[i * n + c]) for i in range(len(moduli))], moduli)
        assert limbs32_to_int(out[c * vl:(c + 1) * vl]) == v
        assert pyref.limbs_to_int(out64[c * b64.value_len:(c + 1) * b64.value_len]) == v
    assert np.array_equal(b32.decompose_big_uint_values_to(out, n), res)
    # centred lift and the scaled accumulations (base.rs:279-416) == the 64-bit restatement on the same integers
    for sm in (2, 3, 16):
        if sm >= min(moduli):
            continue
        small = rng.integers(0, sm, n, dtype=np.uint64).astype(np.uint32)
        lifted = b32.wrapping_decompose_small_values_to(small, sm)
        assert np.array_equal(lifted, b64.wrapping_decompose_small_values_to(small.astype(np.uint64), sm).astype(np.uint32))
        fv = [int(rng.integers(0, q)) for q in moduli]
        acc = rand32(rng, moduli, n)
        acc64 = acc.astype(np.uint64)
        b32.add_wrapping_decompose_small_values_scaled(small, acc, sm, [shoup32(v, q) for v, q in zip(fv, moduli)])
        b64.add_wrapping_decompose_small_values_scaled(small.astype(np.uint64), acc64, sm, [(v, (v << 64) // q) for v, q in zip(fv, moduli)])
        assert np.array_equal(acc, acc64.astype(np.uint32))
        b32.add_decompose_small_values_scaled(small, acc, [shoup32(v, q) for v, q in zip(fv, moduli)])
        b64.add_decompose_small_values_scaled(small.astype(np.uint64), acc64, [(v, (v << 64) // q) for v, q in zip(fv, moduli)])
        assert np.array_equal(acc, acc64.astype(np.uint32))


def test_rns32_errors(orc):
    with pytest.raises(orc.OracleError):
        orc.RNSBase32([])
    with pytest.raises(orc.OracleError):
        orc.RNSBase32([21, 35])
    with pytest.raises(orc.OracleError):
        orc.RNSBase32([1 << 30, 97])     # BarrettModulus::<u32>::new: leading_zeros > 1
    with pytest.raises(orc.OracleError):
        orc.BigUintApproxSignedBasis32(orc.RNSBase32(Q30), 32)   # basis.rs:51: log_basis < T::BITS


@pytest.mark.parametrize("moduli,log_basis,rev", [(REF_U32, 7, None), (REF_U32, 6, None), (Q30, 15, None), (Q30, 15, 4),
                                                  (Q30, 29, None), (Q30, 1, None), (Q30, 31, 2), (Q30[:1], 10, None),
                                                  (ntt_primes_below(9, 28, 4), 13, None), (ntt_primes_below(16, 30, 4), 20, 9)])
def test_basis32_digits(orc, moduli, log_basis, rev):
    """Digits == pyref.Gadget (Python integers) and == the 64-bit restatement; the reference test's recomposition bound
    (big_uint.rs:84-131): |sum_j d_j * scalar_j - v| <= 2^(drop - 1) modulo Q."""
    rng = np.random.default_rng(log_basis)
    b32, b64 = orc.RNSBase32(moduli), orc.RNSBase(moduli)
    s32, s64 = orc.BigUintApproxSignedBasis32(b32, log_basis, rev), orc.BigUintApproxSignedBasis(b64, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    assert (s32.decompose_length, s32.drop_bits, s32.basis_value, s32.init_mode) == (g.ell, g.drop, g.B, s64.init_mode)
    vl, n = b32.value_len, 200
    for j in range(g.ell):
        assert limbs32_to_int(s32.scalars[j * vl:(j + 1) * vl]) == g.scalar(j)
        assert s32.scalars_residue[j * len(moduli):(j + 1) * len(moduli)].tolist() == [g.scalar(j) % q for q in moduli]
    vals = [int.from_bytes(rng.bytes(4 * vl + 8), "little") % g.Q for _ in range(n)]
    vals[:5] = [0, 1, g.Q - 1, g.Q // 2, g.threshold or 1]
    v = np.concatenate([int_to_limbs32(x, vl) for x in vals])
    adj, carries = s32.init_value_carry_slice_to(v, n)
    v2 = v.copy()
    c2 = s32.init_value_carry_slice_inplace(v2, n)
    assert np.array_equal(adj, v2) and np.array_equal(carries, c2)
    v64 = np.concatenate([pyref.int_to_limbs(x, b64.value_len) for x in vals])
    c64 = s64.init_value_carry_slice_inplace(v64, n)
    assert np.array_equal(carries, c64)
    assert [limbs32_to_int(adj[c * vl:(c + 1) * vl]) for c in range(n)] == \
        [pyref.limbs_to_int(v64[c * b64.value_len:(c + 1) * b64.value_len]) for c in range(n)]
    digits = []
    for j in range(g.ell):
        sd = s32.decompose_slice_to(j, adj, carries.copy(), n)
        sd64 = s64.decompose_slice_to(j, v64, c64.copy(), n)
        assert [limbs32_to_int(sd[c * vl:(c + 1) * vl]) for c in range(n)] == \
            [pyref.limbs_to_int(sd64[c * b64.value_len:(c + 1) * b64.value_len]) for c in range(n)]
        d = s32.unsigned_decompose_slice_to(j, adj, carries, n)
        d64 = s64.unsigned_decompose_slice_to(j, v64, c64, n)
        assert np.array_equal(d, d64.astype(np.uint32)) and np.array_equal(carries, c64)
        digits.append(d)
    for c, x in enumerate(vals):
        assert [int(d[c]) for d in digits] == g.unsigned_digits(x)
        rec = sum(sdg * g.scalar(j) for j, sdg in enumerate(g.signed_digits(x)))
        err = (rec - x) % g.Q
        assert min(err, g.Q - err) <= (1 << max(g.drop - 1, 0))


@pytest.mark.parametrize("moduli,log_n,k,log_basis,rev", [(Q30, 3, 1, 15, None), (REF_U32, 4, 2, 7, None),
                                                          (ntt_primes_below(9, 28, 4), 4, 1, 13, 5)])
def test_external_product32_equals_schoolbook(orc, moduli, log_n, k, log_basis, rev):
    """CrtGlwe<u32>::mul_dcrt_ggsw_to over U32DcrtTable == sum_i sum_j digit_ij (*) key_ij on Python integers, and the GLev
    row on big-integer input == the same row on residues."""
    rng = np.random.default_rng(log_n + k)
    n, L = 1 << log_n, len(moduli)
    table, base = orc.U32DcrtTable(log_n, moduli), orc.RNSBase32(moduli)
    basis = orc.BigUintApproxSignedBasis32(base, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    ell = g.ell
    glwe = rand32(rng, moduli, n, k + 1)
    key_coeff = rand32(rng, moduli, n, (k + 1) * ell * (k + 1))
    ggsw = key_coeff.copy()
    table.transform_slice(ggsw)
    out = orc.mul_dcrt32_ggsw_to(table, base, basis, k, glwe, ggsw)
    table.inverse_transform_slice(out)
    exp = pyref.external_product_coeff(moduli, n, k, g, glwe.reshape(k + 1, L, n).tolist(),
                                       key_coeff.reshape(k + 1, ell, k + 1, L, n).tolist())
    assert out.reshape(k + 1, L, n).tolist() == exp
    W = L * n
    poly, glev = glwe[:W].copy(), ggsw[:ell * (k + 1) * W].copy()
    a1 = rand32(rng, moduli, n, k + 1)
    a2 = a1.copy()
    orc.add_dcrt32_glev_mul_crt_poly_assign(table, base, basis, k, a1, glev, poly)
    orc.add_dcrt32_glev_mul_big_uint_poly_assign(table, base, basis, k, a2, glev, base.compose_multiple_values_to(poly, n))
    assert np.array_equal(a1, a2)


# ---- BaseConverter<u32> (primus_rns/src/converter.rs with T = u32) ----
CONV32 = [([17, 19, 23], [29, 31]), (Q30, REF_U32), (Q30[:2], Q30[2:] + REF_U32), (REF_U32, Q30),
          (ntt_primes_below(9, 30, 4), ntt_primes_below(2, 29, 4)), (ntt_primes_below(20, 30, 4), ntt_primes_below(3, 28, 4)),
          (ntt_primes_below(32, 30, 4), ntt_primes_below(2, 27, 4)), (Q30, ntt_primes_below(12, 29, 4))]


@pytest.mark.parametrize("mod_in,mod_out", CONV32, ids=lambda m: f"L{len(m)}")
def test_conv32_fast_convert_is_the_definition_and_the_64_bit_restatement(orc, mod_in, mod_out):
    """The reference's case (rns.rs:281-343: (17, 19, 23) -> (29, 31)) and wider bases, incl. more than 16 input moduli
    (two chunks of the 64-bit accumulator, slice.rs:387-404): sum_i [x_i (Q/q_i)^-1]_{q_i} (Q/q_i) mod p_j."""
    rng = np.random.default_rng(len(mod_in) * 100 + len(mod_out))
    n = 40
    conv = orc.BaseConverter32(orc.RNSBase32(mod_in), orc.RNSBase32(mod_out))
    conv64 = orc.BaseConverter(orc.RNSBase(mod_in), orc.RNSBase(mod_out))
    Q = 1
    for q in mod_in:
        Q *= q
    M = conv.base_change_matrix.reshape(len(mod_out), len(mod_in))
    for j, p in enumerate(mod_out):
        for i, q in enumerate(mod_in):
            assert int(M[j, i]) == (Q // q) % p
    x = rand32(rng, mod_in, n)
    x[0] = 0
    for i, q in enumerate(mod_in):
        x[i * n + n - 1] = q - 1
    out = conv.fast_convert_array(x, n)
    for t in range(n):
        s = sum((int(x[i * n + t]) * pow(Q // q, -1, q) % q) * (Q // q) for i, q in enumerate(mod_in))
        assert [int(out[j * n + t]) for j in range(len(mod_out))] == [s % p for p in mod_out]
    assert np.array_equal(out.astype(np.uint64), conv64.fast_convert_array(x.astype(np.uint64), n))


def test_conv32_accepts_unreduced_input_words(orc):
    """fill_fast_convert_array_scratch multiplies whatever word it is given (converter.rs:160-176): any u32, not only
    residues below q_i."""
    mod_in, mod_out = Q30, REF_U32
    conv = orc.BaseConverter32(orc.RNSBase32(mod_in), orc.RNSBase32(mod_out))
    rng = np.random.default_rng(7)
    n = 64
    x = rng.integers(0, 1 << 32, 3 * n, dtype=np.uint64).astype(np.uint32)
    x[:4] = 0xFFFFFFFF
    Q = mod_in[0] * mod_in[1] * mod_in[2]
    out = conv.fast_convert_array(x, n)
    for t in range(n):
        s = sum((int(x[i * n + t]) * pow(Q // q, -1, q) % q) * (Q // q) for i, q in enumerate(mod_in))
        assert [int(out[j * n + t]) for j in range(2)] == [s % p for p in mod_out]


@pytest.mark.parametrize("mod_in,p", [([17, 19, 23], 37), (Q30, REF_U32[0]), (Q30[:2], 134215681),
                                      (ntt_primes_below(9, 30, 4), 134176769), (ntt_primes_below(24, 29, 4), 1073479681)])
def test_conv32_exact_convert_is_the_centred_value(orc, mod_in, p):
    rng = np.random.default_rng(p % 1000 + len(mod_in))
    n = 64
    conv = orc.BaseConverter32(orc.RNSBase32(mod_in), orc.RNSBase32([p]))
    conv64 = orc.BaseConverter(orc.RNSBase(mod_in), orc.RNSBase([p]))
    Q = 1
    for q in mod_in:
        Q *= q
    x = rand32(rng, mod_in, n)
    x[0::n] = 0
    out = conv.exact_convert_array(x, n)
    checked = 0
    for t in range(n):
        v = crt_compose([int(x[i * n + t]) for i in range(len(mod_in))], mod_in)
        frac = v / Q
        if abs(frac - 0.5) < 1e-9:
            continue
        assert int(out[t]) == (v % p if frac < 0.5 else (v - Q) % p)
        checked += 1
    assert checked >= n - 2
    assert np.array_equal(out.astype(np.uint64), conv64.exact_convert_array(x.astype(np.uint64), n))
    with pytest.raises(orc.OracleError):
        orc.BaseConverter32(orc.RNSBase32(mod_in), orc.RNSBase32(REF_U32)).exact_convert_array(x, n)
