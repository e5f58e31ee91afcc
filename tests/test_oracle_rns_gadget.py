"""Oracle pin #3: RNS base, gadget decomposition and the RNS external product.

Re-creates primus_rns/tests/rns.rs:82-194 (incl. its closed-form cases),
primus_decompose/tests/big_uint.rs:139-439 (gadget property, slice == scalar) and supplies the
end-to-end schoolbook check the reference lacks for CrtGlwe::mul_dcrt_ggsw_to.
"""
import numpy as np
import pytest

import pyref
from pyref import Q61, crt_compose, int_to_limbs, limbs_to_int


def test_rns_rejects_bad_bases(orc):
    with pytest.raises(orc.OracleError) as e:
        orc.RNSBase([])
    assert e.value.code == 16  # EmptyBase (rns.rs:60-64)
    with pytest.raises(orc.OracleError) as e:
        orc.RNSBase([21, 35])
    assert e.value.code == 17  # CoPrimeError (rns.rs:68-72)


def test_rns_single_value_closed_form(orc):
    """rns.rs:77-98: base (3,5,7), residues (2,3,2) <-> 23."""
    base = orc.RNSBase([3, 5, 7])
    v = base.compose([2, 3, 2])
    assert limbs_to_int(v) == 23 and base.value_len == 1
    assert list(base.decompose(v)) == [2, 3, 2]


def test_rns_modulus_major_roundtrip(orc):
    """rns.rs:105-147 with the reference's own 50-bit primes and residue table."""
    moduli = [1_125_899_906_826_241, 1_125_899_906_629_633]
    base = orc.RNSBase(moduli)
    rows = [[0, 0], [1, 2], [97, 131], [moduli[0] - 1, moduli[1] - 2], [123_456_789, 987_654_321]]
    packed = np.array([r[i] for i in range(2) for r in rows], np.uint64)  # modulus-major
    vals = base.compose_multiple_values_to(packed, len(rows))
    for c, r in enumerate(rows):
        assert limbs_to_int(vals[c * base.value_len:(c + 1) * base.value_len]) == crt_compose(r, moduli)
    assert np.array_equal(base.decompose_big_uint_values_to(vals, len(rows)), packed)


@pytest.mark.parametrize("moduli", [Q61, [137438822401, 137438814209, 137438773249], [Q61[0]], Q61[:2]])
def test_rns_compose_matches_python(orc, moduli):
    rng = np.random.default_rng(len(moduli))
    base = orc.RNSBase(moduli)
    Q = int(np.prod([int(m) for m in moduli], dtype=object))
    assert limbs_to_int(base.moduli_product) == Q
    assert base.value_len == (Q.bit_length() + 63) // 64
    for i, m in enumerate(moduli):
        assert limbs_to_int(base.punctured_product[i * base.value_len:(i + 1) * base.value_len]) == Q // m
    n = 64
    res = np.concatenate([rng.integers(0, m, n, dtype=np.uint64) for m in moduli])
    # edge residues
    res[0] = 0
    for i, m in enumerate(moduli):
        res[i * n + 1] = m - 1
    vals = base.compose_multiple_values_to(res, n)
    for c in range(n):
        got = limbs_to_int(vals[c * base.value_len:(c + 1) * base.value_len])
        assert got == crt_compose([res[i * n + c] for i in range(len(moduli))], moduli)


def test_wrapping_decompose_rule(orc):
    """rns.rs:154-194 on base (97,101,103), small moduli 2, 7, 16."""
    moduli = [97, 101, 103]
    base = orc.RNSBase(moduli)
    for sm in (2, 7, 16):
        small = np.array([(i * 5 + 3) % sm for i in range(17)], np.uint64)
        got = base.wrapping_decompose_small_values_to(small, sm)
        exp = [v if (sm == 2 or v < -(-sm // 2)) else m - sm + v for m in moduli for v in map(int, small)]
        assert list(map(int, got)) == exp


@pytest.mark.parametrize("moduli,log_basis,rev", [
    (Q61, 30, None), (Q61, 30, 4), (Q61, 61, None), (Q61, 1, None), (Q61, 7, None),
    ([134215681, 134176769], 7, None), ([134215681, 134176769], 6, None), (Q61[:2], 13, 5),
    ([137438822401, 137438814209, 137438773249], 15, None), (Q61[:1], 20, None),
])
def test_gadget_matches_python_and_property(orc, moduli, log_basis, rev):
    """big_uint.rs:139-322: |sum_j d_j 2^(drop+j logB) - v| <= 2^(drop-1) (mod Q); slice == scalar;
    signed == unsigned + centred lift (big_uint.rs:325-401)."""
    rng = np.random.default_rng(log_basis)
    base = orc.RNSBase(moduli)
    basis = orc.BigUintApproxSignedBasis(base, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    assert (basis.decompose_length, basis.drop_bits) == (g.ell, g.drop)
    L, Q = base.value_len, g.Q
    for j in range(g.ell):
        assert limbs_to_int(basis.scalars[j * L:(j + 1) * L]) == g.scalar(j)
        for i, m in enumerate(moduli):
            assert int(basis.scalars_residue[j * len(moduli) + i]) == g.scalar(j) % m
    n = 257
    vals_int = [int.from_bytes(rng.bytes(40), "little") % Q for _ in range(n)]
    vals_int[:6] = [0, 1, Q - 1, Q // 2, (g.threshold or 1) - 1, g.threshold or 1]
    values = np.concatenate([int_to_limbs(v, L) for v in vals_int])
    carries = basis.init_value_carry_slice_inplace(values, n)
    digits = [basis.unsigned_decompose_slice_to(j, values, carries, n) for j in range(g.ell)]
    bound = (1 << (g.drop - 1)) if g.drop > 0 else 0
    half = (g.B + 1) // 2
    for c, v in enumerate(vals_int):
        us = [int(digits[j][c]) for j in range(g.ell)]
        assert us == g.unsigned_digits(v)
        signed = us if g.B == 2 else [u if u < half else u - g.B for u in us]
        recomposed = sum(d * g.scalar(j) for j, d in enumerate(signed)) % Q
        diff = (recomposed - v) % Q
        assert min(diff, Q - diff) <= bound
    # centred lift of the digits == residues of the signed digits
    lifted = base.wrapping_decompose_small_values_to(digits[0], g.B)
    for i, m in enumerate(moduli):
        for c in range(0, n, 17):
            u = int(digits[0][c])
            s = u if (g.B == 2 or u < half) else u - g.B
            assert int(lifted[i * n + c]) == s % m


@pytest.mark.parametrize("log_n,k,moduli,log_basis,rev", [
    (3, 1, Q61, 30, None), (4, 1, Q61, 30, None), (3, 2, Q61[:2], 20, 3), (4, 1, [134215681, 134176769], 7, None),
])
def test_external_product_matches_schoolbook(orc, log_n, k, moduli, log_basis, rev):
    """CrtGlwe::mul_dcrt_ggsw_to (glwe/crt.rs:200-227) == sum_i sum_j digit_ij (*) key_ij."""
    rng = np.random.default_rng(log_n * 10 + k)
    n, Lm = 1 << log_n, len(moduli)
    table = orc.U64DcrtTable(log_n, moduli)
    base = orc.RNSBase(moduli)
    basis = orc.BigUintApproxSignedBasis(base, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    ell, W = g.ell, Lm * n
    glwe = np.concatenate([rng.integers(0, m, n, dtype=np.uint64) for _ in range(k + 1) for m in moduli])
    key_coeff = np.concatenate([rng.integers(0, m, n, dtype=np.uint64)
                                for _ in range((k + 1) * ell * (k + 1)) for m in moduli])
    ggsw = key_coeff.copy()
    table.transform_slice(ggsw)  # every (row, level, component) polynomial to DCRT form
    out = orc.mul_dcrt_ggsw_to(table, base, basis, k, glwe, ggsw)
    table.inverse_transform_slice(out)
    exp = pyref.external_product_coeff(
        moduli, n, k, g,
        glwe.reshape(k + 1, Lm, n).tolist(),
        key_coeff.reshape(k + 1, ell, k + 1, Lm, n).tolist())
    assert out.reshape(k + 1, Lm, n).tolist() == exp
