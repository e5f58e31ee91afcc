"""RNS bases of more than eight moduli (RNSBase::new, primus_rns/src/base.rs:79-117, and BaseConverter::new,
converter.rs:43-69, take any number): compose / decompose / gadget steps / base conversion / element-wise family /
external product through the C ABI against the oracle at L = 9, 12, 16, 24, 32, against Python integers, and
against the schoolbook product.

Bases of at most 8 moduli carry their constants as kernel arguments; wider ones in a device table (csrc/pfhe_rns.hpp:
RnsWide / BasisWide) with kernels instantiated for the limb count rounded up to a multiple of four — so the cases
below also sit on both sides of every rounding step (value_len 9, 12 | 13, 16 | 17, ...).
"""
import numpy as np
import pytest

import pyref
from gpu_util import rand_rns, to_dev, to_host
from primes import ntt_primes_below
from pyref import crt_compose, int_to_limbs, limbs_to_int

pytestmark = pytest.mark.gpu

# (L, bits per modulus): value_len = 9, 9, 13, 16, 12, 17, 24, 31; (10, 20) -> 4 limbs from ten moduli (wide base,
# by-value basis), (9, 7) -> one limb
BASES = [(9, 61), (12, 45), (13, 61), (16, 61), (24, 30), (17, 62), (24, 61), (32, 61), (10, 20), (9, 7)]


def base_moduli(L, bits, log_n=None):
    return ntt_primes_below(L, bits, (0 if bits < 12 else 4) if log_n is None else log_n)


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def test_limits(pf):
    """32 moduli are accepted, 33 refused (PFHE_ERR_UNSUPPORTED, after the reference's own errors)."""
    m = ntt_primes_below(33, 40, 1)
    assert pf.RNSBase(m[:32]).moduli_count() == 32
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase(m)
    assert e.value.kind == "Unsupported"
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase(m[:32] + [m[0]])
    assert e.value.kind == "CoPrimeError"


@pytest.mark.parametrize("L,bits", BASES)
@pytest.mark.parametrize("count", [1, 777])
def test_compose_and_decompose(pf, orc, L, bits, count):
    moduli = base_moduli(L, bits)
    rng = np.random.default_rng(L * 100 + bits)
    base, obase = pf.RNSBase(moduli), orc.RNSBase(moduli)
    vl = base.big_uint_value_len()
    assert vl == obase.value_len and np.array_equal(base.moduli_product(), obase.moduli_product)
    res = rand_rns(rng, moduli, count)
    for i, m in enumerate(moduli):   # extreme columns: all zero, all maximal
        res[i * count] = 0
        res[i * count + count - 1] = m - 1
    out = np.empty(count * vl, np.uint64)
    base.compose_multiple_values_to(res, out, count)
    assert np.array_equal(out, obase.compose_multiple_values_to(res, count))
    for c in sorted({0, count // 2, count - 1}):
        assert limbs_to_int(out[c * vl:(c + 1) * vl]) == crt_compose([int(res[i * count + c]) for i in range(L)], moduli)
    # decompose_big_uint_values_to (base.rs:457-481) takes the composed values back
    back = np.empty_like(res)
    base.decompose_big_uint_values_to(out, back, count)
    assert np.array_equal(back, res)
    assert np.array_equal(back, obase.decompose_big_uint_values_to(out, count))
    # the device-pointer forms
    dout, dback = to_dev(np.zeros_like(out)), to_dev(np.zeros_like(res))
    base.compose_multiple_values_to_dev(to_dev(res), dout, count)
    base.decompose_big_uint_values_to_dev(dout, dback, count)
    assert np.array_equal(to_host(dout), out) and np.array_equal(to_host(dback), res)


@pytest.mark.parametrize("L,bits,log_basis,rev", [(9, 61, 30, None), (12, 45, 13, 7), (13, 61, 61, None), (16, 61, 30, 11),
                                                  (24, 30, 7, None), (17, 62, 45, None), (32, 61, 30, None),
                                                  (10, 20, 9, None), (9, 7, 3, None), (9, 61, 1, 40)])
def test_gadget_steps(pf, orc, L, bits, log_basis, rev):
    """init_value_carry / unsigned digits / signed digits modulo Q / centred lift / scaled accumulation, slice by slice."""
    moduli = base_moduli(L, bits)
    rng = np.random.default_rng(L + log_basis)
    base, obase = pf.RNSBase(moduli), orc.RNSBase(moduli)
    basis, obasis = pf.BigUintApproxSignedBasis(base, log_basis, rev), orc.BigUintApproxSignedBasis(obase, log_basis, rev)
    assert (basis.decompose_length(), basis.log_basis(), basis.drop_bits(), basis.basis_value()) == \
        (obasis.decompose_length, obasis.log_basis, obasis.drop_bits, obasis.basis_value)
    assert np.array_equal(basis.scalars(), obasis.scalars)
    assert np.array_equal(basis.scalars_residue(), obasis.scalars_residue)
    g = pyref.Gadget(moduli, log_basis, rev)
    vl, n = base.big_uint_value_len(), 300
    vals_int = [int.from_bytes(rng.bytes(8 * vl + 8), "little") % g.Q for _ in range(n)]
    vals_int[:6] = [0, 1, g.Q - 1, g.Q // 2, (g.threshold or 1) - 1, g.threshold or 1]
    values = np.concatenate([int_to_limbs(v, vl) for v in vals_int])
    ov = values.copy()
    oc = obasis.init_value_carry_slice_inplace(ov, n)
    gv, gc = values.copy(), np.zeros(n, np.uint8)
    basis.init_value_carry_slice_inplace(gv, gc)
    assert np.array_equal(gv, ov) and np.array_equal(gc, oc)
    lift = basis.basis_value() < min(moduli)
    acc, oacc = rand_rns(rng, moduli, n), None
    oacc = acc.copy()
    factors = [int(rng.integers(0, q)) for q in moduli]
    fpairs = [(f, (f << 64) // q) for f, q in zip(factors, moduli)]
    for j in range(basis.decompose_length()):
        sv, sc = gv.copy(), gc.copy()
        osd = obasis.decompose_slice_to(j, ov, oc.copy(), n)
        sd = np.empty_like(gv)
        basis.decompose_slice_to(j, sv, sd, sc)
        assert np.array_equal(sd, osd), j
        od = obasis.unsigned_decompose_slice_to(j, ov, oc, n)
        gd = np.empty(n, np.uint64)
        basis.unsigned_decompose_slice_to(j, gv, gd, gc)
        assert np.array_equal(gd, od) and np.array_equal(gc, oc) and np.array_equal(sc, gc), j
        if lift:  # the centred lift needs B < q_i (base.rs:288-292)
            lifted = np.empty(L * n, np.uint64)
            base.wrapping_decompose_small_values_to(gd, lifted, n, basis.basis_value())
            assert np.array_equal(lifted, obase.wrapping_decompose_small_values_to(od, obasis.basis_value))
            base.add_wrapping_decompose_small_values_scaled(gd, acc, n, basis.basis_value(), fpairs)
            obase.add_wrapping_decompose_small_values_scaled(od, oacc, obasis.basis_value, fpairs)
            assert np.array_equal(acc, oacc), j


@pytest.mark.parametrize("lin,bin_,lout,bout", [(9, 61, 2, 60), (3, 61, 9, 60), (12, 45, 12, 44), (16, 61, 5, 50),
                                               (24, 30, 24, 29), (32, 61, 32, 60), (17, 62, 1, 61), (2, 60, 32, 61)])
@pytest.mark.parametrize("n", [5, 4099])
def test_base_converter(pf, orc, lin, bin_, lout, bout, n):
    mod_in, mod_out = base_moduli(lin, bin_), base_moduli(lout, bout)
    rng = np.random.default_rng(lin * 37 + lout)
    conv = pf.BaseConverter(pf.RNSBase(mod_in), pf.RNSBase(mod_out))
    oin, oout = orc.RNSBase(mod_in), orc.RNSBase(mod_out)
    oconv = orc.BaseConverter(oin, oout)
    assert (conv.input_moduli_count(), conv.output_moduli_count()) == (lin, lout)
    assert np.array_equal(conv.base_change_matrix(), oconv.base_change_matrix)
    x = rand_rns(rng, mod_in, n)
    x[0] = 0
    for i, q in enumerate(mod_in):
        x[i * n + n - 1] = q - 1
    out = np.empty(lout * n, np.uint64)
    conv.fast_convert_array(x, out, n)
    assert np.array_equal(out, oconv.fast_convert_array(x, n))
    # fast conversion = (sum_i t_i * (Q/q_i)) mod p_j on Python integers
    Q = 1
    for q in mod_in:
        Q *= q
    for c in (0, n // 2, n - 1):
        s = sum(((int(x[i * n + c]) * pow(Q // q, -1, q)) % q) * (Q // q) for i, q in enumerate(mod_in))
        assert [int(out[j * n + c]) for j in range(lout)] == [s % p for p in mod_out]
    # exact conversion to the first output modulus: identical f64 correction term, summed in the same order
    e = pf.BaseConverter(pf.RNSBase(mod_in), pf.RNSBase(mod_out[:1]))
    oe = orc.BaseConverter(oin, orc.RNSBase(mod_out[:1]))
    eo = np.empty(n, np.uint64)
    e.exact_convert_array(x, eo, n)
    assert np.array_equal(eo, oe.exact_convert_array(x, n))
    if lout == 2:
        pairs = to_dev(np.zeros(2 * n, np.uint64))
        conv.fast_convert_array_to_pairs_dev(to_dev(x), pairs, n)
        assert np.array_equal(to_host(pairs).reshape(n, 2).T.reshape(-1), out)


@pytest.mark.parametrize("L,bits,log_n,batch", [(9, 61, 4, 3), (12, 45, 7, 2), (16, 61, 11, 1), (24, 30, 5, 2), (32, 61, 9, 1),
                                                (33, 40, 3, 1)])
def test_elementwise_family(pf, orc, L, bits, log_n, batch):
    """CrtPolynomial / DcrtPolynomial ops per limb; add / sub / neg / monomial / inv take any limb count (33 here), the
    per-limb scalar forms up to 32."""
    from test_elementwise import run_gpu
    moduli = base_moduli(L, bits, log_n)
    rng = np.random.default_rng(L + log_n)
    n = 1 << log_n
    a, b = rand_rns(rng, moduli, n, batch), rand_rns(rng, moduli, n, batch)
    a[a == 0] = 1
    o = orc.CrtPolyOps(moduli, n)
    r = int(rng.integers(0, 2 * n))
    if L > 32:
        import torch
        t = pf.U64DcrtTable(log_n, moduli)
        da, db = to_dev(a), to_dev(b)
        out = torch.empty_like(da)
        t.add_to_dev(da, db, out); assert np.array_equal(to_host(out), o.add_to(a, b))
        t.sub_to_dev(da, db, out); assert np.array_equal(to_host(out), o.sub_to(a, b))
        t.neg_to_dev(da, out); assert np.array_equal(to_host(out), o.neg_to(a))
        t.inv_to_dev(da, out); assert np.array_equal(to_host(out), o.inv_to(a))
        t.mul_monomial_to_dev(da, r, out)
        m = a.copy(); o.mul_monomial_assign(m, r); assert np.array_equal(to_host(out), m)
        with pytest.raises(pf.PfheError) as e:
            t.mul_scalar_to_dev(da, [1] * L, out)
        assert e.value.kind == "Unsupported"
        return
    scalars = [int(rng.integers(0, q)) for q in moduli]
    factors = [v for s, q in zip(scalars, moduli) for v in (s, (s << 64) // q)]
    got = run_gpu(pf, moduli, log_n, a, b, scalars, factors, r)
    m = a.copy(); o.mul_monomial_assign(m, r)
    assert np.array_equal(got["mul_monomial"], m)
    assert np.array_equal(got["add"], o.add_to(a, b))
    assert np.array_equal(got["sub"], o.sub_to(a, b))
    assert np.array_equal(got["neg"], o.neg_to(a))
    assert np.array_equal(got["mul_scalar"], o.mul_scalar_to(a, scalars))
    assert np.array_equal(got["mul_factor"], o.mul_factor_to(a, factors))
    assert np.array_equal(got["inv"], o.inv_to(a))
    acc = a.copy(); o.add_mul_scalar_assign(acc, b, scalars); assert np.array_equal(got["add_mul_scalar"], acc)
    acc = a.copy(); o.add_mul_factor_assign(acc, b, factors); assert np.array_equal(got["add_mul_factor"], acc)


def make_case(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared):
    from test_gpu_rns_gadget import make_case as mk
    return mk(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared)


@pytest.mark.parametrize("L,bits,log_n,k,log_basis,rev,batch,shared", [
    (9, 61, 4, 1, 30, 6, 3, True),      # unfused kernels (tiny ring), 9 limbs
    (9, 61, 12, 1, 30, 6, 2, False),    # single block pass
    (9, 61, 16, 1, 30, 4, 1, True),     # two passes: signed digits + lifting strided pass, fused block multiply-accumulate
    (12, 45, 13, 2, 20, 5, 1, True),    # k = 2, value_len 9
    (16, 61, 15, 1, 45, 3, 1, True),    # int64 digits, value_len 16
    (10, 20, 10, 1, 9, None, 3, True),  # wide base, 4-limb integers: the small-ring kernel (int32 digits)
    (24, 30, 16, 1, 13, 3, 1, True),    # 24 limbs of 30 bits
])
def test_external_product(pf, orc, L, bits, log_n, k, log_basis, rev, batch, shared):
    """CrtGlwe::mul_dcrt_ggsw_to (glwe/crt.rs:200-227) over a wide base, NTT-form and coefficient-form output, and
    the GLev row forms on CRT and on big-integer input."""
    moduli = base_moduli(L, bits, log_n)
    rng = np.random.default_rng(L * 7 + log_n)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared)
    table, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    basis = pf.BigUintApproxSignedBasis(base, log_basis, rev)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
    assert np.array_equal(out, exp)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx, into_coeff_form=True)
    otable.inverse_transform_slice(exp)
    assert np.array_equal(out, exp)
    # one GLev row against a polynomial given as big integers == the same polynomial given as residues
    n, ell = 1 << log_n, basis.decompose_length()
    W = L * n
    poly = glwe[:W].copy()
    glev = ggsw[:ell * (k + 1) * W].copy()
    vl = base.big_uint_value_len()
    big = np.empty(n * vl, np.uint64)
    base.compose_multiple_values_to(poly, big, n)
    r1, r2 = to_dev(np.zeros((k + 1) * W, np.uint64)), to_dev(np.zeros((k + 1) * W, np.uint64))
    pf.glev_mul_crt_poly_to_dev(to_dev(glev), to_dev(poly), r1, ctx)
    pf.glev_mul_big_uint_poly_to_dev(to_dev(glev), to_dev(big), r2, ctx)
    assert np.array_equal(to_host(r1), to_host(r2))
    obase, obasis = orc.RNSBase(moduli), None
    obasis = orc.BigUintApproxSignedBasis(obase, log_basis, rev)
    acc = np.zeros((k + 1) * W, np.uint64)
    orc.add_dcrt_glev_mul_crt_poly_assign(otable, obase, obasis, k, acc, glev, poly)
    assert np.array_equal(to_host(r1), acc)


def test_external_product_equals_schoolbook(pf):
    """End-to-end ground truth on Python integers at L = 9, N = 2^4: sum_i sum_j digit_ij (*) key_ij mod (X^N+1, q_r)."""
    log_n, k, L, log_basis, rev = 4, 1, 9, 30, 5
    moduli = base_moduli(L, 61, log_n)
    rng = np.random.default_rng(4242)
    n = 1 << log_n
    table, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    basis = pf.BigUintApproxSignedBasis(base, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    ell = g.ell
    assert ell == basis.decompose_length() == rev
    glwe = rand_rns(rng, moduli, n, k + 1)
    key_coeff = rand_rns(rng, moduli, n, (k + 1) * ell * (k + 1))
    ggsw = key_coeff.copy()
    table.transform_slice(ggsw)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx, into_coeff_form=True)
    exp = pyref.external_product_coeff(moduli, n, k, g, glwe.reshape(k + 1, L, n).tolist(),
                                       key_coeff.reshape(k + 1, ell, k + 1, L, n).tolist())
    assert out.reshape(k + 1, L, n).tolist() == exp


@pytest.mark.parametrize("word_bits", [64, 32])
def test_derived_handles_outlive_their_base(pf, orc, word_bits):
    """A wide base keeps its constants in a device table that the basis, the converter and the plan derived from it share
    (reference-counted: include/pfhe.h, pfhe_rns_create).  Destroying the pfhe_rns first must leave every derived handle
    working — the reference's types own clones of the base (converter.rs:64-68) — and destroying the last of them frees
    the table."""
    import gc
    if word_bits == 64:
        RNS, Basis, Conv, Ctx, Table, dt = pf.RNSBase, pf.BigUintApproxSignedBasis, pf.BaseConverter, pf.DcrtGlevContext, pf.U64DcrtTable, np.uint64
        oRNS, oBasis, oConv = orc.RNSBase, orc.BigUintApproxSignedBasis, orc.BaseConverter
        log_n, k, log_basis, rev = 7, 1, 13, 5
        moduli, mod_out = base_moduli(12, 45, log_n), ntt_primes_below(2, 44, 4)
    else:
        RNS, Basis, Conv, Ctx, Table, dt = (pf.RNSBase32, pf.BigUintApproxSignedBasis32, pf.BaseConverter32, pf.DcrtGlevContext32,
                                            pf.U32DcrtTable, np.uint32)
        oRNS, oBasis, oConv = orc.RNSBase32, orc.BigUintApproxSignedBasis32, orc.BaseConverter32
        log_n, k, log_basis, rev = 7, 1, 13, 5
        moduli, mod_out = ntt_primes_below(12, 30, log_n), ntt_primes_below(2, 29, 4)
    n, L = 1 << log_n, len(moduli)
    rng = np.random.default_rng(word_bits)
    count_of = lambda: int(pf.lib().pfhe_debug_alloc_count())  # noqa: E731
    base, out_base = RNS(moduli), RNS(mod_out)
    basis = Basis(base, log_basis, rev)
    conv = Conv(base, out_base)
    table = Table(log_n, moduli)
    ctx = Ctx(table, base, basis, k)
    ell = basis.decompose_length()
    # destroy the base (and the narrow output base) at the C level; the Python objects must not destroy them again
    for b in (base, out_base):
        b._f("destroy")(b._h)
        b._h = None
    gc.collect()
    before = count_of()
    obase = oRNS(moduli)
    obasis = oBasis(obase, log_basis, rev)
    res = np.concatenate([rng.integers(0, q, n, dtype=np.uint64).astype(dt) for q in moduli])
    out = np.empty(2 * n, dt)
    conv.fast_convert_array(res, out, n)
    assert np.array_equal(out, oConv(obase, oRNS(mod_out)).fast_convert_array(res, n))
    big = obase.compose_multiple_values_to(res, n)
    adj, car = np.empty_like(big), np.zeros(n, np.uint8)
    basis.init_value_carry_slice_to(big, adj, car)
    oadj, ocar = obasis.init_value_carry_slice_to(big.copy(), n)
    assert np.array_equal(adj, oadj) and np.array_equal(car, ocar)
    for level in range(ell):
        dig = np.empty(n, dt)
        basis.unsigned_decompose_slice_to(level, adj, dig, car)
        assert np.array_equal(dig, obasis.unsigned_decompose_slice_to(level, oadj, ocar, n))
    glwe = np.concatenate([rng.integers(0, q, n, dtype=np.uint64).astype(dt) for _ in range(k + 1) for q in moduli])
    ggsw = np.concatenate([rng.integers(0, q, n, dtype=np.uint64).astype(dt) for _ in range((k + 1) * ell * (k + 1)) for q in moduli])
    got = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, got, ctx)
    if word_bits == 64:
        exp = orc.mul_dcrt_ggsw_to(orc.U64DcrtTable(log_n, moduli), obase, obasis, k, glwe.copy(), ggsw)
    else:
        exp = orc.mul_dcrt32_ggsw_to(orc.U32DcrtTable(log_n, moduli), obase, obasis, k, glwe.copy(), ggsw)
    assert np.array_equal(got, exp)
    assert L > 8 and count_of() >= before
    del conv
    gc.collect()
    mid = count_of()
    del ctx, basis
    gc.collect()
    assert count_of() > mid                          # the plan's scratch and, with the last holder, the shared tables
