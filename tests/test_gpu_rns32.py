"""GPU parity of the <u32> instantiations — RNSBase<u32>, BigUintApproxSignedBasis<u32>, CrtGlwe<u32>::mul_dcrt_ggsw_to over
U32DcrtTable (primus_rns/src/base.rs:26-37, primus_decompose/src/big_integer/basis.rs:33, primus_lattice/src/glwe/crt.rs:200-227,
primus_ntt/src/dcrt/prime32.rs:11) — through the C ABI (pfhe_rns32_*, pfhe_basis32_*, pfhe_extprod32_*) against the
oracle's 32-bit-limb restatement (oracle/pfhe_oracle_rns32.c), Python integers and the schoolbook product.

The device code runs the 64-bit kernels with 32-bit words in memory; the oracle does 32-bit limb arithmetic throughout, so
a disagreement in limb packing, limb count or word-straddling windows shows here.
"""
import numpy as np
import pytest

import pyref
from primes import ntt_primes_below
from pyref import crt_compose
from test_gpu_u32 import to_dev32, to_host32
from test_oracle_rns32 import Q30, REF_U32, int_to_limbs32, limbs32_to_int, rand32, shoup32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


# L (narrow / wide), odd and even numbers of u32 limbs (value_len 2, 3, 1, 1, 5, 8, 15, 28)
BASES = [REF_U32, Q30, Q30[:1], [97, 101, 103], ntt_primes_below(5, 30, 4), ntt_primes_below(9, 28, 4),
         ntt_primes_below(16, 30, 4), ntt_primes_below(30, 30, 4)]


def test_errors(pf):
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase32([])
    assert e.value.kind == "EmptyBase"
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase32([21, 35])
    assert e.value.kind == "CoPrimeError"
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase32([1 << 30, 97])          # BarrettModulus::<u32>::new (barrett/mod.rs:39-44)
    assert e.value.kind == "UnrepresentableModulus"
    base = pf.RNSBase32(Q30)
    for lb in (0, 32):                       # basis.rs:51: 0 < log_basis < T::BITS
        with pytest.raises(pf.PfheError) as e:
            pf.BigUintApproxSignedBasis32(base, lb)
        assert e.value.kind == "BadArgument"
    with pytest.raises(TypeError):
        base.compose_multiple_values_to(np.zeros(3, np.uint64), np.zeros(3, np.uint32), 1)
    # B = 2^30 exceeds the moduli: fine for the basis, refused where the centred lift needs B < q_i (base.rs:288-292)
    with pytest.raises(pf.PfheError) as e:
        pf.DcrtGlevContext32(pf.U32DcrtTable(4, Q30), base, pf.BigUintApproxSignedBasis32(base, 30), 1)
    assert e.value.kind == "BadArgument"


@pytest.mark.parametrize("moduli", BASES, ids=lambda m: f"L{len(m)}_{m[0]}")
@pytest.mark.parametrize("count", [1, 1000])
def test_rns32_matches_oracle(pf, orc, moduli, count):
    rng = np.random.default_rng(len(moduli) + count)
    base, obase = pf.RNSBase32(moduli), orc.RNSBase32(moduli)
    vl, L = base.big_uint_value_len(), len(moduli)
    assert vl == obase.value_len and base.moduli_count() == L
    assert np.array_equal(base.moduli_product(), obase.moduli_product)
    res = rand32(rng, moduli, count)
    res[0] = 0
    for i, q in enumerate(moduli):
        res[i * count + count - 1] = q - 1
    out = np.empty(count * vl, np.uint32)
    base.compose_multiple_values_to(res, out, count)
    assert np.array_equal(out, obase.compose_multiple_values_to(res, count))
    c = count // 2
    assert limbs32_to_int(out[c * vl:(c + 1) * vl]) == crt_compose([int(res[i * count + c]) for i in range(L)], moduli)
    back = np.empty_like(res)
    base.decompose_big_uint_values_to(out, back, count)
    assert np.array_equal(back, res)
    dout, dback = to_dev32(np.zeros_like(out)), to_dev32(np.zeros_like(res))
    base.compose_multiple_values_to_dev(to_dev32(res), dout, count)
    base.decompose_big_uint_values_to_dev(dout, dback, count)
    assert np.array_equal(to_host32(dout), out) and np.array_equal(to_host32(dback), res)
    with pytest.raises(pf.PfheError) as e:
        base.compose_multiple_values_to(res[:-1].copy(), out, count)
    assert e.value.kind == "BadLength"
    for sm in (2, 3, 16):
        if sm >= min(moduli):
            continue
        small = rng.integers(0, sm, count, dtype=np.uint64).astype(np.uint32)
        lifted = np.empty(L * count, np.uint32)
        base.wrapping_decompose_small_values_to(small, lifted, count, sm)
        assert np.array_equal(lifted, obase.wrapping_decompose_small_values_to(small, sm))
        f = [shoup32(int(rng.integers(0, q)), q) for q in moduli]
        acc = rand32(rng, moduli, count)
        oacc = acc.copy()
        base.add_wrapping_decompose_small_values_scaled(small, acc, count, sm, f)
        obase.add_wrapping_decompose_small_values_scaled(small, oacc, sm, f)
        assert np.array_equal(acc, oacc)
        base.add_decompose_small_values_scaled(small, acc, count, f)
        obase.add_decompose_small_values_scaled(small, oacc, f)
        assert np.array_equal(acc, oacc)


@pytest.mark.parametrize("moduli,log_basis,rev", [(REF_U32, 7, None), (REF_U32, 6, None), (Q30, 15, None), (Q30, 15, 4),
                                                  (Q30, 29, None), (Q30, 1, None), (Q30, 31, 2), (Q30[:1], 10, None),
                                                  (ntt_primes_below(9, 28, 4), 13, None), (ntt_primes_below(16, 30, 4), 20, 9),
                                                  (ntt_primes_below(30, 30, 4), 31, None)])
def test_basis32_steps_match_oracle(pf, orc, moduli, log_basis, rev):
    rng = np.random.default_rng(log_basis + len(moduli))
    base, obase = pf.RNSBase32(moduli), orc.RNSBase32(moduli)
    basis, obasis = pf.BigUintApproxSignedBasis32(base, log_basis, rev), orc.BigUintApproxSignedBasis32(obase, log_basis, rev)
    assert (basis.decompose_length(), basis.log_basis(), basis.drop_bits(), basis.basis_value()) == \
        (obasis.decompose_length, obasis.log_basis, obasis.drop_bits, obasis.basis_value)
    assert np.array_equal(basis.scalars(), obasis.scalars)
    assert np.array_equal(basis.scalars_residue(), obasis.scalars_residue)
    g = pyref.Gadget(moduli, log_basis, rev)
    vl, n = base.big_uint_value_len(), 333
    vals = [int.from_bytes(rng.bytes(4 * vl + 8), "little") % g.Q for _ in range(n)]
    vals[:6] = [0, 1, g.Q - 1, g.Q // 2, (g.threshold or 1) - 1, g.threshold or 1]
    values = np.concatenate([int_to_limbs32(v, vl) for v in vals])
    ov = values.copy()
    oc = obasis.init_value_carry_slice_inplace(ov, n)
    gv, gc = values.copy(), np.zeros(n, np.uint8)
    basis.init_value_carry_slice_inplace(gv, gc)
    assert np.array_equal(gv, ov) and np.array_equal(gc, oc)
    adj, c2 = np.empty_like(values), np.zeros(n, np.uint8)
    basis.init_value_carry_slice_to(values, adj, c2)
    assert np.array_equal(adj, ov) and np.array_equal(c2, oc)
    for j in range(basis.decompose_length()):
        sc = gc.copy()
        sd = np.empty_like(gv)
        basis.decompose_slice_to(j, gv, sd, sc)
        assert np.array_equal(sd, obasis.decompose_slice_to(j, ov, oc.copy(), n)), j
        od = obasis.unsigned_decompose_slice_to(j, ov, oc, n)
        gd = np.empty(n, np.uint32)
        basis.unsigned_decompose_slice_to(j, gv, gd, gc)
        assert np.array_equal(gd, od) and np.array_equal(gc, oc) and np.array_equal(sc, gc), j
        assert [int(x) for x in gd[:6]] == [g.unsigned_digits(v)[j] for v in vals[:6]]


def make_case32(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared):
    n, L = 1 << log_n, len(moduli)
    otable, obase = orc.U32DcrtTable(log_n, moduli), orc.RNSBase32(moduli)
    obasis = orc.BigUintApproxSignedBasis32(obase, log_basis, rev)
    ell = obasis.decompose_length
    glwe = rand32(rng, moduli, n, batch * (k + 1))
    ggsw = rand32(rng, moduli, n, (1 if shared else batch) * (k + 1) * ell * (k + 1))
    G, K = (k + 1) * L * n, (k + 1) * ell * (k + 1) * L * n
    exp = np.concatenate([orc.mul_dcrt32_ggsw_to(otable, obase, obasis, k, glwe[e * G:(e + 1) * G].copy(),
                                                 ggsw[(0 if shared else e) * K:((0 if shared else e) + 1) * K].copy())
                          for e in range(batch)])
    return otable, obase, obasis, glwe, ggsw, exp


@pytest.mark.parametrize("log_n,k,moduli,log_basis,rev,batch,shared,chunk", [
    (3, 1, Q30, 15, None, 1, True, 0), (4, 1, Q30, 15, None, 5, True, 2), (4, 1, Q30, 15, None, 5, False, 3),
    (6, 2, Q30[:2], 10, 3, 3, True, 1), (10, 1, REF_U32, 7, None, 2, False, 0), (12, 1, Q30, 15, None, 3, True, 2),
    (13, 1, Q30, 13, None, 1, True, 0), (16, 1, Q30, 15, None, 2, False, 1), (15, 1, Q30, 29, None, 2, True, 0),
    (11, 1, ntt_primes_below(9, 28, 11), 13, 6, 2, True, 0), (16, 1, ntt_primes_below(12, 30, 16), 20, 4, 1, True, 0),
])
def test_external_product32_matches_oracle(pf, orc, log_n, k, moduli, log_basis, rev, batch, shared, chunk):
    rng = np.random.default_rng(log_n * 7 + batch)
    otable, obase, obasis, glwe, ggsw, exp = make_case32(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared)
    table, base = pf.U32DcrtTable(log_n, moduli), pf.RNSBase32(moduli)
    basis = pf.BigUintApproxSignedBasis32(base, log_basis, rev)
    ctx = pf.DcrtGlevContext32(table, base, basis, k, chunk)
    assert ctx.scratch_bytes() > 0 and not ctx.in_use()
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
    assert np.array_equal(out, exp)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx, into_coeff_form=True)
    otable.inverse_transform_slice(exp)
    assert np.array_equal(out, exp)
    with pytest.raises(pf.PfheError) as e:
        pf.mul_dcrt_ggsw_to(glwe, ggsw[:-1].copy(), out, ctx)
    assert e.value.kind == "BadLength"
    # GLev row forms: accumulate / overwrite, residues / big integers
    n, L, ell = 1 << log_n, len(moduli), basis.decompose_length()
    W = L * n
    poly, glev = glwe[:W].copy(), ggsw[:ell * (k + 1) * W].copy()
    acc0 = rand32(rng, moduli, n, k + 1)
    oacc = acc0.copy()
    orc.add_dcrt32_glev_mul_crt_poly_assign(otable, obase, obasis, k, oacc, glev, poly)
    dacc = to_dev32(acc0)
    pf.add_dcrt_glev_mul_crt_poly_assign_dev(dacc, to_dev32(glev), to_dev32(poly), ctx)
    assert np.array_equal(to_host32(dacc), oacc)
    big = np.empty(n * base.big_uint_value_len(), np.uint32)
    base.compose_multiple_values_to(poly, big, n)
    dacc2 = to_dev32(acc0)
    pf.add_dcrt_glev_mul_big_uint_poly_assign_dev(dacc2, to_dev32(glev), to_dev32(big), ctx)
    assert np.array_equal(to_host32(dacc2), oacc)
    r1, r2 = to_dev32(np.zeros((k + 1) * W, np.uint32)), to_dev32(np.zeros((k + 1) * W, np.uint32))
    pf.glev_mul_crt_poly_to_dev(to_dev32(glev), to_dev32(poly), r1, ctx)
    pf.glev_mul_big_uint_poly_to_dev(to_dev32(glev), to_dev32(big), r2, ctx)
    zero = np.zeros((k + 1) * W, np.uint32)
    orc.add_dcrt32_glev_mul_crt_poly_assign(otable, obase, obasis, k, zero, glev, poly)
    assert np.array_equal(to_host32(r1), zero) and np.array_equal(to_host32(r2), zero)


def test_external_product32_equals_schoolbook(pf):
    log_n, k, moduli, log_basis = 3, 1, Q30, 15
    rng = np.random.default_rng(42)
    n, L = 1 << log_n, 3
    table, base = pf.U32DcrtTable(log_n, moduli), pf.RNSBase32(moduli)
    basis = pf.BigUintApproxSignedBasis32(base, log_basis)
    g = pyref.Gadget(moduli, log_basis)
    ell = g.ell
    assert ell == basis.decompose_length() == 6
    glwe = rand32(rng, moduli, n, k + 1)
    key_coeff = rand32(rng, moduli, n, (k + 1) * ell * (k + 1))
    ggsw = key_coeff.copy()
    table.transform_slice(ggsw)
    ctx = pf.DcrtGlevContext32(table, base, basis, k)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx, into_coeff_form=True)
    exp = pyref.external_product_coeff(moduli, n, k, g, glwe.reshape(k + 1, L, n).tolist(),
                                       key_coeff.reshape(k + 1, ell, k + 1, L, n).tolist())
    assert out.reshape(k + 1, L, n).tolist() == exp


def test_bench_shape_full_batch_every_ciphertext(pf, orc):
    """The bench leg's shape (N = 2^16, three 30-bit primes, log B = 15 -> ell = 6, k = 1, batch 1024, one shared GGSW):
    every one of the 1024 products against the oracle, one ciphertext per task on all host cores."""
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from gpu_util import usable_cores
    log_n, k, batch = 16, 1, 1024
    n, L = 1 << log_n, 3
    table, base = pf.U32DcrtTable(log_n, Q30), pf.RNSBase32(Q30)
    basis = pf.BigUintApproxSignedBasis32(base, 15)
    ctx = pf.DcrtGlevContext32(table, base, basis, k)
    G = ctx.glwe_len()
    glwe = torch.empty(batch * G, dtype=torch.int32, device="cuda")
    ggsw = torch.empty(ctx.ggsw_len(), dtype=torch.int32, device="cuda")
    table.fill_uniform_dev(glwe, 0x5EED000000000432)
    table.fill_uniform_dev(ggsw, 99)
    out = torch.empty_like(glwe)
    pf.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx)
    torch.cuda.synchronize()
    otable, obase = orc.U32DcrtTable(log_n, Q30), orc.RNSBase32(Q30)
    obasis = orc.BigUintApproxSignedBasis32(obase, 15)
    hk, hg, ho = to_host32(ggsw), to_host32(glwe), to_host32(out)

    def one(e):
        exp = orc.mul_dcrt32_ggsw_to(otable, obase, obasis, k, hg[e * G:(e + 1) * G], hk)
        return bool(np.array_equal(ho[e * G:(e + 1) * G], exp))

    with ThreadPoolExecutor(usable_cores()) as ex:
        ok = list(ex.map(one, range(batch)))
    assert all(ok), [e for e, v in enumerate(ok) if not v][:10]


@pytest.mark.parametrize("batch,shared,chunk,log_basis", [(5, True, 0, 15), (7, False, 3, 15), (4, True, 0, 29), (9, True, 4, 7)])
def test_fused_u32_product_equals_separate_kernels_and_oracle(pf, orc, batch, shared, chunk, log_basis, monkeypatch):
    """N = 2^16, k = 1, at least four ciphertexts per call: balanced int32 digits, digits_strided32_kernel (lift + strided
    pass on B32Arith words) and gadget_block_mulacc32_kernel (block pass + multiply-accumulate + inverse block pass) — the
    same words as the plan created under PFHE_DISABLE_FUSED_EXTPROD (gadget_decompose + the table's transform +
    gadget_mulacc32) on the whole batch, NTT form and coefficient form, accumulating and overwriting GLev rows on residues
    and on big integers; the first two ciphertexts against the oracle."""
    import torch
    log_n, k = 16, 1
    n, L = 1 << log_n, 3
    rng = np.random.default_rng(batch * 31 + log_basis)
    otable, obase, obasis, glwe2, ggsw2, exp = make_case32(orc, rng, log_n, k, Q30, log_basis, None, 2, shared)
    table, base = pf.U32DcrtTable(log_n, Q30), pf.RNSBase32(Q30)
    basis = pf.BigUintApproxSignedBasis32(base, log_basis)
    ell = basis.decompose_length()
    ctx = pf.DcrtGlevContext32(table, base, basis, k, chunk)
    monkeypatch.setenv("PFHE_DISABLE_FUSED_EXTPROD", "1")
    ctx_sep = pf.DcrtGlevContext32(table, base, basis, k, chunk)
    monkeypatch.delenv("PFHE_DISABLE_FUSED_EXTPROD")
    assert ctx.scratch_bytes() > ctx_sep.scratch_bytes()      # the balanced-digit buffer of the fused kernels
    G, K = ctx.glwe_len(), ctx.ggsw_len()
    more = batch - 2
    full_g = np.concatenate([glwe2, rand32(rng, Q30, n, more * (k + 1))])
    full_k = ggsw2 if shared else np.concatenate([ggsw2, rand32(rng, Q30, n, more * (k + 1) * ell * (k + 1))])
    dg, dk = to_dev32(full_g), to_dev32(full_k)
    for coeff in (False, True):
        fused, sep = torch.zeros_like(dg), torch.zeros_like(dg)
        pf.mul_dcrt_ggsw_to_dev(dg, dk, fused, ctx, into_coeff_form=coeff)
        pf.mul_dcrt_ggsw_to_dev(dg, dk, sep, ctx_sep, into_coeff_form=coeff)
        assert torch.equal(fused, sep), coeff
        e = exp.copy()
        if coeff:
            otable.inverse_transform_slice(e)
        assert np.array_equal(to_host32(fused[:2 * G]), e), coeff
    # GLev rows: one row of the GGSW against `batch` polynomials, accumulate and overwrite, residues and big integers
    W = L * n
    polys = full_g[:batch * W].copy()
    glev = full_k[:ell * (k + 1) * W].copy()
    acc0 = rand32(rng, Q30, n, batch * (k + 1))
    a1, a2 = to_dev32(acc0), to_dev32(acc0)
    pf.add_dcrt_glev_mul_crt_poly_assign_dev(a1, to_dev32(glev), to_dev32(polys), ctx)
    pf.add_dcrt_glev_mul_crt_poly_assign_dev(a2, to_dev32(glev), to_dev32(polys), ctx_sep)
    assert torch.equal(a1, a2)
    oacc = acc0[:G].copy()
    orc.add_dcrt32_glev_mul_crt_poly_assign(otable, obase, obasis, k, oacc, glev, polys[:W].copy())
    assert np.array_equal(to_host32(a1[:G]), oacc)
    big = np.empty(batch * n * base.big_uint_value_len(), np.uint32)
    base.compose_multiple_values_to(np.ascontiguousarray(polys.reshape(batch, L, n).transpose(1, 0, 2)).reshape(-1), big, batch * n)
    r1, r2 = to_dev32(np.zeros(batch * G, np.uint32)), to_dev32(np.zeros(batch * G, np.uint32))
    pf.glev_mul_big_uint_poly_to_dev(to_dev32(glev), to_dev32(big), r1, ctx)
    pf.glev_mul_crt_poly_to_dev(to_dev32(glev), to_dev32(polys), r2, ctx_sep)
    assert torch.equal(r1, r2)


# ---- BaseConverter<u32> ----
from test_oracle_rns32 import CONV32  # noqa: E402


@pytest.mark.parametrize("mod_in,mod_out", CONV32, ids=lambda m: f"L{len(m)}")
@pytest.mark.parametrize("n", [1, 5, 4096, 65536 + 3])
def test_conv32_fast_and_exact_match_the_oracle(pf, orc, mod_in, mod_out, n):
    rng = np.random.default_rng(n + len(mod_in))
    conv = pf.BaseConverter32(pf.RNSBase32(mod_in), pf.RNSBase32(mod_out))
    oin = orc.RNSBase32(mod_in)
    oconv = orc.BaseConverter32(oin, orc.RNSBase32(mod_out))
    assert conv.input_moduli_count() == len(mod_in) and conv.output_moduli_count() == len(mod_out)
    assert np.array_equal(conv.base_change_matrix(), oconv.base_change_matrix)
    x = rand32(rng, mod_in, n)
    x[0] = 0
    x[-1] = 0xFFFFFFFF          # an unreduced word: the reference multiplies whatever it is given (converter.rs:160-176)
    exp = oconv.fast_convert_array(x, n)
    out = np.empty(len(mod_out) * n, np.uint32)
    conv.fast_convert_array(x, out, n)
    assert np.array_equal(out, exp)
    dout = to_dev32(np.zeros_like(out))
    conv.fast_convert_array_dev(to_dev32(x), dout, n)
    assert np.array_equal(to_host32(dout), exp)
    e = pf.BaseConverter32(pf.RNSBase32(mod_in), pf.RNSBase32(mod_out[:1]))
    oe = orc.BaseConverter32(oin, orc.RNSBase32(mod_out[:1]))
    eo = np.empty(n, np.uint32)
    e.exact_convert_array(x, eo, n)
    assert np.array_equal(eo, oe.exact_convert_array(x, n))
    deo = to_dev32(np.zeros_like(eo))
    e.exact_convert_array_dev(to_dev32(x), deo, n)
    assert np.array_equal(to_host32(deo), eo)


def test_conv32_exact_rounding_boundary(pf, orc):
    """Values around Q/2: GPU and oracle take the same branch of (sum + 0.5) as u32 for every input."""
    mod_in, p = Q30, REF_U32[0]
    Q = Q30[0] * Q30[1] * Q30[2]
    rng = np.random.default_rng(99)
    vals = [Q // 2 + d for d in range(-40, 41)] + [Q // 2 + int(rng.integers(-2 ** 30, 2 ** 30)) for _ in range(200)]
    n = len(vals)
    x = np.array([v % q for q in mod_in for v in vals], np.uint32)
    e = pf.BaseConverter32(pf.RNSBase32(mod_in), pf.RNSBase32([p]))
    oe = orc.BaseConverter32(orc.RNSBase32(mod_in), orc.RNSBase32([p]))
    eo = np.empty(n, np.uint32)
    e.exact_convert_array(x, eo, n)
    ref = oe.exact_convert_array(x, n)
    assert np.array_equal(eo, ref)
    assert {int(r) for r in ref} <= {v % p for v in vals} | {(v - Q) % p for v in vals}


def test_conv32_pairs_and_errors(pf, orc):
    rng = np.random.default_rng(17)
    n = 5000
    conv = pf.BaseConverter32(pf.RNSBase32(Q30), pf.RNSBase32(REF_U32))
    oconv = orc.BaseConverter32(orc.RNSBase32(Q30), orc.RNSBase32(REF_U32))
    x = rand32(rng, Q30, n)
    exp = oconv.fast_convert_array(x, n)
    pairs = to_dev32(np.zeros(2 * n, np.uint32))
    conv.fast_convert_array_to_pairs_dev(to_dev32(x), pairs, n)
    got = to_host32(pairs)
    assert np.array_equal(got[0::2], exp[:n]) and np.array_equal(got[1::2], exp[n:])
    three = pf.BaseConverter32(pf.RNSBase32(REF_U32), pf.RNSBase32(Q30))
    with pytest.raises(pf.PfheError) as e:
        three.fast_convert_array_to_pairs_dev(to_dev32(x[:2 * n]), pairs, n)
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:      # exact form: exactly one output modulus (converter.rs:284-288)
        conv.exact_convert_array(x, np.empty(n, np.uint32), n)
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:
        conv.fast_convert_array(x, np.empty(2 * n - 1, np.uint32), n)
    assert e.value.kind == "BadLength"
    with pytest.raises(TypeError):               # a <u64> base is not a <u32> base
        pf.BaseConverter32(pf.RNSBase(Q30), pf.RNSBase32(REF_U32))
    with pytest.raises(TypeError):
        pf.BaseConverter(pf.RNSBase32(Q30), pf.RNSBase(REF_U32))
    conv.fast_convert_array(x[:0], np.empty(0, np.uint32), 0)   # empty input


@pytest.mark.parametrize("log_n,batch", [(6, 2), (16, 4)])   # separate kernels / the fused kernels
def test_u32_product_accumulators_at_their_bound(pf, orc, log_n, batch):
    """The lazy 64-bit accumulators of the u32 multiply-accumulate kernels are folded every fifteen terms
    (csrc/pfhe_rns.hpp, kFold32Every): the bound is met exactly when every digit_hat and every key word is q - 1.  Input
    polynomials whose only coefficient is -sum_j 2^(drop + j log B) have the digit -1 at every level, so every digit
    polynomial is the constant q - 1 and so is its transform; with a key of q - 1 everywhere each of the (k+1) ell = 36
    terms adds (q - 1)^2 to every accumulator."""
    k, log_basis = 1, 5
    n, L = 1 << log_n, 3
    g = pyref.Gadget(Q30, log_basis)
    ell = g.ell
    assert (k + 1) * ell >= 2 * 15 + 1
    v = (-sum(1 << (g.drop + j * log_basis) for j in range(ell))) % g.Q
    assert g.signed_digits(v) == [-1] * ell
    poly = np.zeros(L * n, np.uint32)
    for i, q in enumerate(Q30):
        poly[i * n] = v % q
    glwe = np.tile(poly, batch * (k + 1))
    ggsw = np.concatenate([np.full(n, q - 1, np.uint32) for _ in range((k + 1) * ell * (k + 1)) for q in Q30])
    otable, obase = orc.U32DcrtTable(log_n, Q30), orc.RNSBase32(Q30)
    obasis = orc.BigUintApproxSignedBasis32(obase, log_basis)
    assert obasis.decompose_length == ell
    G = (k + 1) * L * n
    exp1 = orc.mul_dcrt32_ggsw_to(otable, obase, obasis, k, glwe[:G].copy(), ggsw)
    # every term is (q - 1)^2 = 1 (mod q) at every position: the sum is the number of terms
    assert all(int(x) == (k + 1) * ell for x in exp1)
    table, base = pf.U32DcrtTable(log_n, Q30), pf.RNSBase32(Q30)
    ctx = pf.DcrtGlevContext32(table, base, pf.BigUintApproxSignedBasis32(base, log_basis), k)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
    assert np.array_equal(out, np.tile(exp1, batch))
    # accumulating form on top of a canonical maximum: one GLev row (ell = 18 terms) added to q - 1 everywhere
    acc = to_dev32(np.concatenate([np.full(n, q - 1, np.uint32) for _ in range(batch * (k + 1)) for q in Q30]))
    glev = ggsw[:ell * (k + 1) * L * n]
    pf.add_dcrt_glev_mul_crt_poly_assign_dev(acc, to_dev32(glev), to_dev32(np.tile(poly, batch)), ctx)
    got = to_host32(acc)
    assert all(int(x) == (ell - 1) % q for x, q in zip(got[::n], Q30 * (batch * (k + 1))))
    assert np.array_equal(got, np.tile(np.concatenate([np.full(n, ell - 1, np.uint32)] * L), batch * (k + 1)))
