"""GPU parity of RNSBase / BigUintApproxSignedBasis / external product against the oracle.

Mirrors primus_rns/tests/rns.rs and primus_decompose/tests/big_uint.rs through the C ABI, and
adds the end-to-end external product the reference never tests.
"""
import numpy as np
import pytest

import pyref
from gpu_util import rand_rns, to_dev, to_host
from pyref import Q61, crt_compose, int_to_limbs, limbs_to_int

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def test_rns_errors(pf):
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase([])
    assert e.value.kind == "EmptyBase"
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase([21, 35])
    assert e.value.kind == "CoPrimeError"
    with pytest.raises(pf.PfheError) as e:
        pf.RNSBase([1 << 62, 97])
    assert e.value.kind == "UnrepresentableModulus"


def test_rns_reference_closed_forms(pf):
    """rns.rs:77-98 ((3,5,7), residues (2,3,2) -> 23) and rns.rs:105-147 (modulus-major layout)."""
    base = pf.RNSBase([3, 5, 7])
    out = np.empty(1, np.uint64)
    base.compose_multiple_values_to(np.array([2, 3, 2], np.uint64), out, 1)
    assert int(out[0]) == 23
    moduli = [1_125_899_906_826_241, 1_125_899_906_629_633]
    base = pf.RNSBase(moduli)
    rows = [[0, 0], [1, 2], [97, 131], [moduli[0] - 1, moduli[1] - 2], [123_456_789, 987_654_321]]
    packed = np.array([r[i] for i in range(2) for r in rows], np.uint64)
    vals = np.empty(len(rows) * base.big_uint_value_len(), np.uint64)
    base.compose_multiple_values_to(packed, vals, len(rows))
    L = base.big_uint_value_len()
    for c, r in enumerate(rows):
        assert limbs_to_int(vals[c * L:(c + 1) * L]) == crt_compose(r, moduli)


def test_wrapping_decompose_reference_case(pf):
    """rns.rs:154-194."""
    moduli = [97, 101, 103]
    base = pf.RNSBase(moduli)
    for sm in (2, 7, 16):
        small = np.array([(i * 5 + 3) % sm for i in range(17)], np.uint64)
        got = np.empty(3 * 17, np.uint64)
        base.wrapping_decompose_small_values_to(small, got, 17, sm)
        exp = [v if (sm == 2 or v < -(-sm // 2)) else m - sm + v for m in moduli for v in map(int, small)]
        assert got.tolist() == exp


@pytest.mark.parametrize("moduli", [Q61, [137438822401, 137438814209, 137438773249], [Q61[0]], Q61[:2]])
@pytest.mark.parametrize("count", [1, 64, 1000])
def test_compose_matches_oracle(pf, orc, moduli, count):
    rng = np.random.default_rng(count)
    base, obase = pf.RNSBase(moduli), orc.RNSBase(moduli)
    assert base.big_uint_value_len() == obase.value_len
    assert np.array_equal(base.moduli_product(), obase.moduli_product)
    res = np.concatenate([rng.integers(0, m, count, dtype=np.uint64) for m in moduli])
    res[0] = 0
    for i, m in enumerate(moduli):
        res[i * count + count - 1] = m - 1
    out = np.empty(count * base.big_uint_value_len(), np.uint64)
    base.compose_multiple_values_to(res, out, count)
    assert np.array_equal(out, obase.compose_multiple_values_to(res, count))
    with pytest.raises(pf.PfheError) as e:
        base.compose_multiple_values_to(res[:-1].copy(), out, count)
    assert e.value.kind == "BadLength"


@pytest.mark.parametrize("moduli", [
    [134215681, 134176769],                                # L = 2, one limb
    Q61[:2],                                               # L = 2, two limbs
    [137438822401, 137438814209, 137438773249],            # L = 3, two limbs
    Q61,                                                   # L = 3, three limbs
    [2305843009213317121, 1152921504606584833, 2305843009211596801],  # 61 / 60 / 61 bits: max > 2 min, general form
    [1125899906826241, 2305843009211596801],               # 50 / 61 bits: the general form only
    [3, 5, 7], [2, 3],
])
def test_compose_mixed_radix_form_equals_general_form(pf, moduli, monkeypatch):
    """The mixed-radix (Garner) lift that bases of 2-3 similar moduli take against the general form (switch read when the
    base is created) and against Python big integers, incl. the extreme residues."""
    count = 4096
    rng = np.random.default_rng(len(moduli) * 1000 + moduli[0] % 997)
    res = np.concatenate([rng.integers(0, m, count, dtype=np.uint64) for m in moduli])
    for i, m in enumerate(moduli):  # all-zero, all-maximal and mixed extreme columns
        res[i * count + 0] = 0
        res[i * count + 1] = m - 1
        res[i * count + 2] = (m - 1) if i % 2 else 0
        res[i * count + 3] = 0 if i % 2 else (m - 1)
    base = pf.RNSBase(moduli)
    monkeypatch.setenv("PFHE_DISABLE_GARNER", "1")
    general = pf.RNSBase(moduli)
    monkeypatch.delenv("PFHE_DISABLE_GARNER")
    L = base.big_uint_value_len()
    a, b = np.empty(count * L, np.uint64), np.empty(count * L, np.uint64)
    base.compose_multiple_values_to(res, a, count)
    general.compose_multiple_values_to(res, b, count)
    assert np.array_equal(a, b)
    for c in list(range(8)) + [count - 1]:
        assert limbs_to_int(a[c * L:(c + 1) * L]) == crt_compose([int(res[i * count + c]) for i in range(len(moduli))], moduli)


@pytest.mark.parametrize("moduli,log_basis,rev", [
    (Q61, 30, None), (Q61, 30, 4), (Q61, 61, None), (Q61, 1, None), (Q61, 7, None),
    ([134215681, 134176769], 7, None), ([134215681, 134176769], 6, None), (Q61[:2], 13, 5),
    ([137438822401, 137438814209, 137438773249], 15, None), (Q61[:1], 20, None),
])
def test_gadget_steps_match_oracle(pf, orc, moduli, log_basis, rev):
    """Steps (2)-(4) of glwe/dcrt.rs:226-244, slice by slice, + the recomposition bound."""
    rng = np.random.default_rng(log_basis)
    base, obase = pf.RNSBase(moduli), orc.RNSBase(moduli)
    basis, obasis = pf.BigUintApproxSignedBasis(base, log_basis, rev), orc.BigUintApproxSignedBasis(obase, log_basis, rev)
    assert (basis.decompose_length(), basis.log_basis(), basis.drop_bits(), basis.basis_value()) == \
        (obasis.decompose_length, obasis.log_basis, obasis.drop_bits, obasis.basis_value)
    assert np.array_equal(basis.scalars(), obasis.scalars)
    assert np.array_equal(basis.scalars_residue(), obasis.scalars_residue)
    g = pyref.Gadget(moduli, log_basis, rev)
    L, n = base.big_uint_value_len(), 515
    vals_int = [int.from_bytes(rng.bytes(40), "little") % g.Q for _ in range(n)]
    vals_int[:6] = [0, 1, g.Q - 1, g.Q // 2, (g.threshold or 1) - 1, g.threshold or 1]
    values = np.concatenate([int_to_limbs(v, L) for v in vals_int])
    ov = values.copy()
    oc = obasis.init_value_carry_slice_inplace(ov, n)
    gv, gc = values.copy(), np.zeros(n, np.uint8)
    basis.init_value_carry_slice_inplace(gv, gc)
    assert np.array_equal(gv, ov) and np.array_equal(gc, oc)
    for j in range(basis.decompose_length()):
        od = obasis.unsigned_decompose_slice_to(j, ov, oc, n)
        gd = np.empty(n, np.uint64)
        basis.unsigned_decompose_slice_to(j, gv, gd, gc)
        assert np.array_equal(gd, od) and np.array_equal(gc, oc), j
        if basis.basis_value() < min(moduli):  # the centred lift needs B < q_i (base.rs:288-292)
            lifted = np.empty(len(moduli) * n, np.uint64)
            base.wrapping_decompose_small_values_to(gd, lifted, n, basis.basis_value())
            assert np.array_equal(lifted, obase.wrapping_decompose_small_values_to(od, obasis.basis_value))


def test_basis_errors(pf):
    base = pf.RNSBase(Q61)
    for lb in (0, 64):  # basis.rs:51
        with pytest.raises(pf.PfheError) as e:
            pf.BigUintApproxSignedBasis(base, lb)
        assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError):
        pf.BigUintApproxSignedBasis(base, 30, 7)  # reverse_length > full length (basis.rs:64)
    # B = 2^61 exceeds the 61-bit moduli: fine for the basis itself, rejected where the centred
    # lift would need B < q_i (primus_rns/src/base.rs:288-292)
    big = pf.BigUintApproxSignedBasis(base, 61)
    with pytest.raises(pf.PfheError) as e:
        pf.DcrtGlevContext(pf.U64DcrtTable(4, Q61), base, big, 1)
    assert e.value.kind == "BadArgument"


def make_case(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared):
    n, Lm = 1 << log_n, len(moduli)
    otable, obase = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(obase, log_basis, rev)
    ell = obasis.decompose_length
    glwe = rand_rns(rng, moduli, n, batch * (k + 1))
    nkeys = 1 if shared else batch
    ggsw = rand_rns(rng, moduli, n, nkeys * (k + 1) * ell * (k + 1))  # already "in NTT form": any residues
    W, glwe_len, ggsw_len = Lm * n, (k + 1) * Lm * n, (k + 1) * ell * (k + 1) * Lm * n
    exp = np.concatenate([
        orc.mul_dcrt_ggsw_to(otable, obase, obasis, k, glwe[e * glwe_len:(e + 1) * glwe_len].copy(),
                             ggsw[(0 if shared else e) * ggsw_len:((0 if shared else e) + 1) * ggsw_len].copy())
        for e in range(batch)])
    return otable, glwe, ggsw, exp


@pytest.mark.parametrize("log_n,k,moduli,log_basis,rev,batch,shared,chunk", [
    (3, 1, Q61, 30, None, 1, True, 0), (4, 1, Q61, 30, None, 5, True, 2), (4, 1, Q61, 30, None, 5, False, 3),
    (6, 2, Q61[:2], 20, 3, 3, True, 1), (10, 1, [134215681, 134176769], 7, None, 2, False, 0),
    (12, 1, Q61, 30, None, 3, True, 2), (13, 1, Q61, 13, None, 1, True, 0),
    # digits wider than 32 bits on two-pass rings: the int64 instantiations of the digit kernels (log B = 40: ell = 4,
    # 45: ell = 4, 61: ell = 3), and on a single-pass ring (unfused kernels)
    (15, 1, Q61, 40, None, 2, True, 0), (16, 1, Q61, 45, None, 2, False, 1), (16, 1, Q61, 60, None, 1, True, 0),
    (12, 1, Q61, 33, None, 2, True, 0),
])
def test_external_product_matches_oracle(pf, orc, log_n, k, moduli, log_basis, rev, batch, shared, chunk):
    """CrtGlwe::mul_dcrt_ggsw_to (glwe/crt.rs:200-227), batched, shared or per-ciphertext GGSW."""
    rng = np.random.default_rng(log_n * 7 + batch)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared)
    table, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    basis = pf.BigUintApproxSignedBasis(base, log_basis, rev)
    ctx = pf.DcrtGlevContext(table, base, basis, k, chunk)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
    assert np.array_equal(out, exp)
    # coefficient-form output == oracle result after DcrtGlwe::into_coeff_form
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx, into_coeff_form=True)
    otable.inverse_transform_slice(exp)
    assert np.array_equal(out, exp)
    with pytest.raises(pf.PfheError) as e:
        pf.mul_dcrt_ggsw_to(glwe, ggsw[:-1].copy(), out, ctx)
    assert e.value.kind == "BadLength"


@pytest.mark.parametrize("log_basis", [31, 32, 50])
def test_wide_digits_generic_primes_and_unfused_path(pf, orc, log_basis, monkeypatch):
    """The digit width is a template parameter of gadget_signed_digits_kernel / digits_strided_kernel (int32 up to
    log B = 31, int64 beyond: big_integer/common.rs:132-140 allows any log B < 64).  Generic-prime (Shoup / Montgomery)
    tables, with the fused multiply-accumulate kernel and with the separate transform + multiply-accumulate kernels
    (PFHE_DISABLE_FUSED_EXTPROD, read when the plan is created), must give the oracle's words."""
    log_n, k, batch = 16, 1, 2
    rng = np.random.default_rng(log_basis)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, Q61, log_basis, None, batch, True)
    monkeypatch.setenv("PFHE_DISABLE_PM", "1")
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    monkeypatch.delenv("PFHE_DISABLE_PM")
    basis = pf.BigUintApproxSignedBasis(base, log_basis)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    monkeypatch.setenv("PFHE_DISABLE_FUSED_EXTPROD", "1")
    ctx_unfused = pf.DcrtGlevContext(pf.U64DcrtTable(log_n, Q61), base, basis, k)
    monkeypatch.delenv("PFHE_DISABLE_FUSED_EXTPROD")
    out = np.empty_like(glwe)
    for c in (ctx, ctx_unfused):
        pf.mul_dcrt_ggsw_to(glwe, ggsw, out, c)
        assert np.array_equal(out, exp)
    e2 = exp.copy()
    otable.inverse_transform_slice(e2)
    for c in (ctx, ctx_unfused):
        pf.mul_dcrt_ggsw_to(glwe, ggsw, out, c, into_coeff_form=True)
        assert np.array_equal(out, e2)


def test_external_product_equals_schoolbook(pf, orc):
    """End-to-end ground truth on Python integers: sum_i sum_j digit_ij (*) key_ij mod (X^N+1, q_r)."""
    log_n, k, moduli, log_basis = 3, 1, Q61, 30
    rng = np.random.default_rng(42)
    n, Lm = 1 << log_n, 3
    table, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    basis = pf.BigUintApproxSignedBasis(base, log_basis)
    g = pyref.Gadget(moduli, log_basis)
    ell = g.ell
    glwe = rand_rns(rng, moduli, n, k + 1)
    key_coeff = rand_rns(rng, moduli, n, (k + 1) * ell * (k + 1))
    ggsw = key_coeff.copy()
    table.transform_slice(ggsw)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx, into_coeff_form=True)
    exp = pyref.external_product_coeff(moduli, n, k, g, glwe.reshape(k + 1, Lm, n).tolist(),
                                       key_coeff.reshape(k + 1, ell, k + 1, Lm, n).tolist())
    assert out.reshape(k + 1, Lm, n).tolist() == exp


def test_add_dcrt_glev_mul_crt_poly_assign(pf, orc):
    """glwe/dcrt.rs:178-255: accumulate one GLev row into an existing DcrtGlwe."""
    log_n, k, moduli, log_basis, batch = 8, 1, Q61, 30, 3
    rng = np.random.default_rng(8)
    n, Lm = 1 << log_n, 3
    otable, obase = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(obase, log_basis)
    ell, W = obasis.decompose_length, Lm * n
    acc = rand_rns(rng, moduli, n, batch * (k + 1))
    glev = rand_rns(rng, moduli, n, ell * (k + 1))
    poly = rand_rns(rng, moduli, n, batch)
    exp = acc.copy()
    for e in range(batch):
        a = exp[e * (k + 1) * W:(e + 1) * (k + 1) * W]
        orc.add_dcrt_glev_mul_crt_poly_assign(otable, obase, obasis, k, a, glev, poly[e * W:(e + 1) * W].copy())
    table, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    ctx = pf.DcrtGlevContext(table, base, pf.BigUintApproxSignedBasis(base, log_basis), k)
    dacc = to_dev(acc)
    pf.add_dcrt_glev_mul_crt_poly_assign_dev(dacc, to_dev(glev), to_dev(poly), ctx)
    assert np.array_equal(to_host(dacc), exp)
    # DcrtGlev::mul_crt_poly_to (glev/dcrt.rs:45-110) overwrites: equals the accumulation into zeros
    exp0 = np.zeros_like(acc)
    for e in range(batch):
        a = exp0[e * (k + 1) * W:(e + 1) * (k + 1) * W]
        orc.add_dcrt_glev_mul_crt_poly_assign(otable, obase, obasis, k, a, glev, poly[e * W:(e + 1) * W].copy())
    dres = to_dev(acc)  # stale contents must be overwritten
    pf.glev_mul_crt_poly_to_dev(to_dev(glev), to_dev(poly), dres, ctx)
    assert np.array_equal(to_host(dres), exp0)


def test_config4_shape_vs_oracle(pf, orc):
    """BASELINE config 4 shape (N=2^16, 3 primes, k=1, logB=30 -> ell=6), small batch, shared GGSW."""
    log_n, k, batch = 16, 1, 2
    rng = np.random.default_rng(4)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, Q61, 30, None, batch, True)
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    assert (basis.decompose_length(), basis.drop_bits()) == (6, 3)
    ctx = pf.DcrtGlevContext(table, base, basis, k, 1)
    dout = to_dev(np.zeros_like(glwe))
    pf.mul_dcrt_ggsw_to_dev(to_dev(glwe), to_dev(ggsw), dout, ctx)
    assert np.array_equal(to_host(dout), exp)


def test_config4_full_batch_properties(pf, orc):
    """BASELINE config 4 at its full batch (1024 ciphertexts, one shared 36 MiB GGSW, chunked and
    pipelined over two streams): oracle spot checks, batch-independence (a ciphertext's result does not
    depend on its neighbours), zero-in => zero-out (tfhe_external_product.rs:134 analogue) and
    coefficient-form output == inverse transform of the NTT-form output."""
    import ctypes as C
    import torch
    from primus_fhe_amd._lib import check, u64p
    log_n, k, batch = 16, 1, 1024
    n, L = 1 << log_n, 3
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    glwe_len, ggsw_len = ctx.glwe_len(), ctx.ggsw_len()
    mods = np.array(Q61, np.uint64)

    def fill(words, seed):
        x = torch.empty(words, dtype=torch.int64, device="cuda")
        check(pf.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, seed, None))
        return x

    glwe, ggsw = fill(batch * glwe_len, 41), fill(ggsw_len, 42)
    glwe[5 * glwe_len:6 * glwe_len] = 0  # one all-zero ciphertext
    out = torch.empty_like(glwe)
    pf.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx)
    torch.cuda.synchronize()
    otable, obase = orc.U64DcrtTable(log_n, Q61), orc.RNSBase(Q61)
    obasis = orc.BigUintApproxSignedBasis(obase, 30)
    hk = to_host(ggsw)
    for e in (0, 31, 32, 777, batch - 1):  # chunk boundaries included (chunk = 32)
        exp = orc.mul_dcrt_ggsw_to(otable, obase, obasis, k, to_host(glwe[e * glwe_len:(e + 1) * glwe_len]).copy(), hk)
        assert np.array_equal(to_host(out[e * glwe_len:(e + 1) * glwe_len]), exp), e
    assert int(out[5 * glwe_len:6 * glwe_len].abs().max()) == 0
    # batch independence: the same ciphertext alone gives the same result
    single = torch.empty(glwe_len, dtype=torch.int64, device="cuda")
    pf.mul_dcrt_ggsw_to_dev(glwe[777 * glwe_len:778 * glwe_len].clone(), ggsw, single, ctx)
    assert torch.equal(single, out[777 * glwe_len:778 * glwe_len])
    # coefficient form == inverse transform of the NTT form
    coeff = torch.empty_like(glwe)
    pf.mul_dcrt_ggsw_to_dev(glwe, ggsw, coeff, ctx, into_coeff_form=True)
    table.inverse_transform_dev(out)
    assert torch.equal(coeff, out)


@pytest.mark.parametrize("log_n,moduli,log_basis", [(10, Q61[:1], 10), (11, Q61[:2], 20), (10, [1125899906826241, 562949953392641], 13)])
def test_small_ring_fused_path(pf, orc, log_n, moduli, log_basis):
    """N = 2^10 / 2^11, k = 1, batches that fill the chip take the single-kernel path (digits -> transforms ->
    multiply-accumulate -> inverse transforms on chip): equal to the separate kernels on the whole batch and to
    the oracle on sampled ciphertexts, for NTT-form and coefficient-form output and for the accumulating row API."""
    import os
    import torch
    k, batch = 1, 1100
    n, L = 1 << log_n, len(moduli)
    W = (k + 1) * L * n
    rng = np.random.default_rng(log_n + L)
    ot, ob = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(ob, log_basis)
    ell = obasis.decompose_length
    t, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    ctx = pf.DcrtGlevContext(t, base, pf.BigUintApproxSignedBasis(base, log_basis), k)
    os.environ["PFHE_DISABLE_FUSED_EXTPROD"] = "1"  # read when a plan is created: the separate kernels for every shape
    try:
        ctx_plain = pf.DcrtGlevContext(t, base, pf.BigUintApproxSignedBasis(base, log_basis), k)
    finally:
        del os.environ["PFHE_DISABLE_FUSED_EXTPROD"]
    glwe = rand_rns(rng, moduli, n, batch * (k + 1))
    ggsw = rand_rns(rng, moduli, n, (k + 1) * ell * (k + 1))
    dg, dk = to_dev(glwe), to_dev(ggsw)
    outs = {}
    for coeff in (False, True):
        fused = torch.zeros_like(dg)
        pf.mul_dcrt_ggsw_to_dev(dg, dk, fused, ctx, into_coeff_form=coeff)
        plain = torch.zeros_like(dg)
        pf.mul_dcrt_ggsw_to_dev(dg, dk, plain, ctx_plain, into_coeff_form=coeff)
        assert torch.equal(fused, plain)
        outs[coeff] = to_host(fused)
    for e in (0, 517, batch - 1):
        exp = orc.mul_dcrt_ggsw_to(ot, ob, obasis, k, glwe[e * W:(e + 1) * W].copy(), ggsw)
        assert np.array_equal(outs[False][e * W:(e + 1) * W], exp)
        ot.inverse_transform_slice(exp)
        assert np.array_equal(outs[True][e * W:(e + 1) * W], exp)
    # accumulating single-row API (glwe/dcrt.rs:178-255) through the same kernel
    acc = rand_rns(rng, moduli, n, batch * (k + 1))
    glev = rand_rns(rng, moduli, n, ell * (k + 1))
    poly = rand_rns(rng, moduli, n, batch)
    dacc = to_dev(acc)
    pf.add_dcrt_glev_mul_crt_poly_assign_dev(dacc, to_dev(glev), to_dev(poly), ctx)
    got = to_host(dacc)
    for e in (0, 733):
        a = acc[e * W:(e + 1) * W].copy()
        orc.add_dcrt_glev_mul_crt_poly_assign(ot, ob, obasis, k, a, glev, poly[e * L * n:(e + 1) * L * n].copy())
        assert np.array_equal(got[e * W:(e + 1) * W], a)


def test_chunked_batch_equals_one_chunk(pf):
    """Chunks of a batch run one after the other on the caller's stream: a plan with chunks of 2 ciphertexts (4 chunks, the
    last one short) gives the words of a plan that holds the whole batch, for the fused (N = 2^12) and the unfused
    (N = 2^9) kernels, with one GGSW per ciphertext."""
    import torch
    for log_n in (12, 9):
        k, batch, n = 1, 7, 1 << log_n
        rng = np.random.default_rng(log_n)
        t, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
        basis = pf.BigUintApproxSignedBasis(base, 30)
        ell = basis.decompose_length()
        dg = to_dev(rand_rns(rng, Q61, n, batch * (k + 1)))
        dk = to_dev(rand_rns(rng, Q61, n, batch * (k + 1) * ell * (k + 1)))  # one GGSW per ciphertext
        outs = []
        for chunk in (2, 16):
            ctx = pf.DcrtGlevContext(t, base, basis, k, chunk)
            out = torch.zeros_like(dg)
            pf.mul_dcrt_ggsw_to_dev(dg, dk, out, ctx, into_coeff_form=True)
            outs.append(out)
        assert torch.equal(outs[0], outs[1])


def test_config4_full_batch_every_ciphertext(pf, orc):
    """BASELINE config 4 at its full batch: every one of the 1024 external products compared with the oracle
    (scalar restatement of CrtGlwe::mul_dcrt_ggsw_to, one ciphertext per task on all host cores)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from gpu_util import usable_cores
    from primus_fhe_amd._lib import check, u64p
    log_n, k, batch = 16, 1, 1024
    n, L = 1 << log_n, 3
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    G = ctx.glwe_len()
    mods = np.array(Q61, np.uint64)

    def fill(words, seed):
        x = torch.empty(words, dtype=torch.int64, device="cuda")
        check(pf.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, seed, None))
        return x

    glwe, ggsw = fill(batch * G, 0x5EED000000000004), fill(ctx.ggsw_len(), 99)
    out = torch.empty_like(glwe)
    pf.mul_dcrt_ggsw_to_dev(glwe, ggsw, out, ctx)
    torch.cuda.synchronize()
    otable, obase = orc.U64DcrtTable(log_n, Q61), orc.RNSBase(Q61)
    obasis = orc.BigUintApproxSignedBasis(obase, 30)
    hk, hg, ho = to_host(ggsw), to_host(glwe), to_host(out)

    def one(e):
        exp = orc.mul_dcrt_ggsw_to(otable, obase, obasis, k, hg[e * G:(e + 1) * G], hk)
        return bool(np.array_equal(ho[e * G:(e + 1) * G], exp))

    with ThreadPoolExecutor(usable_cores()) as ex:
        ok = list(ex.map(one, range(batch)))
    assert all(ok), [e for e, v in enumerate(ok) if not v][:10]


@pytest.mark.parametrize("log_n,batch,shared,generic", [
    (13, 48, True, False), (15, 12, False, False), (16, 8, True, False), (16, 7, False, True), (17, 3, True, False),
])
def test_fused_inverse_tail_equals_separate_inverse(pf, orc, log_n, batch, shared, generic, monkeypatch):
    """Coefficient-form output of the fused path (two-pass rings, k = 1): the multiply-accumulate kernel runs the inverse
    transform's block pass on its accumulators and a strided pass finishes (DcrtGlwe::into_coeff_form,
    macros/mod.rs:901-911).  Same words as the NTT-form product followed by the table's inverse transform, and as the
    oracle; pseudo-Mersenne and generic-prime arithmetic, shared and per-ciphertext keys."""
    import torch
    k = 1
    rng = np.random.default_rng(1000 + log_n + batch)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, Q61, 30, None, batch if log_n <= 13 else 2, shared)
    n = 1 << log_n
    if generic:
        monkeypatch.setenv("PFHE_DISABLE_PM", "1")
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    monkeypatch.delenv("PFHE_DISABLE_PM", raising=False)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ell = basis.decompose_length()
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    G, K = ctx.glwe_len(), ctx.ggsw_len()
    # the first ciphertexts are the oracle's case; the rest of the batch (enough to take the fused kernels) is random
    oc = glwe.size // G
    more = batch - oc
    full_g = np.concatenate([glwe, rand_rns(rng, Q61, n, more * (k + 1))]) if more else glwe
    full_k = ggsw if shared or not more else np.concatenate([ggsw, rand_rns(rng, Q61, n, more * (k + 1) * ell * (k + 1))])
    dg, dk = to_dev(full_g), to_dev(full_k)
    fused, sep = torch.zeros_like(dg), torch.zeros_like(dg)
    pf.mul_dcrt_ggsw_to_dev(dg, dk, fused, ctx, into_coeff_form=True)
    pf.mul_dcrt_ggsw_to_dev(dg, dk, sep, ctx)
    table.inverse_transform_dev(sep)
    assert torch.equal(fused, sep)
    otable.inverse_transform_slice(exp)
    assert np.array_equal(to_host(fused[:oc * G]), exp)


@pytest.mark.parametrize("batch,log_n", [(1, 16), (2, 16), (4, 16), (12, 16), (3, 13), (40, 13)])
def test_fused_kernel_threshold(pf, orc, batch, log_n):
    """The fused block pass + multiply-accumulate kernel is taken once a call offers 160 workgroups (N = 2^16, 3 limbs: from
    4 ciphertexts; N = 2^13: from 27), the separate kernels below: the oracle's words on either side of the threshold, NTT
    form and coefficient form."""
    k = 1
    rng = np.random.default_rng(9090 + log_n + batch)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, Q61, 30, None, 1, True)
    n = 1 << log_n
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(table, base, basis, k)
    G = ctx.glwe_len()
    full = np.concatenate([glwe, rand_rns(rng, Q61, n, (batch - 1) * (k + 1))]) if batch > 1 else glwe
    out = np.empty_like(full)
    pf.mul_dcrt_ggsw_to(full, ggsw, out, ctx)
    assert np.array_equal(out[:G], exp)
    pf.mul_dcrt_ggsw_to(full, ggsw, out, ctx, into_coeff_form=True)
    otable.inverse_transform_slice(exp)
    assert np.array_equal(out[:G], exp)


def test_plan_has_one_holder_at_a_time(pf, orc):
    """`&mut DcrtGlevContext` (primus_lattice/src/context/glev.rs:4-10): the reference's borrow checker lets one caller hold
    the product's scratch.  Here a second THREAD that calls into a plan while another holds it gets PFHE_ERR_BUSY ("plan in
    use") instead of racing on the digit buffers.  Deterministic: thread A takes the plan exactly as an entry point does
    (pfhe_extprod_plan_debug_hold) and keeps it while thread B calls — one refusal of the right kind, nothing written —
    then releases it, and B's next call succeeds bit-exactly.  The same thread may nest (the host-pointer entry calls the
    device one), and while a thread is INSIDE a real call a watcher sees the flag."""
    import threading
    log_n, k, batch = 14, 1, 12
    rng = np.random.default_rng(77)
    otable, glwe, ggsw, exp = make_case(orc, rng, log_n, k, Q61, 30, None, batch, True)
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(table, base, basis, k, 2)       # 6 chunks: a long call
    lib = pf.lib()
    assert lib.pfhe_extprod_plan_in_use(ctx._h) == 0
    held, release, results = threading.Event(), threading.Event(), {}

    def a():
        results["hold"] = lib.pfhe_extprod_plan_debug_hold(ctx._h, 1)
        out_a = np.empty_like(glwe)
        pf.mul_dcrt_ggsw_to(glwe, ggsw, out_a, ctx)          # the holder itself may call (nested entry)
        results["a_ok"] = bool(np.array_equal(out_a, exp))
        held.set()
        release.wait(60)
        results["release"] = lib.pfhe_extprod_plan_debug_hold(ctx._h, 0)

    ta = threading.Thread(target=a)
    ta.start()
    assert held.wait(60)
    assert lib.pfhe_extprod_plan_in_use(ctx._h) == 1
    out_b = np.full_like(glwe, 7)
    with pytest.raises(pf.PfheError) as e:                   # this (main) thread is the second caller
        pf.mul_dcrt_ggsw_to(glwe, ggsw, out_b, ctx)
    assert e.value.kind == "Busy" and "in use" in str(e.value)
    assert (out_b == 7).all()                                # refused before anything ran
    assert lib.pfhe_extprod_plan_debug_hold(ctx._h, 1) == 38  # PFHE_ERR_BUSY for the hook too
    assert lib.pfhe_extprod_plan_debug_hold(ctx._h, 0) == 33  # releasing what one does not hold: BAD_ARGUMENT
    release.set()
    ta.join()
    assert results == {"hold": 0, "a_ok": True, "release": 0}
    assert lib.pfhe_extprod_plan_in_use(ctx._h) == 0
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out_b, ctx)
    assert np.array_equal(out_b, exp)
    # while this thread is INSIDE a call the flag is up (seen from a watcher), and it is down after
    seen = []
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            if lib.pfhe_extprod_plan_in_use(ctx._h):
                seen.append(1)
                return

    tw = threading.Thread(target=watch)
    tw.start()
    for _ in range(20):
        pf.mul_dcrt_ggsw_to(glwe, ggsw, out_b, ctx)
        if seen:
            break
    stop.set()
    tw.join()
    assert seen, "the watcher never saw the holder flag during 20 host-pointer products"
    assert lib.pfhe_extprod_plan_in_use(ctx._h) == 0


def test_plan_used_from_one_stream_after_another_is_ordered_by_the_library(pf, orc):
    """A plan's digit buffers are shared by all its calls.  Two products queued back to back on two DIFFERENT streams (no
    event handling by the caller, no synchronisation in between): the library makes the second stream wait for the first
    call's kernels (run_product, csrc/pfhe_capi_rns.hip).  Both results must be the oracle's — a long first call (chunks of
    2 ciphertexts, 24 of them) that is still running when the second is queued."""
    import torch
    log_n, k = 14, 1
    rng = np.random.default_rng(78)
    otable, g1, ggsw, e1 = make_case(orc, rng, log_n, k, Q61, 30, None, 24, True)
    _, g2, ggsw2, e2 = make_case(orc, rng, log_n, k, Q61, 30, None, 3, True)
    table, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    ctx = pf.DcrtGlevContext(table, base, pf.BigUintApproxSignedBasis(base, 30), k, 2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d1, dk1, d2, dk2 = to_dev(g1), to_dev(ggsw), to_dev(g2), to_dev(ggsw2)
    o1, o2 = torch.empty_like(d1), torch.empty_like(d2)
    torch.cuda.synchronize()
    for _ in range(5):
        pf.mul_dcrt_ggsw_to_dev(d1, dk1, o1, ctx, stream=s1.cuda_stream)
        pf.mul_dcrt_ggsw_to_dev(d2, dk2, o2, ctx, stream=s2.cuda_stream)   # other stream, same plan, nothing in between
        pf.mul_dcrt_ggsw_to_dev(d1, dk1, o1, ctx, stream=s1.cuda_stream)   # and back
        s1.synchronize(); s2.synchronize()
        assert np.array_equal(to_host(o1), e1)
        assert np.array_equal(to_host(o2), e2)
