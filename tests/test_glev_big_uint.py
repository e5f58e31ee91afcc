"""DcrtGlwe::add_dcrt_glev_mul_big_uint_poly_assign / DcrtGlev::mul_big_uint_poly_to (primus_lattice/src/glwe/dcrt.rs:
258-338, glev/dcrt.rs:113-175): the GLev product with the polynomial given as big integers modulo Q.
CPU: the restatement equals the CRT form on the composed polynomial; -m gpu: every decomposition kernel variant of
the HIP path against the restatement."""
import os

import numpy as np
import pytest

from gpu_util import rand_rns
from pyref import Q61


def make(orc, rng, log_n, k, moduli, log_basis, batch):
    n, L = 1 << log_n, len(moduli)
    ot, ob = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(ob, log_basis)
    ell = obasis.decompose_length
    poly = rand_rns(rng, moduli, n, batch)                       # CRT polynomials
    big = np.concatenate([ob.compose_multiple_values_to(poly[e * L * n:(e + 1) * L * n], n) for e in range(batch)])
    glev = rand_rns(rng, moduli, n, ell * (k + 1))
    acc = rand_rns(rng, moduli, n, batch * (k + 1))
    return ot, ob, obasis, poly, big, glev, acc


@pytest.mark.parametrize("log_n,k,moduli,log_basis", [(3, 1, Q61, 30), (4, 2, Q61[:2], 20), (3, 1, Q61[:1], 7)])
def test_oracle_big_uint_form_equals_crt_form(orc, log_n, k, moduli, log_basis):
    rng = np.random.default_rng(log_n + k)
    n, L, batch = 1 << log_n, len(moduli), 2
    ot, ob, obasis, poly, big, glev, acc = make(orc, rng, log_n, k, moduli, log_basis, batch)
    W, vl = L * n, ob.value_len
    for e in range(batch):
        a1 = acc[e * (k + 1) * W:(e + 1) * (k + 1) * W].copy()
        a2 = a1.copy()
        orc.add_dcrt_glev_mul_crt_poly_assign(ot, ob, obasis, k, a1, glev, poly[e * W:(e + 1) * W].copy())
        orc.add_dcrt_glev_mul_big_uint_poly_assign(ot, ob, obasis, k, a2, glev, big[e * vl * n:(e + 1) * vl * n].copy())
        assert np.array_equal(a1, a2)


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("switch", [None, "PFHE_DISABLE_FUSED_EXTPROD"])
@pytest.mark.parametrize("log_n,k,moduli,log_basis,batch", [(12, 1, Q61, 30, 3), (13, 2, Q61[:2], 20, 2), (10, 1, Q61[:1], 10, 1100),
                                                            (9, 1, Q61, 30, 5), (16, 1, Q61, 30, 2)])
def test_gpu_matches_oracle(pf, orc, switch, log_n, k, moduli, log_basis, batch):
    import torch
    from gpu_util import to_dev, to_host
    rng = np.random.default_rng(log_n * 7 + k)
    n, L = 1 << log_n, len(moduli)
    ot, ob, obasis, poly, big, glev, acc = make(orc, rng, log_n, k, moduli, log_basis, batch)
    W, vl = L * n, ob.value_len
    if switch:
        os.environ[switch] = "1"  # switches are read when the table / the plan is created
    try:
        t, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
        ctx = pf.DcrtGlevContext(t, base, pf.BigUintApproxSignedBasis(base, log_basis), k)
        dacc = to_dev(acc)
        pf.add_dcrt_glev_mul_big_uint_poly_assign_dev(dacc, to_dev(glev), to_dev(big), ctx)
        dres = to_dev(acc)  # stale contents must be overwritten
        pf.glev_mul_big_uint_poly_to_dev(to_dev(glev), to_dev(big), dres, ctx)
        dcrt = to_dev(acc)
        pf.add_dcrt_glev_mul_crt_poly_assign_dev(dcrt, to_dev(glev), to_dev(poly), ctx)
    finally:
        if switch:
            del os.environ[switch]
    got, res = to_host(dacc), to_host(dres)
    assert torch.equal(dacc, dcrt)  # big-integer input == CRT input of the same polynomial, on the whole batch
    G = (k + 1) * W
    for e in sorted({0, batch // 2, batch - 1}):
        a = acc[e * G:(e + 1) * G].copy()
        orc.add_dcrt_glev_mul_big_uint_poly_assign(ot, ob, obasis, k, a, glev, big[e * vl * n:(e + 1) * vl * n].copy())
        assert np.array_equal(got[e * G:(e + 1) * G], a), e
        z = np.zeros(G, np.uint64)
        orc.add_dcrt_glev_mul_big_uint_poly_assign(ot, ob, obasis, k, z, glev, big[e * vl * n:(e + 1) * vl * n].copy())
        assert np.array_equal(res[e * G:(e + 1) * G], z), e


@pytest.mark.gpu
def test_gpu_errors(pf):
    import torch
    log_n, k = 9, 1
    t, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(t, base, basis, k)
    n, L, vl, ell = 1 << log_n, 3, base.big_uint_value_len(), basis.decompose_length()
    z = lambda w: torch.zeros(w, dtype=torch.int64, device="cuda")
    with pytest.raises(pf.PfheError) as e:  # a CRT-sized polynomial is not a whole number of big-integer polynomials...
        pf.glev_mul_big_uint_poly_to_dev(z(ell * 2 * L * n), z(vl * n + 8), z(2 * L * n), ctx)
    assert e.value.kind == "BadLength"
    with pytest.raises(pf.PfheError) as e:
        pf.add_dcrt_glev_mul_big_uint_poly_assign_dev(z(2 * L * n), z(ell * 2 * L * n - 2), z(vl * n), ctx)
    assert e.value.kind == "BadLength"
    pf.glev_mul_big_uint_poly_to_dev(z(ell * 2 * L * n), z(0), z(0), ctx)  # empty batch
