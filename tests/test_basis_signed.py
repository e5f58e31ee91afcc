"""BigUintApproxSignedBasis: the SIGNED slice decomposition (OnceBigUintSignedDecomposer::decompose_slice_to,
primus_decompose/src/big_integer/common.rs:255-306) and the out-of-place carry initialisation
(init_value_carry_slice_to, basis.rs:371-420).  CPU: restatement vs Python integers; -m gpu: HIP vs restatement."""
import numpy as np
import pytest

import pyref
from pyref import Q61, int_to_limbs, limbs_to_int

CASES = [(Q61, 30, None), (Q61, 7, None), (Q61[:2], 1, None), ([97, 101, 103], 3, None), (Q61, 20, 4),
         ([137438822401, 137438814209, 137438773249], 15, None), (Q61[:1], 20, None), (Q61, 1, None)]


def values_for(g, L, n, seed):
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(40), "little") % g.Q for _ in range(n)]
    vals[:6] = [0, 1, g.Q - 1, g.Q // 2, (g.threshold or 1) - 1, g.threshold or 1]
    return vals, np.concatenate([int_to_limbs(v, L) for v in vals])


@pytest.mark.parametrize("moduli,log_basis,rev", CASES)
def test_oracle_signed_digits_are_residues_mod_q(orc, moduli, log_basis, rev):
    base = orc.RNSBase(moduli)
    basis = orc.BigUintApproxSignedBasis(base, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    L, n = base.value_len, 129
    vals, values = values_for(g, L, n, log_basis)
    adjusted, carries = basis.init_value_carry_slice_to(values, n)
    inplace = values.copy()
    carries2 = basis.init_value_carry_slice_inplace(inplace, n)
    assert np.array_equal(adjusted, inplace) and np.array_equal(carries, carries2)
    assert np.array_equal(values, np.concatenate([int_to_limbs(v, L) for v in vals]))  # input untouched
    uc = carries.copy()
    for j in range(g.ell):
        signed = basis.decompose_slice_to(j, adjusted, carries, n)
        unsigned = basis.unsigned_decompose_slice_to(j, adjusted, uc, n)
        assert np.array_equal(carries, uc)  # the two forms propagate the same carries
        for c, v in enumerate(vals):
            d = g.signed_digits(v)[j]
            assert limbs_to_int(signed[c * L:(c + 1) * L]) == d % g.Q, (j, c)
            assert int(unsigned[c]) == d % g.B


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("moduli,log_basis,rev", CASES)
def test_gpu_matches_oracle(pf, orc, moduli, log_basis, rev):
    import torch
    from gpu_util import to_dev, to_host
    obase, base = orc.RNSBase(moduli), pf.RNSBase(moduli)
    obasis, basis = orc.BigUintApproxSignedBasis(obase, log_basis, rev), pf.BigUintApproxSignedBasis(base, log_basis, rev)
    g = pyref.Gadget(moduli, log_basis, rev)
    L, n = obase.value_len, 1031
    _, values = values_for(g, L, n, 100 + log_basis)
    exp_adj, exp_car = obasis.init_value_carry_slice_to(values, n)
    # host-pointer forms
    adj, car = np.empty_like(values), np.zeros(n, np.uint8)
    basis.init_value_carry_slice_to(values, adj, car)
    assert np.array_equal(adj, exp_adj) and np.array_equal(car, exp_car)
    # device forms
    dv, dadj = to_dev(values), torch.empty(values.size, dtype=torch.int64, device="cuda")
    dcar = torch.zeros(n, dtype=torch.uint8, device="cuda")
    basis.init_value_carry_slice_to_dev(dv, dadj, dcar)
    assert np.array_equal(to_host(dadj), exp_adj) and np.array_equal(dcar.cpu().numpy(), exp_car)
    assert np.array_equal(to_host(dv), values)
    dout = torch.empty_like(dadj)
    ocar, hcar = exp_car.copy(), exp_car.copy()
    for j in range(g.ell):
        exp = obasis.decompose_slice_to(j, exp_adj, ocar, n)
        basis.decompose_slice_to_dev(j, dadj, dout, dcar)
        assert np.array_equal(to_host(dout), exp), j
        assert np.array_equal(dcar.cpu().numpy(), ocar)
        hout = np.empty_like(values)
        basis.decompose_slice_to(j, exp_adj, hout, hcar)
        assert np.array_equal(hout, exp) and np.array_equal(hcar, ocar)


@pytest.mark.gpu
def test_gpu_errors(pf):
    base = pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    L = base.big_uint_value_len()
    v, out, car = np.zeros(4 * L, np.uint64), np.zeros(4 * L, np.uint64), np.zeros(4, np.uint8)
    with pytest.raises(pf.PfheError) as e:
        basis.decompose_slice_to(basis.decompose_length(), v, out, car)  # no such level
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:
        basis.decompose_slice_to(0, v, out[:L], car)
    assert e.value.kind == "BadLength"
    with pytest.raises(pf.PfheError) as e:
        basis.init_value_carry_slice_to(v, out[:L], car)
    assert e.value.kind == "BadLength"
