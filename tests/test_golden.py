"""Committed golden fixtures (tests/golden/*.json, made by tests/golden/make_golden.py).

CPU half: the oracle reproduces every fixture.  GPU half (-m gpu): the HIP path, called through the
C ABI, reproduces the same fixtures -- full vectors at N <= 64, SHA-256 digests at N = 2^10..2^16.
"""
import json
import os

import numpy as np
import pytest

import pyref
from golden_inputs import digest, splitmix_rns, splitmix_uniform

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


NTT_SMALL = load("ntt_small.json")
RNS_SMALL = load("rns_gadget_small.json")
EXT_SMALL = load("extprod_small.json")
DIGESTS = load("digests.json")


def u64(strs):
    return np.array([int(s) for s in strs], np.uint64)


def test_input_stream_is_stable():
    """The SplitMix64 inputs the fixtures were generated from are what the tests regenerate."""
    for c in NTT_SMALL:
        assert splitmix_uniform(c["seed_a"], c["q"], 1 << c["log_n"]).tolist() == [int(v) for v in c["a"]]
    for d in DIGESTS:
        if d["kind"] == "ntt_forward":
            a = splitmix_uniform(d["seed"], int(d["q"]), d["batch"] << d["log_n"])
            assert digest(a) == d["input_sha256"]


# --------------------------------------------------------------------------- oracle vs fixtures

@pytest.mark.parametrize("c", NTT_SMALL, ids=lambda c: f"n{1 << c['log_n']}-q{c['q']}")
def test_oracle_ntt_small(orc, c):
    t = orc.U64NttTable(c["log_n"], c["q"])
    assert t.root == c["root"]
    x = u64(c["a"])
    t.transform_slice(x)
    assert x.tolist() == [int(v) for v in c["ntt_a"]]
    y = u64(c["b"])
    t.transform_slice(y)
    z = np.array([int(p) * int(r) % c["q"] for p, r in zip(x, y)], np.uint64)
    t.inverse_transform_slice(z)
    assert z.tolist() == [int(v) for v in c["a_mul_b"]]


@pytest.mark.parametrize("c", RNS_SMALL, ids=lambda c: f"case{c['case']}")
def test_oracle_rns_gadget_small(orc, c):
    moduli = [int(m) for m in c["moduli"]]
    count = c["count"]
    base = orc.RNSBase(moduli)
    vals = base.compose_multiple_values_to(splitmix_rns(c["seed"], moduli, count), count)
    W = base.value_len
    assert [hex(pyref.limbs_to_int(vals[i * W:(i + 1) * W])) for i in range(count)] == c["values"]
    basis = orc.BigUintApproxSignedBasis(base, c["log_basis"], c["reverse_length"])
    assert (basis.decompose_length, basis.drop_bits) == (c["decompose_length"], c["drop_bits"])
    carries = basis.init_value_carry_slice_inplace(vals, count)
    B, half = 1 << c["log_basis"], ((1 << c["log_basis"]) + 1) // 2
    for j in range(basis.decompose_length):
        u = basis.unsigned_decompose_slice_to(j, vals, carries, count)
        signed = [int(x) if (B == 2 or int(x) < half) else int(x) - B for x in u]
        assert signed == [int(c["signed_digits"][i][j]) for i in range(count)]


def _ext_small_inputs(orc_or_none=None):
    c = EXT_SMALL
    moduli = [int(m) for m in c["moduli"]]
    n, k = 1 << c["log_n"], c["k"]
    ell = pyref.Gadget(moduli, c["log_basis"]).ell
    glwe = splitmix_rns(c["seed_glwe"], moduli, n, k + 1)
    key = splitmix_rns(c["seed_key_coeff"], moduli, n, (k + 1) * ell * (k + 1))
    return c, moduli, n, k, glwe, key


def test_oracle_extprod_small(orc):
    c, moduli, n, k, glwe, key = _ext_small_inputs()
    t, base = orc.U64DcrtTable(c["log_n"], moduli), orc.RNSBase(moduli)
    basis = orc.BigUintApproxSignedBasis(base, c["log_basis"])
    t.transform_slice(key)
    out = orc.mul_dcrt_ggsw_to(t, base, basis, k, glwe, key)
    t.inverse_transform_slice(out)
    assert out.tolist() == [int(v) for v in c["result_coeff"]]


def _digest_ids(d):
    return f"{d['kind']}-n{1 << d['log_n']}-c{d['case']}"


@pytest.mark.parametrize("d", DIGESTS, ids=_digest_ids)
def test_oracle_digests(orc, d):
    n = 1 << d["log_n"]
    if d["kind"] == "ntt_forward":
        q = int(d["q"])
        t = orc.U64NttTable(d["log_n"], q)
        assert t.root == int(d["root"])
        x = splitmix_uniform(d["seed"], q, n * d["batch"])
        t.transform_slice(x)
        assert digest(x) == d["output_sha256"]
    elif d["kind"] == "dcrt_polymul":
        moduli = [int(m) for m in d["moduli"]]
        t = orc.U64DcrtTable(d["log_n"], moduli)
        a = splitmix_rns(d["seed_a"], moduli, n, d["batch"])
        b = splitmix_rns(d["seed_b"], moduli, n, d["batch"])
        t.transform_slice(a)
        t.transform_slice(b)
        W = t.crt_poly_length
        for e in range(d["batch"]):
            t.mul_assign(a[e * W:(e + 1) * W], b[e * W:(e + 1) * W])
        t.inverse_transform_slice(a)
        assert digest(a) == d["output_sha256"]
    else:
        moduli, k = [int(m) for m in d["moduli"]], d["k"]
        t, base = orc.U64DcrtTable(d["log_n"], moduli), orc.RNSBase(moduli)
        basis = orc.BigUintApproxSignedBasis(base, d["log_basis"])
        glwe = splitmix_rns(d["seed_glwe"], moduli, n, d["batch"] * (k + 1))
        ggsw = splitmix_rns(d["seed_ggsw"], moduli, n, (k + 1) * basis.decompose_length * (k + 1))
        W = (k + 1) * t.crt_poly_length
        res = np.concatenate([orc.mul_dcrt_ggsw_to(t, base, basis, k, glwe[e * W:(e + 1) * W].copy(), ggsw)
                              for e in range(d["batch"])])
        assert digest(res) == d["output_sha256"]


# --------------------------------------------------------------------------- HIP path vs fixtures

@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("c", NTT_SMALL, ids=lambda c: f"n{1 << c['log_n']}-q{c['q']}")
def test_gpu_ntt_small(pf, c):
    t = pf.U64NttTable(c["log_n"], c["q"])
    assert t.root() == c["root"]
    x, y = u64(c["a"]), u64(c["b"])
    t.transform_slice(x)
    assert x.tolist() == [int(v) for v in c["ntt_a"]]
    t.transform_slice(y)
    from gpu_util import to_dev, to_host
    dx = to_dev(x)
    t.mul_assign_dev(dx, to_dev(y))
    t.inverse_transform_dev(dx)
    assert to_host(dx).tolist() == [int(v) for v in c["a_mul_b"]]


@pytest.mark.gpu
@pytest.mark.parametrize("c", RNS_SMALL, ids=lambda c: f"case{c['case']}")
def test_gpu_rns_gadget_small(pf, c):
    moduli = [int(m) for m in c["moduli"]]
    count = c["count"]
    base = pf.RNSBase(moduli)
    W = base.big_uint_value_len()
    vals = np.empty(count * W, np.uint64)
    base.compose_multiple_values_to(splitmix_rns(c["seed"], moduli, count), vals, count)
    assert [hex(pyref.limbs_to_int(vals[i * W:(i + 1) * W])) for i in range(count)] == c["values"]
    basis = pf.BigUintApproxSignedBasis(base, c["log_basis"], c["reverse_length"])
    assert (basis.decompose_length(), basis.drop_bits()) == (c["decompose_length"], c["drop_bits"])
    carries = np.zeros(count, np.uint8)
    basis.init_value_carry_slice_inplace(vals, carries)
    B, half = 1 << c["log_basis"], ((1 << c["log_basis"]) + 1) // 2
    u = np.empty(count, np.uint64)
    for j in range(basis.decompose_length()):
        basis.unsigned_decompose_slice_to(j, vals, u, carries)
        signed = [int(x) if (B == 2 or int(x) < half) else int(x) - B for x in u]
        assert signed == [int(c["signed_digits"][i][j]) for i in range(count)]


@pytest.mark.gpu
def test_gpu_extprod_small(pf):
    c, moduli, n, k, glwe, key = _ext_small_inputs()
    t, base = pf.U64DcrtTable(c["log_n"], moduli), pf.RNSBase(moduli)
    ctx = pf.DcrtGlevContext(t, base, pf.BigUintApproxSignedBasis(base, c["log_basis"]), k)
    t.transform_slice(key)
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, key, out, ctx, into_coeff_form=True)
    assert out.tolist() == [int(v) for v in c["result_coeff"]]


@pytest.mark.gpu
@pytest.mark.parametrize("d", DIGESTS, ids=_digest_ids)
def test_gpu_digests(pf, d):
    n = 1 << d["log_n"]
    if d["kind"] == "ntt_forward":
        q = int(d["q"])
        t = pf.U64NttTable(d["log_n"], q)
        assert t.root() == int(d["root"])
        x = splitmix_uniform(d["seed"], q, n * d["batch"])
        t.transform_slice(x)
        assert digest(x) == d["output_sha256"]
    elif d["kind"] == "dcrt_polymul":
        moduli = [int(m) for m in d["moduli"]]
        t = pf.U64DcrtTable(d["log_n"], moduli)
        from gpu_util import to_dev, to_host
        a = to_dev(splitmix_rns(d["seed_a"], moduli, n, d["batch"]))
        b = to_dev(splitmix_rns(d["seed_b"], moduli, n, d["batch"]))
        t.transform_dev(a)
        t.transform_dev(b)
        t.mul_assign_dev(a, b)
        t.inverse_transform_dev(a)
        assert digest(to_host(a)) == d["output_sha256"]
    else:
        moduli, k = [int(m) for m in d["moduli"]], d["k"]
        t, base = pf.U64DcrtTable(d["log_n"], moduli), pf.RNSBase(moduli)
        basis = pf.BigUintApproxSignedBasis(base, d["log_basis"])
        glwe = splitmix_rns(d["seed_glwe"], moduli, n, d["batch"] * (k + 1))
        ggsw = splitmix_rns(d["seed_ggsw"], moduli, n, (k + 1) * basis.decompose_length() * (k + 1))
        ctx = pf.DcrtGlevContext(t, base, basis, k)
        out = np.empty_like(glwe)
        pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
        assert digest(out) == d["output_sha256"]


# --------------------------------------------------------------------------- u32 tables, base conversion

U32 = load("u32_ntt.json")
CONV = load("base_converter.json")


def _sha32(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


def _conv_input(c):
    mod_in = [int(m) for m in c["input_moduli"]]
    return mod_in, np.concatenate([splitmix_uniform(c["seed_base"] + i, q, c["count"]) for i, q in enumerate(mod_in)])


def _check_u32(make_table, root_of):
    for c in U32["small"]:
        t = make_table(c["log_n"], c["q"])
        assert root_of(t) == c["root"]
        x = np.array(c["a"], np.uint32)
        t.transform_slice(x)
        assert x.tolist() == c["ntt_a"]
        t.inverse_transform_slice(x)
        assert x.tolist() == c["a"]
    for d in U32["digests"]:
        t = make_table(d["log_n"], d["q"])
        x = splitmix_uniform(d["seed"], d["q"], d["batch"] << d["log_n"]).astype(np.uint32)
        t.transform_slice(x)
        assert _sha32(x) == d["output_sha256"]


def test_oracle_u32_golden(orc):
    _check_u32(orc.U32NttTable, lambda t: t.root)


@pytest.mark.parametrize("c", CONV, ids=lambda c: f"case{c['case']}")
def test_oracle_converter_golden(orc, c):
    mod_in, x = _conv_input(c)
    mod_out = [int(m) for m in c["output_moduli"]]
    oin = orc.RNSBase(mod_in)
    assert orc.BaseConverter(oin, orc.RNSBase(mod_out)).fast_convert_array(x, c["count"]).tolist() == [int(v) for v in c["fast"]]
    assert orc.BaseConverter(oin, orc.RNSBase(mod_out[:1])).exact_convert_array(x, c["count"]).tolist() == \
        [int(v) for v in c["exact_to_first_output_modulus"]]


@pytest.mark.gpu
def test_gpu_u32_golden(pf):
    _check_u32(pf.U32NttTable, lambda t: t.root())


@pytest.mark.gpu
@pytest.mark.parametrize("c", CONV, ids=lambda c: f"case{c['case']}")
def test_gpu_converter_golden(pf, c):
    mod_in, x = _conv_input(c)
    mod_out = [int(m) for m in c["output_moduli"]]
    n = c["count"]
    base_in = pf.RNSBase(mod_in)
    out = np.empty(len(mod_out) * n, np.uint64)
    pf.BaseConverter(base_in, pf.RNSBase(mod_out)).fast_convert_array(x, out, n)
    assert out.tolist() == [int(v) for v in c["fast"]]
    eo = np.empty(n, np.uint64)
    pf.BaseConverter(base_in, pf.RNSBase(mod_out[:1])).exact_convert_array(x, eo, n)
    assert eo.tolist() == [int(v) for v in c["exact_to_first_output_modulus"]]
