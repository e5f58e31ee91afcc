"""The element-wise family either side of the transforms (primus_poly crt/{add,sub,neg,mul}.rs, dcrt/inv.rs;
CrtGlwe::{add,sub}_element_wise*, mul_scalar_*, mul_factor_to, mul_monic_monomial_assign).

CPU: the C restatement against definitions on Python integers and the committed fixture.
GPU (-m gpu): the HIP kernels, through the C ABI, against the restatement, the fixture and size-independent
properties at the benchmark's data size.
"""
import json
import os

import numpy as np
import pytest

from golden_inputs import digest, splitmix_rns
from pyref import Q61, Q62

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "elementwise.json")))
OPS = ["add", "sub", "neg", "mul_scalar", "add_mul_scalar", "mul_factor", "add_mul_factor", "mul_monomial", "inv"]


def case_inputs(c):
    moduli = [int(q) for q in c["moduli"]]
    n = 1 << c["log_n"]
    a = splitmix_rns(c["seed_a"], moduli, n, c["batch"])
    b = splitmix_rns(c["seed_b"], moduli, n, c["batch"])
    a[a == 0] = 1  # as the generator does: inv needs units
    return moduli, n, a, b, [int(v) for v in c["scalars"]], [int(v) for v in c["factors"]]


def run_oracle(orc, c):
    moduli, n, a, b, scalars, factors = case_inputs(c)
    o = orc.CrtPolyOps(moduli, n)
    got = {"add": o.add_to(a, b), "sub": o.sub_to(a, b), "neg": o.neg_to(a), "mul_scalar": o.mul_scalar_to(a, scalars),
           "mul_factor": o.mul_factor_to(a, factors), "inv": o.inv_to(a)}
    acc = a.copy(); o.add_mul_scalar_assign(acc, b, scalars); got["add_mul_scalar"] = acc
    acc = a.copy(); o.add_mul_factor_assign(acc, b, factors); got["add_mul_factor"] = acc
    m = a.copy(); o.mul_monomial_assign(m, c["r"]); got["mul_monomial"] = m
    return got


def check_against_fixture(c, got):
    assert sorted(got) == sorted(OPS)
    if "expected" in c:
        for op in OPS:
            assert got[op].tolist() == [int(v) for v in c["expected"][op]], op
    else:
        for op in OPS:
            assert digest(got[op]) == c["sha256"][op], op


@pytest.mark.parametrize("c", GOLD["small"] + GOLD["digests"], ids=lambda c: f"case{c['case']}")
def test_oracle_matches_fixture(orc, c):
    check_against_fixture(c, run_oracle(orc, c))


@pytest.mark.parametrize("moduli,log_n", [([97], 0), ([17, 97], 2), (Q61, 5), ([Q62, 1125899906826241], 4)])
def test_oracle_matches_integer_definitions(orc, moduli, log_n):
    rng = np.random.default_rng(log_n)
    n, L, batch = 1 << log_n, len(moduli), 3
    o = orc.CrtPolyOps(moduli, n)
    q_of = [moduli[(i // n) % L] for i in range(batch * L * n)]
    draw = lambda lo=0: np.array([int(rng.integers(lo, q)) for q in q_of], np.uint64)
    a, b = draw(1), draw()
    a[:2] = [1, q_of[1] - 1]
    b[:2] = [0, q_of[1] - 1]
    scalars = [int(rng.integers(0, q)) for q in moduli]
    scalars[0] = moduli[0] - 1
    factors = [v for s, q in zip(scalars, moduli) for v in (s, (s << 64) // q)]
    s_of = [scalars[(i // n) % L] for i in range(a.size)]
    A, B = [int(v) for v in a], [int(v) for v in b]
    assert o.add_to(a, b).tolist() == [(x + y) % q for x, y, q in zip(A, B, q_of)]
    assert o.sub_to(a, b).tolist() == [(x - y) % q for x, y, q in zip(A, B, q_of)]
    assert o.neg_to(b).tolist() == [(-y) % q for y, q in zip(B, q_of)]
    assert o.mul_scalar_to(a, scalars).tolist() == [x * s % q for x, s, q in zip(A, s_of, q_of)]
    assert o.mul_factor_to(a, factors).tolist() == [x * s % q for x, s, q in zip(A, s_of, q_of)]
    acc = a.copy(); o.add_mul_scalar_assign(acc, b, scalars)
    assert acc.tolist() == [(x + y * s) % q for x, y, s, q in zip(A, B, s_of, q_of)]
    acc = a.copy(); o.add_mul_factor_assign(acc, b, factors)
    assert acc.tolist() == [(x + y * s) % q for x, y, s, q in zip(A, B, s_of, q_of)]
    assert o.inv_to(a).tolist() == [pow(x, -1, q) for x, q in zip(A, q_of)]
    for r in sorted({0, 1, n - 1, n, min(n + 1, 2 * n - 1), 2 * n - 1}):
        m = a.copy(); o.mul_monomial_assign(m, r)
        exp = [0] * a.size
        for i, x in enumerate(A):
            d = i % n + r
            exp[(i // n) * n + d % n] = (-x if (d // n) % 2 else x) % q_of[i]
        assert m.tolist() == exp, r
    with pytest.raises(ValueError):
        o.mul_monomial_assign(a.copy(), 2 * n)
    z = a.copy(); z[z.size // 2] = 0
    with pytest.raises(ZeroDivisionError):  # the reference panics (barrett/slice.rs:551, uint/primitive.rs:91)
        o.inv_to(z)


def test_monomial_is_a_negacyclic_product(orc):
    """mul_monomial_assign(r) == schoolbook product with X^r (primus_poly/src/poly/mul.rs:107-134)."""
    q, n = 97, 8
    a = np.arange(1, n + 1, dtype=np.uint64)
    for r in range(2 * n):
        mono = np.zeros(n, np.uint64)
        mono[r % n] = 1 if r < n else q - 1
        m = a.copy(); orc.CrtPolyOps([q], n).mul_monomial_assign(m, r)
        assert np.array_equal(m, orc.naive_negacyclic_mul(q, a, mono))


# ------------------------------------------------------------------------------------------------ GPU


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def run_gpu(pf, moduli, log_n, a, b, scalars, factors, r):
    import torch
    from gpu_util import to_dev, to_host
    t = pf.U64DcrtTable(log_n, moduli)
    da, db = to_dev(a), to_dev(b)
    got = {}
    def out_of(fn, *args):
        o = torch.empty_like(da)
        fn(*args, o)
        return to_host(o)
    got["add"] = out_of(t.add_to_dev, da, db)
    got["sub"] = out_of(t.sub_to_dev, da, db)
    got["neg"] = out_of(t.neg_to_dev, da)
    got["mul_scalar"] = out_of(t.mul_scalar_to_dev, da, scalars)
    got["mul_factor"] = out_of(t.mul_factor_to_dev, da, factors)
    got["inv"] = out_of(t.inv_to_dev, da)
    acc = da.clone(); t.add_mul_scalar_assign_dev(acc, db, scalars); got["add_mul_scalar"] = to_host(acc)
    acc = da.clone(); t.add_mul_factor_assign_dev(acc, db, factors); got["add_mul_factor"] = to_host(acc)
    o = torch.empty_like(da); t.mul_monomial_to_dev(da, r, o); got["mul_monomial"] = to_host(o)
    # the in-place (assign) forms give the same words
    x = da.clone(); t.add_to_dev(x, db, x); assert np.array_equal(to_host(x), got["add"])
    x = da.clone(); t.sub_to_dev(x, db, x); assert np.array_equal(to_host(x), got["sub"])
    x = db.clone(); t.sub_to_dev(da, x, x); assert np.array_equal(to_host(x), got["sub"])  # sub_rev_assign
    x = da.clone(); t.neg_to_dev(x, x); assert np.array_equal(to_host(x), got["neg"])
    x = da.clone(); t.mul_scalar_to_dev(x, scalars, x); assert np.array_equal(to_host(x), got["mul_scalar"])
    x = da.clone(); t.mul_factor_to_dev(x, factors, x); assert np.array_equal(to_host(x), got["mul_factor"])
    x = da.clone(); t.inv_to_dev(x, x); assert np.array_equal(to_host(x), got["inv"])
    x = da.clone(); t.mul_monomial_assign_dev(x, r); assert np.array_equal(to_host(x), got["mul_monomial"])
    assert np.array_equal(to_host(da), a) and np.array_equal(to_host(db), b)  # inputs untouched
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("c", GOLD["small"] + GOLD["digests"], ids=lambda c: f"case{c['case']}")
def test_gpu_matches_fixture(pf, c):
    moduli, n, a, b, scalars, factors = case_inputs(c)
    check_against_fixture(c, run_gpu(pf, moduli, c["log_n"], a, b, scalars, factors, c["r"]))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,moduli,batch", [(0, [97], 5), (1, [17, 97], 3), (3, [Q62], 1), (4, Q61, 7), (7, Q61[:2], 33),
                                                (11, Q61, 9), (16, Q61, 2)])
def test_gpu_matches_oracle(pf, orc, log_n, moduli, batch):
    from gpu_util import rand_rns
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    a, b = rand_rns(rng, moduli, n, batch), rand_rns(rng, moduli, n, batch)
    a[a == 0] = 1
    a[0], b[0] = moduli[0] - 1, moduli[0] - 1
    scalars = [int(rng.integers(0, q)) for q in moduli]
    factors = [v for s, q in zip(scalars, moduli) for v in (s, (s << 64) // q)]
    o = orc.CrtPolyOps(moduli, n)
    for r in sorted({0, 1, n - 1, n, 2 * n - 1, int(rng.integers(0, 2 * n))}):
        got = run_gpu(pf, moduli, log_n, a, b, scalars, factors, r)
        m = a.copy(); o.mul_monomial_assign(m, r)
        assert np.array_equal(got["mul_monomial"], m), r
    assert np.array_equal(got["add"], o.add_to(a, b))
    assert np.array_equal(got["sub"], o.sub_to(a, b))
    assert np.array_equal(got["neg"], o.neg_to(a))
    assert np.array_equal(got["mul_scalar"], o.mul_scalar_to(a, scalars))
    assert np.array_equal(got["mul_factor"], o.mul_factor_to(a, factors))
    assert np.array_equal(got["inv"], o.inv_to(a))
    acc = a.copy(); o.add_mul_scalar_assign(acc, b, scalars); assert np.array_equal(got["add_mul_scalar"], acc)
    acc = a.copy(); o.add_mul_factor_assign(acc, b, factors); assert np.array_equal(got["add_mul_factor"], acc)


@pytest.mark.gpu
def test_gpu_errors(pf):
    import torch
    from gpu_util import to_dev
    moduli, log_n = Q61[:2], 4
    n = 1 << log_n
    t = pf.U64DcrtTable(log_n, moduli)
    a = to_dev(np.ones(2 * n, np.uint64))
    o = torch.empty_like(a)
    with pytest.raises(pf.PfheError) as e:
        t.mul_monomial_to_dev(a, 2 * n, o)
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:
        t.mul_monomial_to_dev(a, 1, a)  # the _to form cannot rotate in place
    assert e.value.kind == "BadArgument"
    with pytest.raises(pf.PfheError) as e:
        t.add_to_dev(a[:n], a[:n], o[:n])  # not a whole RNS polynomial
    assert e.value.kind == "BadLength"
    with pytest.raises(pf.PfheError) as e:
        t.mul_scalar_to_dev(a, [moduli[0], 1], o)  # scalar not reduced
    assert e.value.kind == "BadArgument"
    z = a.clone(); z[n + 3] = 0
    with pytest.raises(pf.PfheError) as e:  # the reference panics
        t.inv_to_dev(z, o)
    assert e.value.kind == "NoInverse"
    t.inv_to_dev(a, o)  # and the table is usable afterwards
    assert torch.equal(o, a)
    for fn in (lambda: t.add_to_dev(a[:0], a[:0], o[:0]), lambda: t.inv_to_dev(a[:0], o[:0]),
               lambda: t.mul_monomial_assign_dev(a[:0], 3)):
        fn()  # empty batches are no-ops


@pytest.mark.gpu
def test_gpu_properties_at_benchmark_size(pf):
    """BASELINE config 4's ciphertext shape (N = 2^16, 3 limbs, k = 1) x 256 ciphertexts = 768 MiB: algebraic
    identities that hold at any size, checked on the whole buffer on the device."""
    import torch
    log_n, moduli, polys = 16, Q61, 512
    n = 1 << log_n
    t = pf.U64DcrtTable(log_n, moduli)
    a = torch.empty(polys * len(moduli) * n, dtype=torch.int64, device="cuda")
    b = torch.empty_like(a)
    t.fill_uniform_dev(a, 11)
    t.fill_uniform_dev(b, 12)
    x, y = torch.empty_like(a), torch.empty_like(a)
    t.add_to_dev(a, b, x); t.sub_to_dev(x, b, x)
    assert torch.equal(x, a)                                   # (a + b) - b = a
    t.neg_to_dev(a, x); t.add_to_dev(x, a, x)
    assert int(x.count_nonzero()) == 0                          # a + (-a) = 0
    r = 40000
    t.mul_monomial_to_dev(a, r, x); t.mul_monomial_to_dev(x, 2 * n - r, y)
    assert torch.equal(y, a)                                   # X^r * X^(2N - r) = 1
    t.mul_monomial_to_dev(a, n, x); t.neg_to_dev(a, y)
    assert torch.equal(x, y)                                   # X^N = -1
    y.copy_(a); t.mul_monomial_assign_dev(y, r); t.mul_monomial_to_dev(a, r, x)
    assert torch.equal(x, y)                                   # tiled in-place form = out-of-place form
    scalars = [q - 2 for q in moduli]
    inv_s = [pow(s, -1, q) for s, q in zip(scalars, moduli)]
    t.mul_scalar_to_dev(a, scalars, x); t.mul_scalar_to_dev(x, inv_s, x)
    assert torch.equal(x, a)                                   # (a * s) * s^-1 = a
    f = [v for s, q in zip(scalars, moduli) for v in (s, (s << 64) // q)]
    t.mul_factor_to_dev(a, f, x); t.mul_scalar_to_dev(a, scalars, y)
    assert torch.equal(x, y)                                   # Shoup factor = Barrett scalar
    y.copy_(b); t.add_mul_factor_assign_dev(y, a, f); t.sub_to_dev(y, x, y)
    assert torch.equal(y, b)                                   # (b + s*a) - s*a = b
    a.clamp_(min=1)                                            # units only
    t.inv_to_dev(a, x); t.inv_to_dev(x, y)
    assert torch.equal(y, a)                                   # (a^-1)^-1 = a
    t.mul_to_dev(a, x, y)
    assert int((y != 1).count_nonzero()) == 0                   # a * a^-1 = 1


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [8, 9, 10, 12, 13, 14, 15])
def test_gpu_monomial_in_place_forms_agree(pf, log_n):
    """mul_monomial_assign: rings of 2^9 .. 2^14 words rotate inside one workgroup's registers, others (2^8, 2^15 here)
    through a scratch tile; both equal the out-of-place kernel for every kind of degree."""
    import torch
    n, polys = 1 << log_n, 37
    t = pf.U64DcrtTable(log_n, Q61)
    a = torch.empty(polys * 3 * n, dtype=torch.int64, device="cuda")
    t.fill_uniform_dev(a, log_n)
    exp = torch.empty_like(a)
    for r in (0, 1, 3, n // 2 + 1, n - 1, n, n + 2, 2 * n - 1):
        t.mul_monomial_to_dev(a, r, exp)
        x = a.clone()
        t.mul_monomial_assign_dev(x, r)
        assert torch.equal(x, exp), r
