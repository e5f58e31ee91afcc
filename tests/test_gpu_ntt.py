"""GPU parity (bit-exact) of the HIP NTT path against the CPU oracle, through the C ABI.

Mirrors primus_ntt/src/ntt/prime64/tests.rs and primus_ntt/tests/ntt.rs, with the oracle
(oracle/pfhe_oracle.c) standing where the reference's UintNttTable stands.
"""
import os

import numpy as np
import pytest

import pyref
from gpu_util import rand_mod, rand_rns, to_dev, to_host
from pyref import Q61, Q62

pytestmark = pytest.mark.gpu

PRIMES = [Q62, Q61[0], 132120577, 1125899906826241]


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def max_log(q):
    k = 0
    while (q - 1) % (1 << (k + 2)) == 0:
        k += 1
    return k


@pytest.mark.parametrize("q", PRIMES)
@pytest.mark.parametrize("log_n", list(range(0, 18)))
def test_forward_inverse_match_oracle(pf, orc, q, log_n):
    if log_n > max_log(q):
        pytest.skip("modulus has no 2N-th root")
    rng = np.random.default_rng(1000 * log_n + q % 997)
    n = 1 << log_n
    batch = 5 if log_n <= 12 else 2
    t, o = pf.U64NttTable(log_n, q), orc.U64NttTable(log_n, q)
    assert (t.poly_length(), t.modulus(), t.root(), t.inv_root(), t.inv_n()) == (n, q, o.root, o.inv_root, o.inv_n)
    a = rand_mod(rng, q, n * batch)
    a[:min(n, 4)] = [0, q - 1, 1, q // 2][:min(n, 4)]
    ref = a.copy(); o.transform_slice(ref)
    got = a.copy(); t.transform_slice(got)
    assert np.array_equal(got, ref)
    t.inverse_transform_slice(got)
    assert np.array_equal(got, a)
    # lazy variants: documented ranges, equal mod q (prime64/tests.rs:15-47,100-106)
    lz = a.copy(); t.lazy_transform_slice(lz)
    assert lz.max() < 4 * q and np.array_equal(lz % np.uint64(q), ref)
    lzi = ref.copy(); t.lazy_inverse_transform_slice(lzi)
    assert lzi.max() < 2 * q and np.array_equal(lzi % np.uint64(q), a)


@pytest.mark.parametrize("log_n", [4, 9, 12, 14, 16])
def test_lazy_forward_accepts_4q_inputs(pf, orc, log_n):
    q = Q62
    rng = np.random.default_rng(log_n)
    t, o = pf.U64NttTable(log_n, q), orc.U64NttTable(log_n, q)
    a = rng.integers(0, 4 * q, 1 << log_n, dtype=np.uint64)
    ref = (a % np.uint64(q)).copy(); o.transform_slice(ref)
    got = a.copy(); t.lazy_transform_slice(got)
    assert got.max() < 4 * q and np.array_equal(got % np.uint64(q), ref)


@pytest.mark.parametrize("batch", [1, 2, 3, 7, 33, 257])
@pytest.mark.parametrize("log_n", [3, 5, 8, 11])
def test_ragged_batches(pf, orc, log_n, batch):
    """batch sizes that do not fill a workgroup (several small polynomials share one)."""
    q = Q61[1]
    rng = np.random.default_rng(batch * 31 + log_n)
    t, o = pf.U64NttTable(log_n, q), orc.U64NttTable(log_n, q)
    a = rand_mod(rng, q, batch << log_n)
    ref = a.copy(); o.transform_slice(ref)
    got = a.copy(); t.transform_slice(got)
    assert np.array_equal(got, ref)
    t.inverse_transform_slice(got)
    assert np.array_equal(got, a)


def test_empty_and_bad_lengths(pf):
    t = pf.U64NttTable(6, Q61[0])
    t.transform_slice(np.empty(0, np.uint64))  # empty batch is a no-op
    for bad in (1, 63, 65, 100):
        with pytest.raises(pf.PfheError) as e:
            t.transform_slice(np.zeros(bad, np.uint64))
        assert e.value.kind == "BadLength"
    d = pf.U64DcrtTable(6, Q61)
    with pytest.raises(pf.PfheError) as e:
        d.transform_slice(np.zeros(64 * 2, np.uint64))  # needs a multiple of L*N
    assert e.value.kind == "BadLength"


def test_create_errors(pf):
    with pytest.raises(pf.PfheError) as e:
        pf.U64NttTable(20, 1125899906826241)
    assert e.value.kind == "NoPrimitiveRoot"
    with pytest.raises(pf.PfheError) as e:
        pf.U64DcrtTable(16, [Q61[0], 132120577 * 0 + 1125899906826241])
    assert e.value.kind == "NoPrimitiveRoot"


@pytest.mark.parametrize("q,log_n", [(132120577, 10), (Q61[2], 6), (Q62, 13), (Q61[0], 16), (Q61[0], 2), (Q61[0], 0)])
def test_monomial_transforms(pf, orc, q, log_n):
    t, o = pf.U64NttTable(log_n, q), orc.U64NttTable(log_n, q)
    n = 1 << log_n
    rng = np.random.default_rng(9)
    out = np.empty(n, np.uint64)
    for degree in sorted({0, 1, 2, n // 2, n - 1, n, n + 3, 2 * n - 1}):
        for coeff in [0, 1, q - 1, int(rng.integers(2, q - 1))]:
            t.transform_monomial(coeff, degree, out)
            assert np.array_equal(out, o.transform_monomial(coeff, degree)), (degree, coeff)
        t.transform_coeff_one_monomial(degree, out)
        assert np.array_equal(out, o.transform_coeff_one_monomial(degree))
        t.transform_coeff_minus_one_monomial(degree, out)
        assert np.array_equal(out, o.transform_coeff_minus_one_monomial(degree))


@pytest.mark.parametrize("log_n,batch", [(3, 4), (10, 3), (14, 2), (16, 2)])
def test_dcrt_table(pf, orc, log_n, batch):
    """U64DcrtTable: modulus-major limbs, each with its own table (dcrt/prime64.rs:98-127)."""
    rng = np.random.default_rng(log_n)
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    assert (d.poly_length(), d.moduli_count(), d.crt_poly_length()) == (1 << log_n, 3, 3 << log_n)
    assert d.moduli() == Q61 and d.roots() == [o.table(i).root for i in range(3)]
    a = rand_rns(rng, Q61, 1 << log_n, batch)
    ref = a.copy(); o.transform_slice(ref)
    got = a.copy(); d.transform_slice(got)
    assert np.array_equal(got, ref)
    d.inverse_transform_slice(got)
    assert np.array_equal(got, a)
    mono = np.empty(3 << log_n, np.uint64)
    d.transform_monomial(5, 3, mono)
    exp = np.concatenate([o.table(i).transform_monomial(5, 3) for i in range(3)])
    assert np.array_equal(mono, exp)
    d.transform_coeff_one_monomial(7, mono)
    assert np.array_equal(mono, np.concatenate([o.table(i).transform_coeff_one_monomial(7) for i in range(3)]))
    d.transform_coeff_minus_one_monomial(7, mono)
    assert np.array_equal(mono, np.concatenate([o.table(i).transform_coeff_minus_one_monomial(7) for i in range(3)]))


@pytest.mark.parametrize("log_n,batch", [(4, 3), (11, 5), (13, 2)])
def test_pointwise_device_ops(pf, orc, log_n, batch):
    """DcrtPolynomial::mul_assign / add_mul_assign, elementwise and shared multiplicand."""
    rng = np.random.default_rng(7 + log_n)
    n = 1 << log_n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    W = 3 * n
    a, b, acc = (rand_rns(rng, Q61, n, batch) for _ in range(3))
    bs = b[:W].copy()
    for shared in (False, True):
        bb = bs if shared else b
        exp_mul, exp_fma = a.copy(), acc.copy()
        for i in range(batch):
            bi = bb if shared else bb[i * W:(i + 1) * W]
            x = exp_mul[i * W:(i + 1) * W]; o.mul_assign(x, np.ascontiguousarray(bi))
            y = exp_fma[i * W:(i + 1) * W]; o.add_mul_assign(y, np.ascontiguousarray(a[i * W:(i + 1) * W]), np.ascontiguousarray(bi))
        da, db, dacc = to_dev(a), to_dev(bb), to_dev(acc)
        d.add_mul_assign_dev(dacc, da, db)
        d.mul_assign_dev(da, db)
        assert np.array_equal(to_host(da), exp_mul)
        assert np.array_equal(to_host(dacc), exp_fma)


@pytest.mark.parametrize("log_n", [3, 5])
def test_polymul_equals_schoolbook(pf, log_n):
    """NTT -> pointwise -> INTT == schoolbook product mod (X^N+1, q) on Python integers."""
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    d = pf.U64DcrtTable(log_n, Q61)
    a, b = rand_rns(rng, Q61, n, 2), rand_rns(rng, Q61, n, 1)
    bh = b.copy(); d.transform_slice(bh)
    da = to_dev(a)
    d.mul_dcrt_polynomial_dev(da, to_dev(bh))
    got = to_host(da)
    for e in range(2):
        for r, q in enumerate(Q61):
            exp = pyref.negacyclic_mul(a[(e * 3 + r) * n:(e * 3 + r + 1) * n], b[r * n:(r + 1) * n], q)
            assert got[(e * 3 + r) * n:(e * 3 + r + 1) * n].tolist() == exp


def test_config3_shape_vs_oracle(pf, orc):
    """BASELINE config 3 shape (N=2^16, 3 primes) on a small batch: fused polymul == oracle."""
    log_n, batch = 16, 3
    rng = np.random.default_rng(3)
    n = 1 << log_n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    a, bh = rand_rns(rng, Q61, n, batch), rand_rns(rng, Q61, n, 1)
    exp = a.copy(); o.transform_slice(exp)
    for i in range(batch):
        o.mul_assign(exp[i * 3 * n:(i + 1) * 3 * n], bh)
    o.inverse_transform_slice(exp)
    da = to_dev(a)
    d.mul_dcrt_polynomial_dev(da, to_dev(bh))
    assert np.array_equal(to_host(da), exp)


def test_full_size_roundtrip_and_spot_checks(pf, orc):
    """BASELINE config 3' at its full batch (4096 x 3 x 2^16 words = 6 GiB): device-generated
    inputs, forward + inverse round trip over the whole batch, oracle spot checks."""
    import ctypes as C
    import torch
    log_n, batch = 16, 4096
    n, L = 1 << log_n, 3
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    words = batch * L * n
    x = torch.empty(words, dtype=torch.int64, device="cuda")
    mod = np.array(Q61, np.uint64)
    from primus_fhe_amd._lib import check, u64p
    check(pf.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mod.ctypes.data_as(u64p), 3, n,
                                         0x5EED000000000003, None))
    orig = x.clone()
    d.transform_dev(x)
    torch.cuda.synchronize()
    for e in (0, 1777, batch - 1):
        ref = to_host(orig[e * L * n:(e + 1) * L * n]).copy()
        assert all(int(ref[r * n:(r + 1) * n].max()) < Q61[r] for r in range(3))
        o.transform_slice(ref)
        assert np.array_equal(to_host(x[e * L * n:(e + 1) * L * n]), ref)
    d.inverse_transform_dev(x)
    assert torch.equal(x, orig)


@pytest.mark.parametrize("q", [Q61[0], Q61[2], 1125899906826241, 1152921504606830593, 562949953392641])
@pytest.mark.parametrize("log_n", [4, 9, 12, 13])
def test_pseudo_mersenne_path_equals_generic_path(pf, orc, q, log_n, monkeypatch):
    """Primes q = 2^K - c take the PmArith kernels; PFHE_DISABLE_PM forces the generic ShoupArith
    kernels for the same table.  Canonical outputs must be identical (and equal to the oracle),
    lazy outputs must agree mod q."""
    if log_n > max_log(q):
        pytest.skip("modulus has no 2N-th root")
    rng = np.random.default_rng(log_n + q % 1000)
    n = 1 << log_n
    a = rand_mod(rng, q, 3 * n)
    a[:4] = [0, q - 1, 1, q // 2]
    o = orc.U64NttTable(log_n, q)
    ref = a.copy(); o.transform_slice(ref)
    outs = []
    for disable in (False, True):
        if disable:
            monkeypatch.setenv("PFHE_DISABLE_PM", "1")
        else:
            monkeypatch.delenv("PFHE_DISABLE_PM", raising=False)
        t = pf.U64NttTable(log_n, q)
        f = a.copy(); t.transform_slice(f)
        assert np.array_equal(f, ref)
        lz = a.copy(); t.lazy_transform_slice(lz)
        assert lz.max() < 4 * q and np.array_equal(lz % np.uint64(q), ref)
        lzi = ref.copy(); t.lazy_inverse_transform_slice(lzi)
        assert lzi.max() < 2 * q and np.array_equal(lzi % np.uint64(q), a)
        t.inverse_transform_slice(f)
        assert np.array_equal(f, a)
        outs.append(f)
    assert np.array_equal(outs[0], outs[1])


def test_large_batch_forms_match_oracle(pf, orc, monkeypatch):
    """Batches >= 256 MiB of N = 2^16 run as tiles + 1 launches of the pipelined kernel on the caller's stream
    (pfhe_ntt.hip `transform`); results must equal the two plain launches and the oracle, and work
    queued on the caller's stream afterwards must see the finished data."""
    import torch
    log_n, batch = 16, 352  # 352 * 3 * 512 KiB = 528 MiB
    n, L = 1 << log_n, 3
    rng = np.random.default_rng(77)
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    a = rand_rns(rng, Q61, n, batch)
    x = to_dev(a)
    d.transform_dev(x)
    y = x.clone()  # ordered after the join on the same (current) stream
    got = to_host(y)
    # two forms of the same transform (switches are read when a table is created): pipelined (the default at this
    # size) and one launch per pass
    monkeypatch.setenv("PFHE_DISABLE_PIPELINED", "1")
    d1 = pf.U64DcrtTable(log_n, Q61)
    monkeypatch.delenv("PFHE_DISABLE_PIPELINED")
    for other in (d1,):
        x1 = to_dev(a)
        other.transform_dev(x1)
        z = x1.clone()
        assert np.array_equal(got, to_host(z))
        other.inverse_transform_dev(x1)
        assert np.array_equal(to_host(x1), a)
    for e in (0, 123, batch - 1):
        ref = a[e * L * n:(e + 1) * L * n].copy()
        o.transform_slice(ref)
        assert np.array_equal(got[e * L * n:(e + 1) * L * n], ref)
    d.inverse_transform_dev(x)
    assert np.array_equal(to_host(x), a)


@pytest.mark.parametrize("log_n,batch", [(4, 3), (12, 4)])
def test_glwe_butterfly_ops(pf, orc, log_n, batch):
    """DcrtGlwe::butterfly_mul_factor_to / butterfly_mul_dcrt_polynomial_to (glwe/dcrt.rs:128-175):
    (a, b) = (a + s, (a - s) * w) with a shared multiplicand, ShoupFactor pairs or plain residues."""
    rng = np.random.default_rng(log_n)
    n, W = 1 << log_n, 3 << log_n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    a, s = rand_rns(rng, Q61, n, batch), rand_rns(rng, Q61, n, batch)
    a[:3] = [0, Q61[0] - 1, 5]; s[:3] = [Q61[0] - 1, Q61[0] - 1, 7]
    w = rand_rns(rng, Q61, n, 1)
    pairs = np.empty(2 * W, np.uint64)
    pairs[0::2] = w
    pairs[1::2] = [(int(v) << 64) // Q61[i // n] for i, v in enumerate(w)]
    exp_a, exp_b = a.copy(), np.empty_like(a)
    for e in range(batch):
        exp_b[e * W:(e + 1) * W] = o.butterfly_mul_factor_to(exp_a[e * W:(e + 1) * W], s[e * W:(e + 1) * W].copy(), pairs)
    chk = a[:W].copy()
    assert np.array_equal(o.butterfly_mul_to(chk, s[:W].copy(), w), exp_b[:W])  # both reference forms agree
    for factor in (True, False):
        da, db = to_dev(a), to_dev(np.zeros_like(a))
        if factor:
            d.butterfly_mul_factor_to_dev(da, to_dev(s), to_dev(pairs), db)
        else:
            d.butterfly_mul_dcrt_polynomial_to_dev(da, to_dev(s), to_dev(w), db)
        assert np.array_equal(to_host(da), exp_a) and np.array_equal(to_host(db), exp_b)
    with pytest.raises(pf.PfheError) as e:
        d.butterfly_mul_dcrt_polynomial_to_dev(to_dev(a), to_dev(s), to_dev(w[:-1].copy()), to_dev(a))
    assert e.value.kind == "BadLength"


@pytest.mark.parametrize("log_n,moduli,batch", [
    (4, Q61, 5), (7, Q61[:1], 3), (10, [1125899906826241, 1125899906629633], 4), (12, Q61, 3),
    (13, [Q62], 2), (14, Q61[:2], 2), (15, [Q62, Q61[0]], 2), (16, Q61, 2), (17, [Q62], 1),
])
@pytest.mark.parametrize("shared", [True, False])
def test_fused_polymul_matches_oracle(pf, orc, log_n, moduli, batch, shared):
    """pfhe_dcrt_mul_dcrt_polynomial_dev (NTT -> product fused into the inverse transform's first
    pass -> INTT) for both arithmetic policies, one shared or one multiplicand per polynomial."""
    rng = np.random.default_rng(log_n + batch)
    n, L = 1 << log_n, len(moduli)
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    a = rand_rns(rng, moduli, n, batch)
    bh = rand_rns(rng, moduli, n, 1 if shared else batch)
    exp = a.copy(); o.transform_slice(exp)
    for i in range(batch):
        s = slice(i * L * n, (i + 1) * L * n)
        o.mul_assign(exp[s], bh[:L * n] if shared else bh[s])
    o.inverse_transform_slice(exp)
    da = to_dev(a)
    d.mul_dcrt_polynomial_dev(da, to_dev(bh))
    assert np.array_equal(to_host(da), exp)


def test_misaligned_device_buffer_is_rejected(pf):
    """Device buffers must be 16-byte aligned (the kernels move 16-byte vectors)."""
    import torch
    d = pf.U64DcrtTable(10, Q61)
    x = torch.zeros(3 * 1024 + 1, dtype=torch.int64, device="cuda")
    with pytest.raises(pf.PfheError) as e:
        d.transform_dev(x[1:])
    assert e.value.kind == "BadArgument" and "aligned" in str(e.value)
    d.transform_dev(x[:-1])
    t32 = pf.U32NttTable(10, 132120577)
    y = torch.zeros(1024 + 2, dtype=torch.int32, device="cuda")
    with pytest.raises(pf.PfheError) as e:
        t32.transform_dev(y[2:])
    assert e.value.kind == "BadArgument"


@pytest.mark.parametrize("log_n,moduli,batch", [(0, [97], 5), (3, Q61, 3), (10, Q61[:2], 4), (13, [Q62], 2)])
def test_out_of_place_products(pf, orc, log_n, moduli, batch):
    """NttPolynomial / DcrtPolynomial mul_to and mul_add_to (ntt/mul.rs:100-107, ntt/mod.rs:169-187)."""
    import torch
    rng = np.random.default_rng(log_n)
    n, L = 1 << log_n, len(moduli)
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    if n * L % 2:  # odd word count per unit: keep every device buffer 16-byte aligned by using one unit
        batch = 2
    a, b, c = (rand_rns(rng, moduli, n, batch) for _ in range(3))
    W = L * n
    exp_mul, exp_fma = a.copy(), c.copy()
    for e in range(batch):
        o.mul_assign(exp_mul[e * W:(e + 1) * W], b[e * W:(e + 1) * W])
        o.add_mul_assign(exp_fma[e * W:(e + 1) * W], a[e * W:(e + 1) * W], b[e * W:(e + 1) * W])
    da, db, dc = to_dev(a), to_dev(b), to_dev(c)
    out = torch.zeros_like(da)
    d.mul_to_dev(da, db, out)
    assert np.array_equal(to_host(out), exp_mul) and np.array_equal(to_host(da), a)
    d.mul_add_to_dev(da, db, dc, out)
    assert np.array_equal(to_host(out), exp_fma) and np.array_equal(to_host(dc), c)
    if L == 1:
        t = pf.U64NttTable(log_n, moduli[0])
        t.mul_add_to_dev(da, db, dc, out)
        assert np.array_equal(to_host(out), exp_fma)
        t.mul_to_dev(da, db, da)  # output may alias an input
        assert np.array_equal(to_host(da), exp_mul)


@pytest.mark.parametrize("log_n,moduli,k,batch", [(4, Q61, 1, 5), (9, Q61[:2], 2, 3), (12, Q61, 1, 4)])
def test_add_dcrt_glwe_mul_dcrt_polynomial_assign(pf, orc, log_n, moduli, k, batch):
    """glwe/dcrt.rs:107-126, batched: one multiplicand polynomial per ciphertext, shared by its k+1 polynomials."""
    rng = np.random.default_rng(log_n + k)
    n, L = 1 << log_n, len(moduli)
    W = L * n
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    acc, glwe = rand_rns(rng, moduli, n, batch * (k + 1)), rand_rns(rng, moduli, n, batch * (k + 1))
    poly = rand_rns(rng, moduli, n, batch)
    exp = acc.copy()
    for e in range(batch):
        for c in range(k + 1):
            s = slice((e * (k + 1) + c) * W, (e * (k + 1) + c + 1) * W)
            o.add_mul_assign(exp[s], glwe[s], poly[e * W:(e + 1) * W])
    dacc = to_dev(acc)
    d.add_dcrt_glwe_mul_dcrt_polynomial_assign_dev(dacc, to_dev(glwe), to_dev(poly), k + 1)
    assert np.array_equal(to_host(dacc), exp)
    # DcrtGlwe::mul_dcrt_polynomial_to (glwe/dcrt.rs:377-395), batched the same way; result may alias the input
    expm = glwe.copy()
    for e in range(batch):
        for c in range(k + 1):
            s = slice((e * (k + 1) + c) * W, (e * (k + 1) + c + 1) * W)
            o.mul_assign(expm[s], poly[e * W:(e + 1) * W])
    dg, dres = to_dev(glwe), to_dev(acc)
    d.glwe_mul_dcrt_polynomial_to_dev(dg, to_dev(poly), dres, k + 1)
    assert np.array_equal(to_host(dres), expm) and np.array_equal(to_host(dg), glwe)
    d.glwe_mul_dcrt_polynomial_to_dev(dg, to_dev(poly), dg, k + 1)
    assert np.array_equal(to_host(dg), expm)
    with pytest.raises(pf.PfheError) as e:
        d.add_dcrt_glwe_mul_dcrt_polynomial_assign_dev(dacc, to_dev(glwe), to_dev(poly[:-W].copy()), k + 1)
    assert e.value.kind == "BadLength"


@pytest.mark.parametrize("shared", [True, False])
def test_fused_polymul_large_batch_tiled_path(pf, orc, shared):
    """Batches >= 256 MiB run the pipelined forms (tiles + 2 launches of the three-pass product); the product must
    follow the tiling of a per-element multiplicand.  Compared with the plain three-pass, four-pass and unfused forms on
    the whole batch and with the oracle on three elements."""
    import os
    import torch
    log_n, batch = 16, 360  # 360 x 3 x 512 KiB = 540 MiB
    n, L = 1 << log_n, 3
    W = L * n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    mods = np.array(Q61, np.uint64)
    import ctypes as C
    from primus_fhe_amd._lib import check, u64p

    def fill(words, seed):
        x = torch.empty(words, dtype=torch.int64, device="cuda")
        check(pf.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), L, n, seed, None))
        return x

    a, bh = fill(batch * W, 11), fill(W if shared else batch * W, 12)
    fused = a.clone()
    d.mul_dcrt_polynomial_dev(fused, bh)
    # d runs the pipelined forms; d_tiled one launch per pass (switches are read at table creation)
    os.environ["PFHE_DISABLE_PIPELINED"] = "1"
    try:
        d_tiled = pf.U64DcrtTable(log_n, Q61)
    finally:
        del os.environ["PFHE_DISABLE_PIPELINED"]
    tiled = a.clone()
    d_tiled.mul_dcrt_polynomial_dev(tiled, bh)
    inv_tiled = a.clone()
    d_tiled.inverse_transform_dev(inv_tiled)
    assert torch.equal(fused, tiled)
    inv_plain = a.clone()
    d.inverse_transform_dev(inv_plain)
    assert torch.equal(inv_plain, inv_tiled)
    # the same product composed from the separate entry points (transform, point-wise product, inverse transform)
    plain = a.clone()
    d_tiled.transform_dev(plain)
    d_tiled.mul_assign_dev(plain, bh)
    d_tiled.inverse_transform_dev(plain)
    assert torch.equal(fused, plain)


@pytest.mark.parametrize("L,batch,tiles", [(3, 171, 0), (3, 173, 5), (1, 513, 3), (1, 700, 7), (3, 352, 64), (2, 257, 0)])
def test_pipelined_form_tile_arithmetic(pf, L, batch, tiles, monkeypatch):
    """Odd batch sizes, tile counts that do not divide them, more tiles than is sensible: the pipelined form must equal
    the two plain launches bit for bit, forward and inverse (switches are read when a table is created)."""
    import torch
    log_n = 16
    n = 1 << log_n
    moduli = Q61[:L]
    if tiles:
        monkeypatch.setenv("PFHE_PIPE_TILES", str(tiles))
    t = pf.U64DcrtTable(log_n, moduli)
    name, launches = t.transform_form(batch * L * n)
    assert name == "ntt_pipe_fwd_kernel" and launches >= 3
    monkeypatch.setenv("PFHE_DISABLE_PIPELINED", "1")
    plain = pf.U64DcrtTable(log_n, moduli)
    assert plain.transform_form(batch * L * n) == ("ntt_strided_kernel<K=4,fwd> + ntt_block_kernel<12,fwd>", 2)
    x = _fill(pf, batch * L * n, moduli, n, 1234 + batch)
    y = x.clone()
    t.transform_dev(x)
    plain.transform_dev(y)
    assert torch.equal(x, y)
    t.inverse_transform_dev(x)
    plain.inverse_transform_dev(y)
    assert torch.equal(x, y)


def test_transform_form_reports_the_launch_plan(pf, monkeypatch):
    """pfhe_dcrt_transform_form: what bench.py's roofline object is built from."""
    n, L = 1 << 16, 3
    t = pf.U64DcrtTable(16, Q61)
    assert t.transform_form(4096 * L * n) == ("ntt_pipe_fwd_kernel", 25)
    assert t.transform_form(4096 * L * n, inverse=True) == ("ntt_pipe_inv_kernel", 25)
    assert t.transform_form(256 * L * n) == ("ntt_pipe_fwd_kernel", 3)      # 384 MiB: 2 tiles
    assert t.transform_form(64 * L * n) == ("ntt_strided_kernel<K=4,fwd> + ntt_block_kernel<12,fwd>", 2)
    assert pf.U64DcrtTable(12, [Q61[0]]).transform_form(1 << 12) == ("ntt_block_kernel<12,fwd>", 1)
    t14 = pf.U64DcrtTable(14, [Q61[0]])     # BASELINE config 2: resident workgroups from two polynomials per CU
    assert t14.transform_form(4096 << 14) == ("ntt_persist_kernel<14,fwd>", 1)
    assert t14.transform_form(4096 << 14, inverse=True) == ("ntt_persist_kernel<14,inv>", 1)
    assert t14.transform_form(16 << 14) == ("ntt_block_kernel<14,fwd>", 1)
    monkeypatch.setenv("PFHE_DISABLE_PIPELINED", "1")
    t2 = pf.U64DcrtTable(16, Q61)
    assert t2.transform_form(4096 * L * n) == ("ntt_strided_kernel<K=4,fwd> + ntt_block_kernel<12,fwd>", 2)
    assert t2.transform_form(4096 * L * n, inverse=True) == ("ntt_block_kernel<12,inv> + ntt_strided_kernel<K=4,inv>", 2)
    with pytest.raises(pf.PfheError) as e:
        t.transform_form(5)
    assert e.value.kind == "BadLength"


def _fill(pf, words, moduli, n, seed):
    import ctypes as C
    import torch
    from primus_fhe_amd._lib import check, u64p
    x = torch.empty(words, dtype=torch.int64, device="cuda")
    mods = np.array(moduli, np.uint64)
    check(pf.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), words, mods.ctypes.data_as(u64p), len(moduli), n,
                                         seed, None))
    return x


def test_config3_full_batch_every_element(pf, orc):
    """BASELINE config 3' at its full batch: EVERY one of the 12 288 limb transforms is compared with the oracle's
    scalar restatement (all host cores, slab by slab), forward and inverse — a tiling or tile-boundary bug that
    cancels in a round trip cannot hide."""
    import torch
    from gpu_util import oracle_map
    log_n, batch = 16, 4096
    n, L = 1 << log_n, 3
    W = L * n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    x = _fill(pf, batch * W, Q61, n, 0x5EED000000000003)
    orig = x.clone()
    d.transform_dev(x)
    torch.cuda.synchronize()
    slab = 256  # RNS polynomials per host slab (384 MiB)
    for s0 in range(0, batch, slab):
        ref = to_host(orig[s0 * W:(s0 + slab) * W]).copy()
        oracle_map(o.transform_slice, ref, W)
        assert np.array_equal(to_host(x[s0 * W:(s0 + slab) * W]), ref), s0
    # inverse direction on the transformed batch, every element again (the oracle inverts its own forward output)
    y = x.clone()
    d.inverse_transform_dev(y)
    assert torch.equal(y, orig)
    for s0 in (0, batch // 2, batch - slab):  # and the oracle's inverse of the GPU's forward output
        ref = to_host(x[s0 * W:(s0 + slab) * W]).copy()
        oracle_map(o.inverse_transform_slice, ref, W)
        assert np.array_equal(to_host(orig[s0 * W:(s0 + slab) * W]), ref), s0


@pytest.mark.parametrize("shared", [True, False])
def test_config3_polymul_full_batch_every_element(pf, orc, shared):
    """BASELINE config 3 at its full batch: NTT -> pointwise product -> INTT of all 4096 RNS polynomials (the pipelined
    three-pass form: tiles + 2 launches of ntt_pipe_mid_kernel), EVERY limb polynomial compared with the oracle
    (transform_slice, DcrtPolynomial::mul_assign, inverse_transform_slice), shared and per-element multiplicand."""
    import torch
    from gpu_util import oracle_map
    log_n, batch = 16, 4096
    n, L = 1 << log_n, 3
    W = L * n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    x = _fill(pf, batch * W, Q61, n, 0x5EED000000000003)
    bh = _fill(pf, W if shared else batch * W, Q61, n, 77)
    orig = x.clone()
    d.mul_dcrt_polynomial_dev(x, bh)
    torch.cuda.synchronize()
    hb = to_host(bh)
    slab = 256
    for s0 in range(0, batch, slab):
        ref = to_host(orig[s0 * W:(s0 + slab) * W]).copy()
        oracle_map(o.transform_slice, ref, W)
        if shared:
            oracle_map(lambda c: [o.mul_assign(c[i:i + W], hb) for i in range(0, c.size, W)], ref, W)
        else:
            mb = hb[s0 * W:(s0 + slab) * W]
            for i in range(slab):
                o.mul_assign(ref[i * W:(i + 1) * W], mb[i * W:(i + 1) * W])
        oracle_map(o.inverse_transform_slice, ref, W)
        assert np.array_equal(to_host(x[s0 * W:(s0 + slab) * W]), ref), s0


def test_config2_full_batch(pf, orc):
    """BASELINE config 2: N = 2^14, one 61-bit prime, 4096 polynomials — forward and inverse, every polynomial
    against the oracle."""
    import torch
    from gpu_util import oracle_map
    log_n, batch, q = 14, 4096, Q61[0]
    n = 1 << log_n
    d, o = pf.U64NttTable(log_n, q), orc.U64NttTable(log_n, q)
    x = _fill(pf, batch * n, [q], n, 0x5EED000000000002)
    orig = x.clone()
    d.transform_dev(x)
    ref = to_host(orig).copy()
    oracle_map(o.transform_slice, ref, n)
    assert np.array_equal(to_host(x), ref)
    inv = to_host(x).copy()
    d.inverse_transform_dev(x)
    assert torch.equal(x, orig)
    oracle_map(o.inverse_transform_slice, inv, n)
    assert np.array_equal(inv, to_host(orig))


WIDE40 = [1099511628161, 1099511629121, 1099511629889, 1099511630209, 1099511630593, 1099511630849, 1099511631937,
          1099511633153, 1099511634113, 1099511635009, 1099511636161, 1099511638529, 1099511639297, 1099511640001,
          1099511641153, 1099511641729, 1099511643137, 1099511643521]  # 18 primes = 1 mod 64


def test_monomials_of_a_wide_rns_base(pf, orc):
    """DcrtTable::transform_monomial / coeff_one / coeff_minus_one (dcrt/mod.rs:105-134) with more moduli than one launch
    carries scalars for (16): the limbs are served in groups."""
    log_n = 4
    n = 1 << log_n
    d = pf.U64DcrtTable(log_n, WIDE40)
    tabs = [orc.U64NttTable(log_n, q) for q in WIDE40]
    out = np.empty(len(WIDE40) * n, np.uint64)
    for degree in (0, 1, n - 1, n + 3):
        d.transform_monomial(5, degree, out)
        assert np.array_equal(out, np.concatenate([t.transform_monomial(5, degree) for t in tabs])), degree
        d.transform_coeff_minus_one_monomial(degree, out)
        assert np.array_equal(out, np.concatenate([t.transform_coeff_minus_one_monomial(degree) for t in tabs]))
        d.transform_coeff_one_monomial(degree, out)
        assert np.array_equal(out, np.concatenate([t.transform_coeff_one_monomial(degree) for t in tabs]))


@pytest.mark.parametrize("log_n", [4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("lazy", [False, True])
def test_dcrt_small_rings_with_per_lane_primes(pf, orc, log_n, lazy):
    """Block passes below 2^10 hold several polynomials per wave, so the prime (q, c, K) is a per-lane value in the asm
    butterflies (pfhe_pm_asm.hpp): forward and inverse DCRT transforms of three pseudo-Mersenne limbs, batch 8, against
    the oracle for every small size."""
    rng = np.random.default_rng(70 + log_n)
    n, batch = 1 << log_n, 8
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    a = rand_rns(rng, Q61, n, batch)
    fwd, exp = to_dev(a), a.copy()
    o.transform_slice(exp)
    if lazy:
        d.transform_dev(fwd, lazy=True)
        got = to_host(fwd)
        qs = np.tile(np.repeat(np.array(Q61, np.uint64), n), batch)
        assert np.all(got < 4 * qs) and np.array_equal(got % qs, exp)
        inv = to_dev(exp)
        d.inverse_transform_dev(inv, lazy=True)
        gi = to_host(inv)
        assert np.all(gi < 2 * qs) and np.array_equal(gi % qs, a)
    else:
        d.transform_dev(fwd)
        assert np.array_equal(to_host(fwd), exp)
        d.inverse_transform_dev(fwd)
        assert np.array_equal(to_host(fwd), a)


# generic NTT-friendly primes (= 1 mod 2^18, NOT of pseudo-Mersenne shape), 61 / 61 / 59 / 45 / 33 bits.  The
# Montgomery-form transforms (MontArith) take tables whose primes all lie in [2^48, 2^61); a table with a smaller prime
# keeps the Shoup transforms, and the same assertions hold for it.
GENERIC = [1635294906373636097, 1220953465133989889, 387168985270714369, 27672964759553, 7717519361]
# the edges of that range: the largest such prime below 2^61 (forward lazy values reach 2^63 + 3q, 98 % of 2^64), 53 bits,
# the first one above 2^48 (where the quotient estimate of the closing reduction is least exact) and the last one below
GENERIC_EDGE = [2305843008942112769, 6804446355652609, 281474980380673, 281474975662081]


@pytest.mark.parametrize("log_n,moduli,batch", [
    (4, GENERIC[:3], 5), (5, GENERIC[1:4], 8), (7, GENERIC[:1], 3), (9, GENERIC[2:5], 4), (10, GENERIC[:2], 3),
    (12, GENERIC[3:], 2), (13, GENERIC[:2], 2), (14, GENERIC[1:2], 2), (15, GENERIC[:2], 2), (16, GENERIC[:3], 2),
    (17, GENERIC[4:], 1),
    (4, GENERIC_EDGE[:3], 4), (6, GENERIC_EDGE[:3], 3), (8, GENERIC_EDGE[1:3], 4), (9, GENERIC_EDGE[2:3], 5),
    (11, GENERIC_EDGE[:3], 2), (12, GENERIC_EDGE[2:3], 3), (13, GENERIC_EDGE[1:3], 2), (14, GENERIC_EDGE[:1], 2),
    (16, GENERIC_EDGE[:3], 2), (16, GENERIC_EDGE[2:], 1), (17, GENERIC_EDGE[:1], 1),
])
def test_generic_primes_montgomery_transforms(pf, orc, log_n, moduli, batch, monkeypatch):
    """Tables whose primes are below 2^61 but not pseudo-Mersenne run their transforms in Montgomery form (MontArith: 7
    multiplies per twiddle product instead of Shoup's 10): canonical outputs equal the oracle's and the Shoup path's
    (PFHE_DISABLE_MONT), lazy outputs honour [0,4q) / [0,2q) and agree mod q, for every plan shape; the fused
    NTT -> product -> INTT too."""
    rng = np.random.default_rng(900 + log_n)
    n, L = 1 << log_n, len(moduli)
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    monkeypatch.setenv("PFHE_DISABLE_MONT", "1")
    d_shoup = pf.U64DcrtTable(log_n, moduli)
    monkeypatch.delenv("PFHE_DISABLE_MONT")
    a = rand_rns(rng, moduli, n, batch)
    a[:L * n:n] = [q - 1 for q in moduli]  # extreme residues in the first polynomial
    exp = a.copy(); o.transform_slice(exp)
    x, xs = to_dev(a), to_dev(a)
    d.transform_dev(x); d_shoup.transform_dev(xs)
    assert np.array_equal(to_host(x), exp) and np.array_equal(to_host(xs), exp)
    d.inverse_transform_dev(x)
    assert np.array_equal(to_host(x), a)
    qs = np.tile(np.repeat(np.array(moduli, np.uint64), n), batch)
    lz = to_dev((a + qs * rng.integers(0, 4, a.size).astype(np.uint64)))  # lazy inputs in [0,4q)
    d.transform_dev(lz, lazy=True)
    got = to_host(lz)
    assert np.all(got < 4 * qs) and np.array_equal(got % qs, exp)
    li = to_dev(exp)
    d.inverse_transform_dev(li, lazy=True)
    gi = to_host(li)
    assert np.all(gi < 2 * qs) and np.array_equal(gi % qs, a)
    # NTT -> product -> INTT (shared multiplicand)
    bh = rand_rns(rng, moduli, n, 1)
    pe = exp.copy()
    for i in range(batch):
        o.mul_assign(pe[i * L * n:(i + 1) * L * n], bh)
    o.inverse_transform_slice(pe)
    pa = to_dev(a)
    d.mul_dcrt_polynomial_dev(pa, to_dev(bh))
    assert np.array_equal(to_host(pa), pe)


def test_generic_primes_large_batch_pipelined(pf, orc):
    """The pipelined forms (tiles + 1 launches) with Montgomery-form butterflies: 540 MiB of 61-bit generic primes, forward,
    inverse and the three-pass product, against the Shoup path on the whole batch and the oracle on three elements."""
    import os
    import torch
    log_n, batch = 16, 360
    mods = GENERIC[:3]
    n, L = 1 << log_n, 3
    W = L * n
    d, o = pf.U64DcrtTable(log_n, mods), orc.U64DcrtTable(log_n, mods)
    os.environ["PFHE_DISABLE_MONT"] = "1"
    try:
        ds = pf.U64DcrtTable(log_n, mods)
    finally:
        del os.environ["PFHE_DISABLE_MONT"]
    a = _fill(pf, batch * W, mods, n, 21)
    bh = _fill(pf, W, mods, n, 22)
    f1, f2 = a.clone(), a.clone()
    d.transform_dev(f1); ds.transform_dev(f2)
    assert torch.equal(f1, f2)
    for e in (0, 177, 359):
        x = to_host(a[e * W:(e + 1) * W]).copy(); o.transform_slice(x)
        assert np.array_equal(to_host(f1[e * W:(e + 1) * W]), x)
    d.inverse_transform_dev(f1)
    assert torch.equal(f1, a)
    p1, p2 = a.clone(), a.clone()
    d.mul_dcrt_polynomial_dev(p1, bh); ds.mul_dcrt_polynomial_dev(p2, bh)
    assert torch.equal(p1, p2)


@pytest.mark.parametrize("moduli,shared", [(Q61[:1], True), (Q61[:2], False), (GENERIC[:1], True)])
def test_polymul_2p14_resident_workgroups(pf, orc, moduli, shared, monkeypatch):
    """N = 2^14 batches of at least two polynomials per CU take ntt_persist_mid_kernel (resident workgroups that prefetch
    their next polynomial): same words as one workgroup per polynomial — what a call on fewer polynomials than that runs,
    here the same batch in slices of 64 units — and the oracle on the first, a middle and the last element;
    pseudo-Mersenne and Montgomery arithmetic, ragged shares (601)."""
    import torch
    log_n, units = 14, 601
    n, L = 1 << log_n, len(moduli)
    W = L * n
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    assert d.transform_form(units * W)[0] == "ntt_persist_kernel<14,fwd>" and d.transform_form(64 * W)[0] == "ntt_block_kernel<14,fwd>"
    a = _fill(pf, units * W, moduli, n, 141)
    bh = _fill(pf, W if shared else units * W, moduli, n, 142)
    x, y = a.clone(), a.clone()
    d.mul_dcrt_polynomial_dev(x, bh)
    for u0 in range(0, units, 64):
        u1 = min(units, u0 + 64)
        d.mul_dcrt_polynomial_dev(y[u0 * W:u1 * W], bh if shared else bh[u0 * W:u1 * W])
    assert torch.equal(x, y)
    for e in (0, 300, units - 1):
        r = to_host(a[e * W:(e + 1) * W]).copy()
        o.transform_slice(r)
        o.mul_assign(r, to_host(bh[:W] if shared else bh[e * W:(e + 1) * W]).copy())
        o.inverse_transform_slice(r)
        assert np.array_equal(to_host(x[e * W:(e + 1) * W]), r), e
    # the transforms alone, both directions, same comparison
    f, f1 = a.clone(), a.clone()
    d.transform_dev(f)
    for u0 in range(0, units, 64):
        d.transform_dev(f1[u0 * W:min(units, u0 + 64) * W])
    assert torch.equal(f, f1)
    d.inverse_transform_dev(f)
    assert torch.equal(f, a)


@pytest.mark.parametrize("env,log_n,batch", [
    ({"PFHE_DISABLE_PIPELINED": "1"}, 16, 2),       # one launch per pass
    ({"PFHE_PIPELINED_MIN_MB": "1"}, 16, 4),        # the pipelined kernel on a 6 MiB batch
    ({"PFHE_PIPELINED_MIN_MB": "1", "PFHE_PIPE_TILES": "3"}, 16, 5),
])
@pytest.mark.parametrize("generic", [False, True])
def test_plan_tuning_switches_change_the_plan_not_the_words(pf, orc, env, log_n, batch, generic, monkeypatch):
    """The three switches NttTuning::from_env reads (at table creation: PFHE_DISABLE_PIPELINED, PFHE_PIPELINED_MIN_MB,
    PFHE_PIPE_TILES) against the oracle: forward, inverse, lazy forward and the polynomial product, pseudo-Mersenne and
    generic-prime (Montgomery) arithmetic.  The switches only choose how the stages are split over launches."""
    n = 1 << log_n
    moduli = Q61[:2]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if generic:
        monkeypatch.setenv("PFHE_DISABLE_PM", "1")
    d = pf.U64DcrtTable(log_n, moduli)
    o = orc.U64DcrtTable(log_n, moduli)
    rng = np.random.default_rng(77 + log_n + batch)
    a, bh = rand_rns(rng, moduli, n, batch), rand_rns(rng, moduli, n, 1)
    fwd = a.copy()
    o.transform_slice(fwd)
    x = to_dev(a)
    d.transform_dev(x)
    assert np.array_equal(to_host(x), fwd)
    d.inverse_transform_dev(x)
    assert np.array_equal(to_host(x), a)
    d.transform_dev(x, lazy=True)
    qs = np.repeat(np.tile(np.array(moduli, np.uint64), batch), n)
    lz = to_host(x)
    assert (lz < 4 * qs).all() and np.array_equal(lz % qs, fwd)
    exp = fwd.copy()
    W = len(moduli) * n
    for e in range(batch):
        o.mul_assign(exp[e * W:(e + 1) * W], bh)
    o.inverse_transform_slice(exp)
    y = to_dev(a)
    d.mul_dcrt_polynomial_dev(y, to_dev(bh))
    assert np.array_equal(to_host(y), exp)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("log_n", [12, 16])
def test_fused_polymul_contract_canonical_multiplicand_at_its_extremes(pf, orc, generic, log_n, monkeypatch):
    """pfhe_dcrt_mul_dcrt_polynomial_dev takes CANONICAL operands (include/pfhe.h; the reference's reduce_mul_slice_assign
    does, primus_reduce/src/slice_ops.rs:137-229).  Inside, the forward half hands RAW words (up to 2^63 + 3q) to one
    128 -> 64-bit reduction whose precondition is product < q * 2^64: the largest canonical multiplicand, q - 1 in every
    word, against data that drives the forward butterflies to their bounds (q - 1 in every word, and a spike) must still be
    exact, for pseudo-Mersenne and for Montgomery (generic-prime) tables.  A lazily transformed multiplicand ([0, 4q)) is
    outside the contract and documented as such; the documented route — transform with lazy = 0 — is what is used here."""
    if generic:
        monkeypatch.setenv("PFHE_DISABLE_PM", "1")
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    monkeypatch.delenv("PFHE_DISABLE_PM", raising=False)
    n, L = 1 << log_n, 3
    qs = np.repeat(np.array(Q61, np.uint64), n)
    top = qs - np.uint64(1)
    spike = np.zeros(L * n, np.uint64)
    spike[::n] = top[::n]
    rng = np.random.default_rng(3)
    rnd = rand_rns(rng, Q61, n, 1)
    for data in (top, spike, rnd):
        for mul_coeff in (top, rnd):
            mh = mul_coeff.copy()
            o.transform_slice(mh)                      # canonical NTT-domain multiplicand
            assert (mh < qs).all()
            mh_top = top.copy()                        # and the extreme canonical multiplicand itself
            for m in (mh, mh_top):
                ref = data.copy()
                o.transform_slice(ref)
                o.mul_assign(ref, m)
                o.inverse_transform_slice(ref)
                x = to_dev(data.copy())
                d.mul_dcrt_polynomial_dev(x, to_dev(m))
                assert np.array_equal(to_host(x), ref)
