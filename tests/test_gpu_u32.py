"""GPU parity of U32NttTable / U32DcrtTable (through the C ABI) against the oracle.

Mirrors primus_ntt/src/ntt/prime32/tests.rs: canonical outputs bit-exact, lazy outputs in the
documented range and equal mod q, round trips, monomials, table constants and constructor errors.
"""
import numpy as np
import pytest

from test_oracle_u32 import Q27, Q29, Q30, rand32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def to_dev32(a):
    import torch
    assert a.dtype == np.uint32
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def to_host32(t):
    return t.detach().cpu().numpy().view(np.uint32)


def test_errors(pf):
    q_big = next(q for q in range((1 << 30) + 1, (1 << 30) + (1 << 22), 2048)
                 if all(q % p for p in range(3, 33000, 2)))
    with pytest.raises(pf.PfheError) as e:
        pf.U32NttTable(10, q_big)
    assert e.value.kind == "ModulusTooLarge"
    with pytest.raises(pf.PfheError) as e:
        pf.U32NttTable(21, Q27)
    assert e.value.kind == "NoPrimitiveRoot"
    t = pf.U32NttTable(5, Q27)
    with pytest.raises(pf.PfheError) as e:
        t.transform_slice(np.zeros(33, np.uint32))
    assert e.value.kind == "BadLength"
    with pytest.raises(TypeError):
        t.transform_slice(np.zeros(32, np.uint64))


@pytest.mark.parametrize("log_n,q,batch", [
    (0, 17, 3), (1, 17, 5), (2, 17, 4), (3, Q27, 7), (4, Q27, 3), (5, Q27, 9), (6, Q30[0], 1), (7, Q27, 33),
    (8, Q29, 5), (9, Q27, 2), (10, Q27, 4), (11, Q30[1], 3), (12, Q30[2], 3), (13, Q30[0], 2), (14, Q30[0], 2),
    (15, Q30[1], 2), (16, Q30[0], 3), (17, Q27, 2), (18, Q27, 1),
])
def test_u32_ntt_matches_oracle(pf, orc, log_n, q, batch):
    rng = np.random.default_rng(log_n * 13 + batch)
    n = 1 << log_n
    t, o = pf.U32NttTable(log_n, q), orc.U32NttTable(log_n, q)
    assert (t.poly_length(), t.log_n(), t.modulus(), t.root(), t.inv_root(), t.inv_n()) == \
        (n, log_n, q, o.root, o.inv_root, o.inv_n)
    a = rand32(rng, q, n * batch)
    a[:min(n, 4)] = [0, q - 1, 1, q // 2][:min(n, 4)]
    ref = a.copy(); o.transform_slice(ref)
    got = a.copy(); t.transform_slice(got)
    assert np.array_equal(got, ref)
    t.inverse_transform_slice(got)
    assert np.array_equal(got, a)
    iref = a.copy(); o.inverse_transform_slice(iref)
    igot = a.copy(); t.inverse_transform_slice(igot)
    assert np.array_equal(igot, iref)
    if log_n == 0:
        return
    # lazy variants: the reference's exact Barrett-32 arithmetic, hence identical lazy words
    lz = a.copy(); t.lazy_transform_slice(lz)
    lzo = a.copy(); o.lazy_transform_slice(lzo)
    assert lz.max() < 4 * q and np.array_equal(lz, lzo)
    lzi = ref.copy(); t.lazy_inverse_transform_slice(lzi)
    lzio = ref.copy(); o.lazy_inverse_transform_slice(lzio)
    assert lzi.max() < 2 * q and np.array_equal(lzi, lzio)


@pytest.mark.parametrize("log_n", [4, 5, 10, 13, 16])
def test_lazy_inputs_up_to_4q_and_2q(pf, orc, log_n):
    """prime32/tests.rs:13-112."""
    q = Q27 if log_n <= 20 and (Q27 - 1) % (2 << log_n) == 0 else Q30[0]
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    t, o = pf.U32NttTable(log_n, q), orc.U32NttTable(log_n, q)
    a = rand32(rng, 4 * q, 2 * n)
    got, ref = a.copy(), a.copy()
    t.lazy_transform_slice(got); o.lazy_transform_slice(ref)
    assert got.max() < 4 * q and np.array_equal(got, ref)
    b = rand32(rng, 2 * q, 2 * n)
    got, ref = b.copy(), b.copy()
    t.lazy_inverse_transform_slice(got); o.lazy_inverse_transform_slice(ref)
    assert got.max() < 2 * q and np.array_equal(got, ref)


@pytest.mark.parametrize("log_n,batch", [(3, 5), (6, 3), (10, 3), (12, 2), (16, 2)])
def test_u32_dcrt_and_polymul(pf, orc, log_n, batch):
    rng = np.random.default_rng(log_n)
    n, L = 1 << log_n, 3
    t, o = pf.U32DcrtTable(log_n, Q30), orc.U32DcrtTable(log_n, Q30)
    assert (t.poly_length(), t.moduli_count(), t.crt_poly_length(), t.moduli()) == (n, L, L * n, Q30)
    assert t.roots() == [x.root for x in o.tables]
    a = np.concatenate([rand32(rng, q, n) for _ in range(batch) for q in Q30])
    b = np.concatenate([rand32(rng, q, n) for _ in range(batch) for q in Q30])
    ref = a.copy(); o.transform_slice(ref)
    got = a.copy(); t.transform_slice(got)
    assert np.array_equal(got, ref)
    # device path: polynomial product, elementwise and with one shared multiplicand
    da, db = to_dev32(a), to_dev32(b)
    t.transform_dev(da); t.transform_dev(db)
    fb = b.copy(); o.transform_slice(fb)
    exp = ref.copy(); o.mul_assign(exp, fb)
    acc0 = np.concatenate([rand32(rng, q, n) for _ in range(batch) for q in Q30])
    dacc = to_dev32(acc0)
    t.add_mul_assign_dev(dacc, da, db)
    eacc = acc0.copy(); o.add_mul_assign(eacc, ref, fb)
    assert np.array_equal(to_host32(dacc), eacc)
    t.mul_assign_dev(da, db)
    assert np.array_equal(to_host32(da), exp)
    t.inverse_transform_dev(da)
    o.inverse_transform_slice(exp)
    assert np.array_equal(to_host32(da), exp)
    shared = to_dev32(fb[:L * n].copy())
    dc = to_dev32(ref)
    t.mul_assign_dev(dc, shared)
    exp2 = ref.copy(); o.mul_assign(exp2, fb[:L * n].copy())
    assert np.array_equal(to_host32(dc), exp2)


@pytest.mark.parametrize("log_n", [0, 1, 4, 10, 16])
def test_u32_monomials(pf, orc, log_n):
    q = Q30[0] if log_n > 10 else Q27
    n = 1 << log_n
    t, o = pf.U32NttTable(log_n, q), orc.U32NttTable(log_n, q)
    out = np.empty(n, np.uint32)
    for coeff, degree in [(0, 3), (5, 0), (1, 1), (q - 1, n - 1), (12345, n // 2), (777, 2 * n + 1)]:
        t.transform_monomial(coeff, degree % n if degree < 2 * n else degree % n, out)
        assert np.array_equal(out, o.transform_monomial(coeff, degree % n))
    for degree in (0, 1 % n, n - 1):
        t.transform_coeff_one_monomial(degree, out)
        assert np.array_equal(out, o.transform_coeff_one_monomial(degree))
        t.transform_coeff_minus_one_monomial(degree, out)
        assert np.array_equal(out, o.transform_coeff_minus_one_monomial(degree))
    d = pf.U32DcrtTable(log_n, [q]) if log_n > 10 else pf.U32DcrtTable(log_n, [Q27, Q29] if log_n <= 10 else [q])
    od = orc.U32DcrtTable(log_n, d.moduli())
    outd = np.empty(d.crt_poly_length(), np.uint32)
    d.transform_coeff_minus_one_monomial(n // 2, outd)
    assert np.array_equal(outd, od.transform_coeff_minus_one_monomial(n // 2))
    d.transform_monomial(3, 1 % n, outd)
    assert np.array_equal(outd, od.transform_monomial(3, 1 % n))
    with pytest.raises(pf.PfheError):
        t.transform_monomial(q, 1, out)


def test_u32_full_batch_round_trip_properties(pf, orc, monkeypatch):
    """N = 2^16, 3 primes, 1024 RNS polynomials (768 MiB): round trip + oracle spot checks; the transform switches
    that 64-bit tables react to at this size must not change a u32 table's results."""
    import torch
    log_n, batch = 16, 1024
    n, L = 1 << log_n, 3
    t, o = pf.U32DcrtTable(log_n, Q30), orc.U32DcrtTable(log_n, Q30)
    monkeypatch.setenv("PFHE_DISABLE_PIPELINED", "1")
    t_plain = pf.U32DcrtTable(log_n, Q30)
    monkeypatch.delenv("PFHE_DISABLE_PIPELINED")
    x = torch.empty(batch * L * n, dtype=torch.int32, device="cuda")
    t.fill_uniform_dev(x, 0x5EED_0000_0000_0032)
    orig = x.clone()
    t.transform_dev(x)
    y = orig.clone()
    t_plain.transform_dev(y)
    assert torch.equal(x, y)
    t_plain.inverse_transform_dev(y)
    assert torch.equal(y, orig)
    del y
    for e in (0, 511, 1023):
        s = slice(e * L * n, (e + 1) * L * n)
        ref = to_host32(orig[s]).copy(); o.transform_slice(ref)
        assert np.array_equal(to_host32(x[s]), ref)
    t.inverse_transform_dev(x)
    assert torch.equal(x, orig)


@pytest.mark.parametrize("log_n", [5, 11, 16])
def test_u32_and_u64_tables_agree_on_the_gpu(pf, log_n):
    """The same prime through the two independent device paths (packed Barrett-32 words vs 64-bit Shoup)."""
    q = Q27
    rng = np.random.default_rng(log_n)
    a = rand32(rng, q, 3 << log_n)
    t32, t64 = pf.U32NttTable(log_n, q), pf.U64NttTable(log_n, q)
    assert t32.root() == t64.root()
    x32, x64 = a.copy(), a.astype(np.uint64)
    t32.transform_slice(x32); t64.transform_slice(x64)
    assert np.array_equal(x32.astype(np.uint64), x64)
    t32.inverse_transform_slice(x32); t64.inverse_transform_slice(x64)
    assert np.array_equal(x32, a) and np.array_equal(x64, a.astype(np.uint64))


@pytest.mark.parametrize("batch,tiles", [(1400, 0), (1371, 5)])
def test_u32_pipelined_inverse_equals_plain_passes(pf, orc, batch, tiles, monkeypatch):
    """Round 5: from 1 GiB of u32 data the N = 2^16 transforms take the pipelined form (ntt_pipe_{fwd,inv}_kernel<B32Arith,
    11>, tiles of 512 MiB).  Ragged batches and tile counts: bit-identical to the two plain launches in both directions,
    oracle on the polynomials either side of the tile boundaries."""
    import torch
    log_n = 16
    n, L = 1 << log_n, 3
    if tiles:
        monkeypatch.setenv("PFHE_PIPE_TILES", str(tiles))
    t = pf.U32DcrtTable(log_n, Q30)
    monkeypatch.setenv("PFHE_DISABLE_PIPELINED", "1")
    t_plain = pf.U32DcrtTable(log_n, Q30)
    monkeypatch.delenv("PFHE_DISABLE_PIPELINED")
    o = orc.U32DcrtTable(log_n, Q30)
    x = torch.empty(batch * L * n, dtype=torch.int32, device="cuda")
    t.fill_uniform_dev(x, 0x5EED_0000_0000_0033)
    orig = x.clone()
    y = orig.clone()
    t_plain.transform_dev(y)
    z = y.clone()
    t.transform_dev(x)                           # forward: pipelined by default
    assert torch.equal(x, y)
    for e in sorted({0, 1, batch // 5, batch // 2 - 1, batch // 2, batch - 1}):
        s = slice(e * L * n, (e + 1) * L * n)
        ref = to_host32(orig[s]).copy(); o.transform_slice(ref)
        assert np.array_equal(to_host32(x[s]), ref), e
    t.inverse_transform_dev(x)                   # inverse: pipelined by default
    assert torch.equal(x, orig)
    t_plain.inverse_transform_dev(z)
    assert torch.equal(z, orig)
    lz = y.clone()
    t.inverse_transform_dev(lz, lazy=True)       # lazy inverse through the pipelined kernels: [0, 2q), same residues
    h = to_host32(lz[:L * n]).astype(np.uint64)
    qs = np.repeat(np.array(Q30, np.uint64), n)
    assert (h < 2 * qs).all() and np.array_equal(h % qs, to_host32(orig[:L * n]).astype(np.uint64))
