"""Seeded randomized differential tests: HIP path (through the C ABI) vs the oracle over random
NTT-friendly primes of 20..62 bits (both arithmetic policies get exercised: most random primes do
not have the pseudo-Mersenne shape), 1..6 RNS limbs, ragged batch sizes and every transform plan."""
import numpy as np
import pytest

from gpu_util import to_dev, to_host

pytestmark = pytest.mark.gpu


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def random_ntt_primes(rng, count, bits, log_n):
    """`count` distinct primes of exactly `bits` bits with 2^(log_n+1) | q - 1."""
    step = 1 << (log_n + 1)
    lo, hi = (1 << (bits - 1)) // step + 1, (1 << bits) // step
    assert hi - lo >= 64 * count, "not enough candidates: raise `bits`"
    out = []
    for _ in range(200000):
        k = int(rng.integers(lo, hi))
        q = k * step + 1
        if q.bit_length() == bits and q not in out and is_prime(q):
            out.append(q)
            if len(out) == count:
                return out
    raise AssertionError("prime search exhausted")


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


import os

SCALE = int(os.environ.get("PFHE_FUZZ_SCALE", "1"))  # PFHE_FUZZ_SCALE=10 widens every sweep tenfold
CASES = list(range(60 * SCALE))


@pytest.mark.parametrize("case", CASES)
def test_random_u64_tables(pf, orc, case):
    rng = np.random.default_rng(1000 + case)
    log_n = int(rng.integers(1, 15)) if case % 5 else int(rng.integers(15, 18))
    bits = int(rng.integers(max(log_n + 12, 20), 63))
    L = int(rng.integers(1, 7))
    batch = int(rng.integers(1, 6)) if log_n > 10 else int(rng.integers(1, 70))
    moduli = random_ntt_primes(rng, L, bits, log_n)
    n = 1 << log_n
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    assert d.roots() == [o.table(i).root for i in range(L)]
    a = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for _ in range(batch) for q in moduli])
    b = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for _ in range(batch) for q in moduli])
    ref = a.copy(); o.transform_slice(ref)
    da = to_dev(a)
    d.transform_dev(da)
    assert np.array_equal(to_host(da), ref), (log_n, moduli, batch)
    # polynomial product through the fused path, per-element multiplicand
    fb = b.copy(); o.transform_slice(fb)
    exp = ref.copy()
    W = L * n
    for e in range(batch):
        o.mul_assign(exp[e * W:(e + 1) * W], fb[e * W:(e + 1) * W])
    o.inverse_transform_slice(exp)
    dprod = to_dev(a)
    d.mul_dcrt_polynomial_dev(dprod, to_dev(fb))
    assert np.array_equal(to_host(dprod), exp)
    # lazy forward on [0,4q) inputs agrees mod q and stays in range
    lz = np.concatenate([rng.integers(0, 4 * q, n, dtype=np.uint64) for _ in range(batch) for q in moduli])
    can = lz.copy()
    qs = np.repeat(np.tile(np.array(moduli, np.uint64), batch), n)
    can %= qs
    o.transform_slice(can)
    dl = to_dev(lz)
    d.transform_dev(dl, lazy=True)
    got = to_host(dl)
    assert (got < 4 * qs).all() and np.array_equal(got % qs, can)


@pytest.mark.parametrize("case", range(30 * SCALE))
def test_random_u32_tables(pf, orc, case):
    rng = np.random.default_rng(2000 + case)
    log_n = int(rng.integers(1, 15)) if case % 5 else int(rng.integers(15, 18))
    bits = int(rng.integers(min(max(log_n + 12, 14), 30), 31))
    L = int(rng.integers(1, 5))
    batch = int(rng.integers(1, 6)) if log_n > 10 else int(rng.integers(1, 70))
    moduli = random_ntt_primes(rng, L, bits, log_n)
    n = 1 << log_n
    d, o = pf.U32DcrtTable(log_n, moduli), orc.U32DcrtTable(log_n, moduli)
    a = np.concatenate([rng.integers(0, q, n, dtype=np.uint64).astype(np.uint32) for _ in range(batch) for q in moduli])
    ref = a.copy(); o.transform_slice(ref)
    got = a.copy(); d.transform_slice(got)
    assert np.array_equal(got, ref), (log_n, moduli, batch)
    d.inverse_transform_slice(got)
    assert np.array_equal(got, a)
    lz = a.copy(); d.lazy_transform_slice(lz)
    lzo = a.copy(); o.lazy_transform_slice(lzo)
    assert np.array_equal(lz, lzo)


@pytest.mark.parametrize("case", range(16 * SCALE))
def test_random_external_products(pf, orc, case):
    rng = np.random.default_rng(3000 + case)
    log_n = int(rng.integers(3, 13)) if case % 4 else 16
    L = int(rng.integers(1, 4))
    bits = int(rng.integers(40, 62))
    k = int(rng.integers(1, 3)) if log_n < 16 else 1
    moduli = random_ntt_primes(rng, L, bits, log_n)
    total_bits = sum(q.bit_length() for q in moduli)
    log_basis = int(rng.integers(4, min(31, min(moduli).bit_length() - 1)))
    ell_full = 1
    Q = 1
    for q in moduli:
        Q *= q
    ell_full = Q.bit_length() // log_basis
    rev = None if case % 3 else int(rng.integers(1, ell_full + 1))
    batch = int(rng.integers(1, 4))
    shared = bool(case % 2)
    n = 1 << log_n
    ot, ob = orc.U64DcrtTable(log_n, moduli), orc.RNSBase(moduli)
    obasis = orc.BigUintApproxSignedBasis(ob, log_basis, rev)
    ell = obasis.decompose_length
    W = (k + 1) * L * n
    glwe = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for _ in range(batch * (k + 1)) for q in moduli])
    ggsw = np.concatenate([rng.integers(0, q, n, dtype=np.uint64)
                           for _ in range((1 if shared else batch) * (k + 1) * ell * (k + 1)) for q in moduli])
    G = (k + 1) * ell * W
    exp = np.concatenate([orc.mul_dcrt_ggsw_to(ot, ob, obasis, k, glwe[e * W:(e + 1) * W].copy(),
                                               ggsw[:G] if shared else ggsw[e * G:(e + 1) * G])
                          for e in range(batch)])
    t, base = pf.U64DcrtTable(log_n, moduli), pf.RNSBase(moduli)
    ctx = pf.DcrtGlevContext(t, base, pf.BigUintApproxSignedBasis(base, log_basis, rev), k, int(rng.integers(0, 3)))
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
    assert np.array_equal(out, exp), (log_n, moduli, k, log_basis, rev, batch, shared, total_bits)


def pm_prime(K: int, log_n: int, largest_c: bool):
    """A prime 2^K - c with 2^(log_n+1) | q - 1 and c < 2^(K-33): the largest (or smallest) admissible c."""
    step = 1 << (log_n + 1)
    jmax = ((1 << (K - 33)) // step)
    js = range(jmax, 0, -1) if largest_c else range(1, jmax + 1)
    for j in js:
        c = j * step - 1
        if c < (1 << (K - 33)) and is_prime((1 << K) - c):
            return (1 << K) - c
    return None


@pytest.mark.parametrize("K", [40, 41, 44, 48, 52, 56, 59, 60, 61])
@pytest.mark.parametrize("largest_c", [True, False])
def test_pseudo_mersenne_bounds(pf, orc, K, largest_c):
    """The pseudo-Mersenne policy at the edges of its admissible shape (K = 40..61, c just below
    2^(K-33)) with extreme inputs (q-1 everywhere; 4q-1 for the lazy forward transform)."""
    log_n = min(12, K - 36)
    q = pm_prime(K, log_n, largest_c)
    if q is None:
        pytest.skip("no admissible prime")
    n = 1 << log_n
    t, o = pf.U64NttTable(log_n, q), orc.U64NttTable(log_n, q)
    rng = np.random.default_rng(K)
    for a in (np.full(2 * n, q - 1, np.uint64), rng.integers(0, q, 2 * n, dtype=np.uint64)):
        ref = a.copy(); o.transform_slice(ref)
        d = to_dev(a)
        t.transform_dev(d)
        assert np.array_equal(to_host(d), ref), (K, q)
        t.inverse_transform_dev(d)
        assert np.array_equal(to_host(d), a)
        iref = a.copy(); o.inverse_transform_slice(iref)
        d = to_dev(a); t.inverse_transform_dev(d)
        assert np.array_equal(to_host(d), iref)
    lz = np.full(n, 4 * q - 1, np.uint64)
    can = (lz % np.uint64(q)).copy(); o.transform_slice(can)
    d = to_dev(lz); t.transform_dev(d, lazy=True)
    got = to_host(d)
    assert got.max() < 4 * q and np.array_equal(got % np.uint64(q), can)
    # point-wise products take the folding multiply for these primes: extreme and random operands
    dt, ot = pf.U64DcrtTable(log_n, [q]), orc.U64DcrtTable(log_n, [q])
    for x, y, z in ((np.full(n, q - 1, np.uint64),) * 3, tuple(rng.integers(0, q, n, dtype=np.uint64) for _ in range(3))):
        e1 = x.copy(); ot.mul_assign(e1, y)
        d1 = to_dev(x); dt.mul_assign_dev(d1, to_dev(y)); assert np.array_equal(to_host(d1), e1)
        e2 = [(int(u) * int(v) + int(w)) % q for u, v, w in zip(x, y, z)]
        d2 = to_dev(z); dt.add_mul_assign_dev(d2, to_dev(x), to_dev(y)); assert to_host(d2).tolist() == e2
    li = np.full(n, 2 * q - 1, np.uint64)
    cani = (li % np.uint64(q)).copy(); o.inverse_transform_slice(cani)
    d = to_dev(li); t.inverse_transform_dev(d, lazy=True)
    got = to_host(d)
    assert got.max() < 2 * q and np.array_equal(got % np.uint64(q), cani)


# ---- round 6: the <u32> RNS / gadget / conversion / product layer and bases of up to 32 moduli, randomized ----
def random_coprime_moduli(rng, count, max_bits):
    """`count` pairwise-coprime odd moduli (not necessarily prime) of random sizes in [3, 2^max_bits)."""
    from math import gcd
    out = []
    while len(out) < count:
        bits = int(rng.integers(max(2, max_bits - 12), max_bits + 1))
        c = int(rng.integers(1 << (bits - 1), 1 << bits)) | 1
        if c > 2 and all(gcd(c, m) == 1 for m in out):
            out.append(c)
    return out


def _family(pf, orc, word_bits):
    if word_bits == 32:
        return (pf.RNSBase32, pf.BigUintApproxSignedBasis32, pf.BaseConverter32, orc.RNSBase32, orc.BigUintApproxSignedBasis32,
                orc.BaseConverter32, np.uint32)
    return pf.RNSBase, pf.BigUintApproxSignedBasis, pf.BaseConverter, orc.RNSBase, orc.BigUintApproxSignedBasis, orc.BaseConverter, np.uint64


@pytest.mark.parametrize("word_bits", [32, 64])
@pytest.mark.parametrize("case", range(24 * SCALE))
def test_random_bases_compose_digits_and_conversion(pf, orc, case, word_bits):
    """Random base width (1..32 moduli, every constants form and limb-count template), random composite moduli, random
    log_basis and reverse_length: compose, out-of-place carry init, every level's unsigned and signed digits, decompose
    back, and a conversion into a second random base — each against the oracle on the same words."""
    rng = np.random.default_rng(7000 + case + 1000 * word_bits)
    RNS, Basis, Conv, oRNS, oBasis, oConv, dt = _family(pf, orc, word_bits)
    L = int(rng.integers(1, 33)) if case % 3 else int(rng.integers(1, 9))
    moduli = random_coprime_moduli(rng, L, word_bits - 2)
    base, obase = RNS(moduli), oRNS(moduli)
    vl = base.big_uint_value_len()
    assert vl == obase.value_len
    count = int(rng.integers(1, 700))
    res = np.concatenate([rng.integers(0, q, count, dtype=np.uint64).astype(dt) for q in moduli])
    big = np.empty(count * vl, dt)
    base.compose_multiple_values_to(res, big, count)
    obig = obase.compose_multiple_values_to(res, count)
    assert np.array_equal(big, obig), (moduli, count)
    back = np.empty_like(res)
    base.decompose_big_uint_values_to(big, back, count)
    assert np.array_equal(back, res)
    Q = 1
    for q in moduli:
        Q *= q
    log_basis = int(rng.integers(1, min(word_bits, Q.bit_length() + 1)))
    ell_full = max(1, Q.bit_length() // log_basis)
    rev = None if case % 2 else int(rng.integers(1, ell_full + 1))
    try:
        obasis = oBasis(obase, log_basis, rev)
    except orc.OracleError:
        with pytest.raises(pf.PfheError):
            Basis(base, log_basis, rev)
        return
    basis = Basis(base, log_basis, rev)
    assert (basis.decompose_length(), basis.drop_bits()) == (obasis.decompose_length, obasis.drop_bits), (moduli, log_basis, rev)
    adj, car = np.empty_like(big), np.zeros(count, np.uint8)
    basis.init_value_carry_slice_to(big, adj, car)
    oadj, ocar = obasis.init_value_carry_slice_to(obig.copy(), count)
    assert np.array_equal(adj, oadj) and np.array_equal(car, ocar), (moduli, log_basis, rev)
    car2, ocar2 = car.copy(), ocar.copy()
    for level in range(basis.decompose_length()):
        dig = np.empty(count, dt)
        basis.unsigned_decompose_slice_to(level, adj, dig, car)
        assert np.array_equal(dig, obasis.unsigned_decompose_slice_to(level, oadj, ocar, count)), (moduli, log_basis, rev, level)
        assert np.array_equal(car, ocar)
        sig = np.empty_like(adj)
        basis.decompose_slice_to(level, adj, sig, car2)
        assert np.array_equal(sig, obasis.decompose_slice_to(level, oadj, ocar2, count)), (moduli, log_basis, rev, level)
    # conversion into a second random base that is coprime to nothing in particular (BaseConverter::new takes any two bases)
    Lo = int(rng.integers(1, 9)) if case % 4 else int(rng.integers(9, 33))
    mod_out = random_coprime_moduli(rng, Lo, word_bits - 2)
    conv, oconv = Conv(base, RNS(mod_out)), oConv(obase, oRNS(mod_out))
    out = np.empty(Lo * count, dt)
    conv.fast_convert_array(res, out, count)
    assert np.array_equal(out, oconv.fast_convert_array(res, count)), (moduli, mod_out)
    e, oe = Conv(base, RNS(mod_out[:1])), oConv(obase, oRNS(mod_out[:1]))
    eo = np.empty(count, dt)
    e.exact_convert_array(res, eo, count)
    assert np.array_equal(eo, oe.exact_convert_array(res, count)), (moduli, mod_out[0])


@pytest.mark.parametrize("case", range(16 * SCALE))
def test_random_u32_external_products(pf, orc, case):
    from test_gpu_rns32 import make_case32
    rng = np.random.default_rng(8000 + case)
    log_n = int(rng.integers(3, 13)) if case % 4 else 16
    L = int(rng.integers(1, 5)) if case % 5 else int(rng.integers(5, 12))
    bits = int(rng.integers(max(log_n + 12, 20), 31))
    k = int(rng.integers(1, 3)) if log_n < 16 else 1
    moduli = random_ntt_primes(rng, L, bits, log_n)
    Q = 1
    for q in moduli:
        Q *= q
    log_basis = int(rng.integers(4, min(31, min(moduli).bit_length() - 1)))   # the lift needs B below every modulus
    ell_full = max(1, Q.bit_length() // log_basis)
    rev = None if case % 3 else int(rng.integers(1, min(ell_full, 8) + 1))
    if rev is None and ell_full > 10:
        rev = 6                      # keep the GGSW of a wide base small
    batch = int(rng.integers(1, 6)) if log_n == 16 else int(rng.integers(1, 4))
    shared = bool(case % 2)
    otable, obase, obasis, glwe, ggsw, exp = make_case32(orc, rng, log_n, k, moduli, log_basis, rev, batch, shared)
    t, base = pf.U32DcrtTable(log_n, moduli), pf.RNSBase32(moduli)
    ctx = pf.DcrtGlevContext32(t, base, pf.BigUintApproxSignedBasis32(base, log_basis, rev), k, int(rng.integers(0, 3)))
    out = np.empty_like(glwe)
    pf.mul_dcrt_ggsw_to(glwe, ggsw, out, ctx)
    assert np.array_equal(out, exp), (log_n, moduli, k, log_basis, rev, batch, shared)
