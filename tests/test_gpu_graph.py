"""The *_dev entry points can be captured into a HIP graph and replayed (launch-bound inner loops of a
caller, e.g. a bootstrapping loop of small external products): they allocate nothing per call, and the
internal fork/join over the plan's (or the transform's pooled) streams is expressed with events, which
stream capture follows."""
import numpy as np
import pytest

from gpu_util import rand_rns, to_dev
from pyref import Q61

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


@pytest.mark.parametrize("chunk,batch", [(0, 2), (1, 3)])  # one chunk (caller's stream) / pipelined chunks
def test_external_product_and_ntt_in_a_graph(pf, chunk, batch):
    import torch
    log_n, k = 12, 1
    n, L = 1 << log_n, 3
    rng = np.random.default_rng(chunk)
    t, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(t, base, basis, k, chunk)
    ell = basis.decompose_length()
    dg = to_dev(rand_rns(rng, Q61, n, batch * (k + 1)))
    dk = to_dev(rand_rns(rng, Q61, n, (k + 1) * ell * (k + 1)))
    out = torch.zeros_like(dg)
    s = torch.cuda.Stream()

    def work():
        pf.mul_dcrt_ggsw_to_dev(dg, dk, out, ctx, into_coeff_form=True, stream=s)
        t.transform_dev(out, stream=s)

    with torch.cuda.stream(s):
        work()  # eager reference (also creates every lazily-created stream / event outside the capture)
    s.synchronize()
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        work()
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("pipelined", [True, False])
def test_large_batch_transform_in_a_graph(pf, pipelined, monkeypatch):
    """>= 256 MiB: the pipelined form launches its tiles on the capturing stream itself (captured as is); without it
    the transform is two plain launches on that stream."""
    import ctypes as C
    import torch
    from primus_fhe_amd._lib import check, u64p
    log_n, batch = 16, 352
    n, L = 1 << log_n, 3
    if not pipelined:
        monkeypatch.setenv("PFHE_DISABLE_PIPELINED", "1")  # switches are read when a table is created
    t = pf.U64DcrtTable(log_n, Q61)
    mods = np.array(Q61, np.uint64)
    x = torch.empty(batch * L * n, dtype=torch.int64, device="cuda")
    check(pf.lib().pfhe_fill_uniform_dev(0, C.c_void_p(x.data_ptr()), x.numel(), mods.ctypes.data_as(u64p), L, n, 3, None))
    orig = x.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t.transform_dev(x, stream=s)
    s.synchronize()
    ref = x.clone()
    x.copy_(orig)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        t.transform_dev(x, stream=s)
        t.inverse_transform_dev(x, stream=s)
        t.transform_dev(x, stream=s)
    x.copy_(orig)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, ref)


def test_cmux_style_step_in_a_graph(pf):
    """acc += ggsw [.] (acc * X^r - acc): monomial rotation, subtraction, external product and addition — the
    element-wise kernels and the product captured together and replayed (a blind-rotation step)."""
    import torch
    log_n, k, r = 11, 1, 777
    n = 1 << log_n
    rng = np.random.default_rng(5)
    t, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(t, base, basis, k)
    ell = basis.decompose_length()
    acc0 = to_dev(rand_rns(rng, Q61, n, k + 1))
    dk = to_dev(rand_rns(rng, Q61, n, (k + 1) * ell * (k + 1)))
    acc, diff, prod = acc0.clone(), torch.empty_like(acc0), torch.empty_like(acc0)
    s = torch.cuda.Stream()

    def step():
        t.mul_monomial_to_dev(acc, r, diff, stream=s)
        t.sub_to_dev(diff, acc, diff, stream=s)
        pf.mul_dcrt_ggsw_to_dev(diff, dk, prod, ctx, into_coeff_form=True, stream=s)
        t.add_to_dev(acc, prod, acc, stream=s)

    with torch.cuda.stream(s):
        step()
        step()
    s.synchronize()
    ref = acc.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    # beyond N = 2^14 the in-place rotation needs a scratch tile: it refuses to be captured, and must not poison
    # the capture machinery afterwards (at N = 2^11 it runs in registers and is capturable like the rest)
    big = pf.U64DcrtTable(15, Q61[:1])
    xb = torch.zeros(1 << 15, dtype=torch.int64, device="cuda")
    with pytest.raises(pf.PfheError):
        with torch.cuda.graph(torch.cuda.CUDAGraph(), stream=s):
            big.mul_monomial_assign_dev(xb, 1, stream=s)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        t.mul_monomial_assign_dev(diff, 5, stream=s)
        t.mul_monomial_assign_dev(diff, 2 * n - 5, stream=s)
    before = diff.clone()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(diff, before)  # X^5 * X^(2N-5) = 1
    acc.copy_(acc0)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(acc, ref)


def test_monomial_transforms_in_a_graph(pf):
    """transform_monomial_dev & co. are ONE launch on the caller's stream (the per-limb coefficient and its Shoup
    quotient travel as kernel arguments): no allocation, copy or synchronisation, hence capturable — a CMUX loop
    builds X^d in NTT form every step."""
    import torch
    log_n = 11
    n = 1 << log_n
    t, t32 = pf.U64NttTable(log_n, Q61[0]), pf.U32NttTable(log_n, 132120577)
    out = torch.zeros(n, dtype=torch.int64, device="cuda")
    out32 = torch.zeros(n, dtype=torch.int32, device="cuda")
    exp = np.zeros(n, np.uint64)
    t.transform_monomial(7, 321, exp)  # host form of the same transform
    exp32 = np.zeros(n, np.uint32)
    t32.transform_monomial(7, 321, exp32)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        t.transform_monomial_dev(7, 321, out, stream=s)
        t32.transform_monomial_dev(7, 321, out32, stream=s)
    out.zero_()
    out32.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), exp)
    assert np.array_equal(out32.cpu().numpy().view(np.uint32), exp32)
    # DcrtTable forms (dcrt/mod.rs:105-134), three limbs, c * X^d / X^d / -X^d
    d = pf.U64DcrtTable(log_n, Q61)
    outs = [torch.zeros(3 * n, dtype=torch.int64, device="cuda") for _ in range(3)]
    exps = [np.zeros(3 * n, np.uint64) for _ in range(3)]
    d.transform_monomial(9, 2 * n - 5, exps[0])
    d.transform_coeff_one_monomial(77, exps[1])
    d.transform_coeff_minus_one_monomial(n + 1, exps[2])
    gd = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gd, stream=s):
        d.transform_monomial_dev(9, 2 * n - 5, outs[0], stream=s)
        d.transform_coeff_one_monomial_dev(77, outs[1], stream=s)
        d.transform_coeff_minus_one_monomial_dev(n + 1, outs[2], stream=s)
    for o in outs:
        o.zero_()
    gd.replay()
    torch.cuda.synchronize()
    for o, e in zip(outs, exps):
        assert np.array_equal(o.cpu().numpy().view(np.uint64), e)


def test_round3_kernels_first_called_inside_a_capture(pf):
    """The persistent N = 2^14 kernel (sets its dynamic-LDS attribute and reads the CU count on first use), the three-pass
    product's middle kernel and the external product's fused inverse tail, each called for the FIRST time inside a
    capture (fresh primes, so no earlier test has configured these instantiations): capture must succeed and the replay must
    equal the eager result."""
    import torch
    rng = np.random.default_rng(31)
    s = torch.cuda.Stream()
    # --- N = 2^14: 600 polynomials >= two per resident workgroup on a 256-CU part: ntt_persist_kernel, both directions
    q14 = Q61[1]
    t14 = pf.U64NttTable(14, q14)
    x = to_dev(rng.integers(0, q14, 600 << 14, dtype=np.uint64))
    orig = x.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        t14.transform_dev(x, stream=s)
        t14.inverse_transform_dev(x, stream=s)
    x.copy_(orig + 0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, orig)  # forward then inverse = identity
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        t14.transform_dev(x, stream=s)
    g2.replay()
    torch.cuda.synchronize()
    y = orig.clone()
    t14.transform_dev(y)
    assert torch.equal(x, y)
    # --- N = 2^13 (one launch of the middle kernel) and N = 2^16 (strided, middle, strided) products
    for log_n, batch in ((13, 5), (16, 3)):
        n = 1 << log_n
        d = pf.U64DcrtTable(log_n, Q61[:2])
        a = to_dev(rand_rns(rng, Q61[:2], n, batch))
        bh = to_dev(rand_rns(rng, Q61[:2], n, 1))
        keep = a.clone()
        gp = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gp, stream=s):
            d.mul_dcrt_polynomial_dev(a, bh, stream=s)
        a.copy_(keep)
        gp.replay()
        torch.cuda.synchronize()
        ref = keep.clone()
        d.mul_dcrt_polynomial_dev(ref, bh)
        assert torch.equal(a, ref)
    # --- external product at N = 2^16, coefficient form (fused inverse tail + one strided pass), batch 8
    log_n, k, batch = 16, 1, 8
    n = 1 << log_n
    t, base = pf.U64DcrtTable(log_n, Q61), pf.RNSBase(Q61)
    basis = pf.BigUintApproxSignedBasis(base, 30)
    ctx = pf.DcrtGlevContext(t, base, basis, k)
    dg = to_dev(rand_rns(rng, Q61, n, batch * (k + 1)))
    dk = to_dev(rand_rns(rng, Q61, n, (k + 1) * basis.decompose_length() * (k + 1)))
    out = torch.zeros_like(dg)
    ge = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ge, stream=s):
        pf.mul_dcrt_ggsw_to_dev(dg, dk, out, ctx, into_coeff_form=True, stream=s)
    out.zero_()
    ge.replay()
    torch.cuda.synchronize()
    ref = torch.zeros_like(dg)
    pf.mul_dcrt_ggsw_to_dev(dg, dk, ref, ctx, into_coeff_form=True)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("log_n,batch", [(11, 3), (16, 4)])   # separate kernels / the fused kernels of N = 2^16
def test_u32_external_product_in_a_graph(pf, log_n, batch):
    """The <u32> product (pfhe_extprod32_mul_dcrt_ggsw_to_dev) and the u32 transform, captured and replayed: same words as
    the eager run, for the shape that takes the separate kernels and for the one that takes the fused ones."""
    import torch
    from test_gpu_u32 import to_dev32
    from test_oracle_rns32 import Q30, rand32
    k, n = 1, 1 << log_n
    rng = np.random.default_rng(log_n)
    t, base = pf.U32DcrtTable(log_n, Q30), pf.RNSBase32(Q30)
    basis = pf.BigUintApproxSignedBasis32(base, 15)
    ctx = pf.DcrtGlevContext32(t, base, basis, k)
    ell = basis.decompose_length()
    dg = to_dev32(rand32(rng, Q30, n, batch * (k + 1)))
    dk = to_dev32(rand32(rng, Q30, n, (k + 1) * ell * (k + 1)))
    out = torch.zeros_like(dg)
    s = torch.cuda.Stream()

    def work():
        pf.mul_dcrt_ggsw_to_dev(dg, dk, out, ctx, into_coeff_form=True, stream=s)
        t.transform_dev(out, stream=s)

    with torch.cuda.stream(s):
        work()
    s.synchronize()
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        work()
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
