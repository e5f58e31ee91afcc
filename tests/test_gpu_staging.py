"""Host-pointer (`*_slice`, `*_to`) entry points: the reference's methods take `&self, &mut [T]` and allocate nothing
per call (primus_ntt/src/ntt/prime64/table.rs:541-563, SURVEY §8b "no hidden allocation per call").  The C ABI's
host-pointer forms stage through a per-device pool of contexts (csrc/pfhe_staging.hpp): steady-state calls must not
allocate, concurrent callers must not share a context, long slices are cut into pieces, and every result stays
bit-exact."""
import os
import threading

import numpy as np
import pytest

from gpu_util import isolated, rand_mod, rand_rns
from pyref import Q61, Q62

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pf():
    import primus_fhe_amd as p
    return p


def alloc_count(pf):
    return int(pf.lib().pfhe_debug_alloc_count())


def test_steady_state_slice_calls_do_not_allocate(pf, orc):
    log_n = 16
    n = 1 << log_n
    t, o = pf.U64NttTable(log_n, Q61[0]), orc.U64NttTable(log_n, Q61[0])
    d, od = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    rng = np.random.default_rng(5)
    a = rand_mod(rng, Q61[0], n)
    r = rand_rns(rng, Q61, n, 2)
    ref_a = a.copy(); o.transform_slice(ref_a)
    ref_r = r.copy(); od.transform_slice(ref_r)
    # warm-up: the pool grows to the largest call
    x = r.copy(); d.transform_slice(x); d.inverse_transform_slice(x)
    x = a.copy(); t.transform_slice(x)
    before = alloc_count(pf)
    for _ in range(100):
        x = a.copy()
        t.transform_slice(x)
        assert np.array_equal(x, ref_a)
        t.inverse_transform_slice(x)
        assert np.array_equal(x, a)
    for _ in range(10):
        y = r.copy()
        d.transform_slice(y)
        assert np.array_equal(y, ref_r)
        d.lazy_inverse_transform_slice(y)
        assert np.array_equal(y % np.tile(np.repeat(np.array(Q61, dtype=np.uint64), n), 2), r)
    assert alloc_count(pf) == before, "host-pointer calls allocated in steady state"


def test_steady_state_u32_converter_and_rns_calls_do_not_allocate(pf, orc):
    q30 = [1073479681, 1071513601, 1070727169]
    log_n = 13
    n = 1 << log_n
    rng = np.random.default_rng(6)
    d32, o32 = pf.U32DcrtTable(log_n, q30), orc.U32DcrtTable(log_n, q30)
    a32 = np.concatenate([rng.integers(0, q, n, dtype=np.uint64).astype(np.uint32) for q in q30])
    ref32 = a32.copy(); o32.transform_slice(ref32)
    base, obase = pf.RNSBase(Q61), orc.RNSBase(Q61)
    res = rand_rns(rng, Q61, n, 1)
    ref_big = obase.compose_multiple_values_to(res, n)

    def once():
        x = a32.copy(); d32.transform_slice(x)
        assert np.array_equal(x, ref32)
        big = np.empty(n * base.big_uint_value_len(), np.uint64)
        base.compose_multiple_values_to(res, big, n)
        assert np.array_equal(big, ref_big)

    once()
    before = alloc_count(pf)
    for _ in range(20):
        once()
    assert alloc_count(pf) == before


def test_long_slice_is_cut_into_pieces_and_stays_exact(pf, orc):
    """47 RNS polynomials of 2^14 (17.6 MiB) exceed one piece (PFHE_STAGE_CHUNK, 8 MiB by default): the slice is pinned in
    place, one stream copies the pieces in, the other transforms each and copies it back; ragged last piece."""
    log_n, batch = 14, 47
    n = 1 << log_n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    rng = np.random.default_rng(7)
    a = rand_rns(rng, Q61, n, batch)
    assert a.nbytes > (16 << 20)
    ref = a.copy(); o.transform_slice(ref)
    x = a.copy(); d.transform_slice(x)
    assert np.array_equal(x, ref)
    d.inverse_transform_slice(x)
    assert np.array_equal(x, a)


def test_concurrent_host_slice_callers_get_their_own_context(pf, orc):
    """NttTable: Send + Sync (ntt/mod.rs:16): four host threads call transform_slice on one table at once."""
    log_n = 15
    n = 1 << log_n
    t, o = pf.U64NttTable(log_n, Q62), orc.U64NttTable(log_n, Q62)
    rng = np.random.default_rng(8)
    inputs = [rand_mod(rng, Q62, n * (i + 1)) for i in range(4)]
    refs = []
    for a in inputs:
        r = a.copy(); o.transform_slice(r); refs.append(r)
    errors = []

    def worker(i):
        try:
            for _ in range(25):
                x = inputs[i].copy()
                t.transform_slice(x)
                if not np.array_equal(x, refs[i]):
                    raise AssertionError(f"thread {i}: forward mismatch")
                t.inverse_transform_slice(x)
                if not np.array_equal(x, inputs[i]):
                    raise AssertionError(f"thread {i}: round trip mismatch")
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    assert not errors, errors
    # one thread now: every call takes the context on top of the pool, which grows to the largest slice once
    [worker(i) for i in range(4)]
    before = alloc_count(pf)
    [worker(i) for i in range(4)]
    assert not errors and alloc_count(pf) == before


def test_staging_release_returns_the_pool(pf):
    t = pf.U64NttTable(12, Q61[0])
    x = np.arange(1 << 12, dtype=np.uint64)
    t.transform_slice(x)
    assert pf.lib().pfhe_staging_release(-1) >= 1
    before = alloc_count(pf)
    t.inverse_transform_slice(x)   # the pool grows again
    assert np.array_equal(x, np.arange(1 << 12, dtype=np.uint64))
    assert alloc_count(pf) > before


@pytest.mark.parametrize("log_n", [10, 14, 15, 16, 17])
@pytest.mark.parametrize("kind", ["pm", "mont", "shoup"])
def test_zero_copy_slices_match_oracle(pf, orc, log_n, kind, monkeypatch):
    """Slices up to the bounce limit are copied into the pool's pinned buffer and transformed by kernels that read and
    write THAT buffer over the link themselves (ntt_transform_through_dev: out-of-place first / last pass for two-pass
    rings, the in-place kernel on the mapped buffer for single-pass ones); larger ones take the copy engines.  Every
    arithmetic policy, both directions, lazy forms, ragged batch, both sides of the limit."""
    moduli = {"pm": Q61, "mont": Q61, "shoup": [Q62]}[kind]
    if kind == "mont":
        monkeypatch.setenv("PFHE_DISABLE_PM", "1")
    d, o = pf.U64DcrtTable(log_n, moduli), orc.U64DcrtTable(log_n, moduli)
    monkeypatch.delenv("PFHE_DISABLE_PM", raising=False)
    n, L = 1 << log_n, len(moduli)
    batch = max(1, min(3, (8 << 20) // (L * n * 8)))
    rng = np.random.default_rng(log_n)
    a = rand_rns(rng, moduli, n, batch)
    assert a.nbytes <= (8 << 20)
    ref = a.copy(); o.transform_slice(ref)
    x = a.copy(); d.transform_slice(x)
    assert np.array_equal(x, ref)
    d.inverse_transform_slice(x)
    assert np.array_equal(x, a)
    qs = np.tile(np.repeat(np.array(moduli, dtype=np.uint64), n), batch)
    lz = a.copy(); d.lazy_transform_slice(lz)
    assert (lz < 4 * qs).all() and np.array_equal(lz % qs, ref)
    lzi = ref.copy(); d.lazy_inverse_transform_slice(lzi)
    assert (lzi < 2 * qs).all() and np.array_equal(lzi % qs, a)
    # an unaligned view of a larger buffer (the slice does not start on a page, nor on 16 bytes of the pinned range)
    big = np.zeros(a.size + 3, np.uint64)
    view = big[1:1 + a.size]
    view[:] = a
    d.transform_slice(view)
    assert np.array_equal(view, ref) and big[0] == 0 and big[-1] == 0 and big[-2] == 0


def test_zero_copy_and_copy_paths_agree(pf, orc):
    """PFHE_STAGE_ZERO_COPY is read once per process and cannot be flipped here; the copy-engine path is what slices above
    the bounce limit take: compare a one-limb polynomial (kernels on the pinned buffer), a 3-limb one (pageable copies)
    and the same polynomials inside a 24-polynomial slice."""
    log_n = 16
    n = 1 << log_n
    d = pf.U64DcrtTable(log_n, Q61)
    rng = np.random.default_rng(99)
    small = rand_rns(rng, Q61, n, 1)
    large = np.concatenate([small, rand_rns(rng, Q61, n, 23)])
    assert small.nbytes <= (8 << 20) < large.nbytes
    one = small[:n].copy()                     # 512 KiB: below the bounce limit
    t1 = pf.U64NttTable(log_n, Q61[0])
    t1.transform_slice(one)
    d.transform_slice(small)
    d.transform_slice(large)
    assert np.array_equal(large[:small.size], small) and np.array_equal(small[:n], one)


def test_memory_the_runtime_cannot_pin_takes_the_bounce_buffer(pf, orc, tmp_path):
    """The input of an out-of-place entry point may live in a read-only mapping: the call must succeed (the library never
    registers or writes caller memory it was not asked to write) and give the oracle's words; memory that already IS
    pinned (torch pinned tensor) is used as it is, mapped for the kernels."""
    import mmap

    import torch
    n = 1 << 14
    base, obase = pf.RNSBase(Q61), orc.RNSBase(Q61)
    rng = np.random.default_rng(11)
    res = rand_rns(rng, Q61, n, 1)
    ref = obase.compose_multiple_values_to(res, n)
    path = tmp_path / "residues.bin"
    path.write_bytes(res.tobytes())
    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        ro = np.frombuffer(mm, dtype=np.uint64)
        assert not ro.flags.writeable
        big = np.empty(n * base.big_uint_value_len(), np.uint64)
        base.compose_multiple_values_to(ro, big, n)
        assert np.array_equal(big, ref)
        del ro
        mm.close()
    # pinned memory of the caller: transform in place
    log_n = 15
    t, o = pf.U64NttTable(log_n, Q62), orc.U64NttTable(log_n, Q62)
    a = rand_mod(rng, Q62, 1 << log_n)
    exp = a.copy(); o.transform_slice(exp)
    pinned = torch.from_numpy(a.view(np.int64).copy()).pin_memory()
    view = pinned.numpy().view(np.uint64)
    t.transform_slice(view)
    assert np.array_equal(view, exp)


def test_a_pending_hip_error_of_the_caller_is_not_reported_as_ours(pf, orc):
    """hipGetLastError() is per thread and sticky.  A caller whose own HIP call failed just before (registering memory
    that is pinned already is refused) must not see that error come back from the library's first launch check: every entry
    point clears the thread's pending error on the way in (DeviceGuard, csrc/pfhe_capi.hip).  Found by
    tools/hazard_suite_probe.sh in round 5."""
    import torch
    rt = torch.cuda.cudart()
    log_n = 14
    t, o = pf.U64NttTable(log_n, Q62), orc.U64NttTable(log_n, Q62)
    rng = np.random.default_rng(77)
    a = rand_mod(rng, Q62, 1 << log_n)
    exp = a.copy(); o.transform_slice(exp)
    pinned = torch.from_numpy(a.view(np.int64).copy()).pin_memory()
    view = pinned.numpy().view(np.uint64)
    rc = rt.cudaHostRegister(view.ctypes.data, view.nbytes, 0)          # refused: the range is pinned already
    assert int(getattr(rc, "value", rc)) != 0
    t.transform_slice(view)
    assert np.array_equal(view, exp)
    dev = torch.from_numpy(a.view(np.int64).copy()).cuda()               # (torch checks the sticky error after its own
    torch.cuda.synchronize()                                            #  calls too: nothing of torch's in between)
    rc = rt.cudaHostRegister(view.ctypes.data, view.nbytes, 0)
    assert int(getattr(rc, "value", rc)) != 0
    t.transform_dev(dev)
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), exp)


def test_long_pageable_slice_with_helper_thread_is_exact_and_reports_errors(pf, orc):
    """Pageable slices of 8 MiB and more: the calling thread copies in and launches piece by piece, a helper thread
    copies each finished piece back (transform_host in csrc/pfhe_capi.hip).  Ragged piece boundaries, both directions,
    repeated calls (the pooled events are reused), and two callers at once."""
    import threading
    log_n = 15
    n = 1 << log_n
    d, o = pf.U64DcrtTable(log_n, Q61), orc.U64DcrtTable(log_n, Q61)
    rng = np.random.default_rng(21)
    a = rand_rns(rng, Q61, n, 29)          # 21.75 MiB: three pieces of 10 + 10 + 9 polynomials
    assert a.nbytes >= (8 << 20)
    ref = a.copy(); o.transform_slice(ref)
    for _ in range(3):
        x = a.copy()
        d.transform_slice(x)
        assert np.array_equal(x, ref)
        d.inverse_transform_slice(x)
        assert np.array_equal(x, a)
    errors = []

    def worker():
        try:
            y = a.copy()
            d.transform_slice(y)
            if not np.array_equal(y, ref):
                raise AssertionError("mismatch under concurrency")
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=worker) for _ in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors
    before = alloc_count(pf)
    x = a.copy(); d.transform_slice(x)
    assert np.array_equal(x, ref) and alloc_count(pf) == before


def path_count(pf, which):
    return int(pf.lib().pfhe_debug_stage_path_count(which))


MAPPED_CALLER, MAPPED_BOUNCE, DMA_CALLER, PAGEABLE, HELPER = range(5)


def _hip():
    import torch
    torch.cuda.init()
    return torch.cuda.cudart()


def _rc(v):
    return int(getattr(v, "value", v))


def _default_paths_only():
    """The path a call takes is asserted for unwrapped calls; tests/conftest.py's hazard probe registers every slice."""
    if os.environ.get("PFHE_TEST_CALLER_REGISTER") == "1":
        pytest.skip("staging path changed by a hazard-probe switch")


@isolated
def test_every_staging_path_is_the_one_meant_and_exact(pf, orc):
    """pfhe_debug_stage_path_count says which way a host-pointer call's bytes travelled: pageable slice up to the bounce
    limit -> kernels on the pool's own pinned buffer; the same slice REGISTERED by the caller -> the pool's buffer too
    (round 5: kernels never run on a registration); a slice in memory the caller ALLOCATED pinned -> kernels on the caller's
    memory; a long caller-pinned slice -> copy engines; a long pageable slice -> the helper-thread form."""
    _default_paths_only()
    import torch
    rt = _hip()
    log_n = 16
    n = 1 << log_n
    t, o = pf.U64NttTable(log_n, Q61[0]), orc.U64NttTable(log_n, Q61[0])
    rng = np.random.default_rng(31)
    a = rand_mod(rng, Q61[0], n)
    ref = a.copy(); o.transform_slice(ref)
    c0 = [path_count(pf, w) for w in range(5)]
    x = a.copy(); t.transform_slice(x)
    assert np.array_equal(x, ref)
    c1 = [path_count(pf, w) for w in range(5)]
    assert c1[MAPPED_BOUNCE] == c0[MAPPED_BOUNCE] + 1 and c1[MAPPED_CALLER] == c0[MAPPED_CALLER]
    y = a.copy()
    assert _rc(rt.cudaHostRegister(y.ctypes.data, y.nbytes, 0)) == 0
    try:
        t.transform_slice(y)
    finally:
        rt.cudaHostUnregister(y.ctypes.data)
    assert np.array_equal(y, ref)
    c2 = [path_count(pf, w) for w in range(5)]
    assert c2[MAPPED_CALLER] == c1[MAPPED_CALLER] and c2[MAPPED_BOUNCE] == c1[MAPPED_BOUNCE] + 1
    pinned = torch.from_numpy(a.view(np.int64).copy()).pin_memory()      # hipHostMalloc memory
    v = pinned.numpy().view(np.uint64)
    t.transform_slice(v)
    assert np.array_equal(v, ref)
    c2b = [path_count(pf, w) for w in range(5)]
    assert c2b[MAPPED_CALLER] == c2[MAPPED_CALLER] + 1 and c2b[MAPPED_BOUNCE] == c2[MAPPED_BOUNCE]
    c2 = c2b
    # 24 polynomials = 12 MiB: registered -> copy engines on the caller's memory; pageable -> helper thread
    big = np.concatenate([a] * 24)
    bref = np.concatenate([ref] * 24)
    z = big.copy()
    assert _rc(rt.cudaHostRegister(z.ctypes.data, z.nbytes, 0)) == 0
    try:
        t.transform_slice(z)
    finally:
        rt.cudaHostUnregister(z.ctypes.data)
    assert np.array_equal(z, bref)
    c3 = [path_count(pf, w) for w in range(5)]
    assert c3[DMA_CALLER] > c2[DMA_CALLER] and c3[HELPER] == c2[HELPER]
    w = big.copy(); t.transform_slice(w)
    assert np.array_equal(w, bref)
    c4 = [path_count(pf, w_) for w_ in range(5)]
    assert c4[HELPER] == c3[HELPER] + 1


@isolated
def test_slice_spanning_two_registrations_is_not_treated_as_one_pinned_range(pf, orc):
    """pinned first and last bytes do not make a pinned range.  One array whose two halves are registered
    SEPARATELY (two registrations, adjacent): a long slice over both must not be given to the copy engines as one pinned
    range — it travels as pageable memory — while a slice inside one registration is (copy engines on the caller's memory)."""
    _default_paths_only()
    rt = _hip()
    log_n = 15
    n = 1 << log_n
    per_half = 20                                   # 20 polynomials = 5 MiB per registration; both: 10 MiB, helper-thread size
    t, o = pf.U64NttTable(log_n, Q62), orc.U64NttTable(log_n, Q62)
    rng = np.random.default_rng(41)
    page = 4096
    words = 2 * per_half * n
    raw = np.zeros(words + page // 8, np.uint64)
    skip = (-raw.ctypes.data % page) // 8          # page-aligned start: registrations cover whole pages
    arr = raw[skip:skip + words]
    arr[:] = rand_mod(rng, Q62, words)
    orig = arr.copy()
    ref = orig.copy(); o.transform_slice(ref)
    half = per_half * n * 8
    assert _rc(rt.cudaHostRegister(arr.ctypes.data, half, 0)) == 0
    assert _rc(rt.cudaHostRegister(arr.ctypes.data + half, half, 0)) == 0
    try:
        c0 = [path_count(pf, w) for w in range(5)]
        t.transform_slice(arr)                      # 10 MiB over both registrations: the runtime refuses it as one
        #                                             range ('invalid argument'); staged chunk by chunk, no helper thread
        c1 = [path_count(pf, w) for w in range(5)]
        assert np.array_equal(arr, ref)
        assert c1[DMA_CALLER] == c0[DMA_CALLER] and c1[MAPPED_CALLER] == c0[MAPPED_CALLER] and c1[PAGEABLE] > c0[PAGEABLE]
        assert c1[HELPER] == c0[HELPER]
        arr[:] = orig
        t.transform_slice(arr[:per_half * n])       # inside the first registration
        c2 = [path_count(pf, w) for w in range(5)]
        assert np.array_equal(arr[:per_half * n], ref[:per_half * n]) and np.array_equal(arr[per_half * n:], orig[per_half * n:])
        assert c2[DMA_CALLER] > c1[DMA_CALLER] and c2[MAPPED_CALLER] == c1[MAPPED_CALLER]
    finally:
        rt.cudaHostUnregister(arr.ctypes.data)
        rt.cudaHostUnregister(arr.ctypes.data + half)


def test_idle_contexts_are_capped_and_the_helper_thread_is_reused(pf, orc):
    """A burst of concurrent callers leaves at most four contexts behind; the helper thread of a
    context is started once and parked between calls (no thread per call)."""
    import os
    log_n = 14
    n = 1 << log_n
    t, o = pf.U64NttTable(log_n, Q61[0]), orc.U64NttTable(log_n, Q61[0])
    rng = np.random.default_rng(51)
    a = rand_mod(rng, Q61[0], n)
    ref = a.copy(); o.transform_slice(ref)
    pf.lib().pfhe_staging_release(-1)
    errors = []
    gate = threading.Barrier(8)

    def worker():
        try:
            gate.wait()
            for _ in range(20):
                x = a.copy()
                t.transform_slice(x)
                if not np.array_equal(x, ref):
                    raise AssertionError("mismatch")
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=worker) for _ in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors
    assert 1 <= pf.lib().pfhe_staging_release(-1) <= 4
    # helper thread: the long pageable form twice; the process's thread count grows by at most one in all
    d, od = pf.U64DcrtTable(15, Q61), orc.U64DcrtTable(15, Q61)
    big = rand_rns(rng, Q61, 1 << 15, 12)           # 9 MiB
    bref = big.copy(); od.transform_slice(bref)

    def nthreads():
        return len(os.listdir("/proc/self/task"))

    y = big.copy(); d.transform_slice(y)
    assert np.array_equal(y, bref)
    after_first = nthreads()
    for _ in range(5):
        y = big.copy(); d.transform_slice(y)
        assert np.array_equal(y, bref)
    assert nthreads() <= after_first + 2, "a helper thread per call"   # (five more calls; the runtime may start a thread of its own)
