// pfhe_capi.hip — extern "C" boundary (include/pfhe.h).  Validates arguments, owns handles,
// never throws.  There is deliberately no CPU fallback: without a HIP device every create()
// fails with PFHE_ERR_NO_DEVICE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "pfhe_capi_internal.hpp"
#include "pfhe_common.hpp"
#include "pfhe_handles.hpp"
#include "pfhe_ntt_device.hpp"
#include "pfhe_pointwise.hpp"
#include "pfhe_staging.hpp"

namespace pfhe {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    (void)hipGetLastError();  // clear the sticky error
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? PFHE_ERR_NO_DEVICE : PFHE_ERR_HIP;
}

static thread_local int g_guard_depth = 0;

DeviceGuard::DeviceGuard(int device) {
    // hipGetLastError() is per host thread and sticky: an error the CALLER's own earlier HIP call left behind (a refused
    // hipHostRegister, say) would otherwise be reported by the first launch check of this entry point as if a kernel of
    // this library had failed (found by tools/hazard_suite_probe.sh, round 5).  Every entry point starts from a clean slate —
    // the OUTERMOST guard of a call only: entry points call one another (host-pointer forms call the device ones), and a
    // nested guard must not swallow the error of a launch this library queued a moment earlier.  (The caller's stale code
    // is consumed; include/pfhe.h says so: check your own HIP calls' return values.)
    if (g_guard_depth++ == 0) (void)hipGetLastError();
    if (hipGetDevice(&prev) != hipSuccess) {
        prev = -1;
        (void)hipGetLastError();
    }
    ok = hipSetDevice(device) == hipSuccess;
    if (!ok) (void)hipGetLastError();
}
DeviceGuard::~DeviceGuard() {
    --g_guard_depth;
    if (prev >= 0) (void)hipSetDevice(prev);
}

static int check_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_last_error("no HIP device available (libpfhe_hip has no CPU fallback)");
        return PFHE_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_last_error("device index out of range");
        return PFHE_ERR_NO_DEVICE;
    }
    return PFHE_OK;
}

TableSet::~TableSet() {
    DeviceGuard g(device);
    for (void *p : allocations) (void)counted_free(p);
}

// Builds host tables for every modulus, uploads them, and fills `out`.
int make_table_set(u32 log_n, const u64 *moduli, size_t count, int device, std::unique_ptr<TableSet> &out) {
    if (count == 0) {
        set_last_error("empty modulus list");
        return PFHE_ERR_BAD_ARGUMENT;
    }
    std::vector<HostTable> host(count);
    for (size_t i = 0; i < count; ++i) PFHE_TRY(build_host_table(log_n, moduli[i], host[i]));
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;

    auto ts = std::make_unique<TableSet>();
    ts->device = device;
    ts->log_n = log_n;
    ts->n = (size_t)1 << log_n;
    ts->L = (u32)count;
    ts->primes.resize(count);
    ts->tune = NttTuning::from_env();
    const size_t bytes = ts->n * sizeof(ulonglong2);
    bool all_pm = std::getenv("PFHE_DISABLE_PM") == nullptr;  // tuning switch: force the generic path
    bool all_mont = std::getenv("PFHE_DISABLE_MONT") == nullptr;  // tuning switch: generic primes keep the Shoup transforms
    for (size_t i = 0; i < count; ++i) {
        u32 pk = 0;
        u64 pc = 0;
        if (!pm_shape(host[i].q, pk, pc)) all_pm = false;
        if (!mont_shape(host[i].q)) all_mont = false;
    }
    const bool use_mont = !all_pm && all_mont && log_n >= 4;
    const auto upload = [&](const void *src, size_t nbytes, const void **dst) -> int {
        void *d = nullptr;
        PFHE_HIP(counted_malloc(&d, nbytes));
        ts->allocations.push_back(d);
        PFHE_HIP(hipMemcpy(d, src, nbytes, hipMemcpyHostToDevice));
        *dst = d;
        return PFHE_OK;
    };
    // the lane-ordered copy of the last four stages' twiddles (NttPrime::fwd_last / inv_last)
    const auto last_order = [&](const std::vector<ulonglong2> &fwd, const std::vector<ulonglong2> &inv,
                                std::vector<ulonglong2> &fl, std::vector<ulonglong2> &il) {
        const size_t n = ts->n, groups = n / 16;
        fl.assign(15 * groups, ulonglong2{0, 0});
        il.assign(15 * groups, ulonglong2{0, 0});
        for (int j = 3; j >= 0; --j) {
            const size_t per = (size_t)8 >> j;  // twiddles per group at distance 2^j
            for (size_t u = 0; u < per; ++u)
                for (size_t g = 0; g < groups; ++g) {
                    const size_t off = (per - 1 + u) * groups + g;
                    fl[off] = fwd[(n >> (j + 1)) + g * per + u];
                    il[off] = inv[1 + n - (n >> j) + g * per + u];
                }
        }
    };
    for (size_t i = 0; i < count; ++i) {
        const void *fwd = nullptr, *inv = nullptr;
        PFHE_TRY(upload(host[i].fwd.data(), bytes, &fwd));
        PFHE_TRY(upload(host[i].inv.data(), bytes, &inv));
        NttPrime &P = ts->primes[i];
        std::memset(&P, 0, sizeof P);
        P.q = host[i].q;
        P.two_q = host[i].q << 1;
        P.q3 = 3 * host[i].q;
        P.inv_n = host[i].inv_n;
        P.inv_n_p = (u64)(((unsigned __int128)host[i].inv_n << 64) / host[i].q);
        P.inv_n_w = host[i].inv_n_w;
        P.inv_n_w_p = (u64)(((unsigned __int128)host[i].inv_n_w << 64) / host[i].q);
        P.bar_lo = host[i].bar_lo;
        P.bar_hi = host[i].bar_hi;
        P.fwd = static_cast<const ulonglong2 *>(fwd);
        P.inv = static_cast<const ulonglong2 *>(inv);
        u32 pk = 0;
        u64 pc = 0;
        P.pm_k = pm_shape(P.q, pk, pc) ? pk : 0;
        P.pm_c = P.pm_k ? pc : 0;
        std::vector<ulonglong2> fl, il;
        if (P.pm_k) {
            // {w, w * 2^32 mod q}: the twiddle product of PmArith splits the multiplicand, not the twiddle
            const auto shifted = [&](u64 w) { return (u64)(((unsigned __int128)w << 32) % P.q); };
            std::vector<ulonglong2> fw(ts->n), iw(ts->n);
            for (size_t k = 0; k < ts->n; ++k) {
                fw[k] = ulonglong2{host[i].fwd[k].x, shifted(host[i].fwd[k].x)};
                iw[k] = ulonglong2{host[i].inv[k].x, shifted(host[i].inv[k].x)};
            }
            const void *fwp = nullptr, *iwp = nullptr;
            PFHE_TRY(upload(fw.data(), bytes, &fwp));
            PFHE_TRY(upload(iw.data(), bytes, &iwp));
            P.fwd_p = static_cast<const ulonglong2 *>(fwp);
            P.inv_p = static_cast<const ulonglong2 *>(iwp);
            P.inv_n_2 = shifted(P.inv_n);
            P.inv_n_w_2 = shifted(P.inv_n_w);
            if (all_pm && log_n >= 4) last_order(fw, iw, fl, il);
        }
        if (!all_pm && log_n >= 4) last_order(host[i].fwd, host[i].inv, fl, il);
        if (use_mont) {
            // {w * 2^32 mod q, w * 2^64 mod q}: the one-word Montgomery product of MontArith (pfhe_mont_asm.hpp)
            const u64 q = P.q;
            const auto mform = [&](u64 w) {
                const u64 a = (u64)(((unsigned __int128)w << 32) % q);
                return ulonglong2{a, (u64)(((unsigned __int128)a << 32) % q)};
            };
            std::vector<ulonglong2> fm(ts->n), im(ts->n), flm, ilm;
            for (size_t k = 0; k < ts->n; ++k) {
                fm[k] = mform(host[i].fwd[k].x);
                im[k] = mform(host[i].inv[k].x);
            }
            last_order(fm, im, flm, ilm);
            const void *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr;
            PFHE_TRY(upload(fm.data(), bytes, &a));
            PFHE_TRY(upload(im.data(), bytes, &b));
            PFHE_TRY(upload(flm.data(), flm.size() * sizeof(ulonglong2), &c));
            PFHE_TRY(upload(ilm.data(), ilm.size() * sizeof(ulonglong2), &d));
            P.fwd_m = static_cast<const ulonglong2 *>(a);
            P.inv_m = static_cast<const ulonglong2 *>(b);
            P.fwd_last_m = static_cast<const ulonglong2 *>(c);
            P.inv_last_m = static_cast<const ulonglong2 *>(d);
            const ulonglong2 nm = mform(P.inv_n), nwm = mform(P.inv_n_w);
            P.inv_n_m = nm.x, P.inv_n_m2 = nm.y, P.inv_n_w_m = nwm.x, P.inv_n_w_m2 = nwm.y;
            u32 inv = 1;  // Newton: q^-1 mod 2^32
            for (int it = 0; it < 5; ++it) inv *= 2u - (u32)q * inv;
            P.qinv32 = 0u - inv;
            P.mont_qest = (u32)((1ull << (63 - __builtin_clzll(q))) / ((q >> 32) + 1));
            P.mont_qf = ((1ull << 63) / q) * q;
        }
        if (!fl.empty()) {
            const void *flp = nullptr, *ilp = nullptr;
            PFHE_TRY(upload(fl.data(), fl.size() * sizeof(ulonglong2), &flp));
            PFHE_TRY(upload(il.data(), il.size() * sizeof(ulonglong2), &ilp));
            P.fwd_last = static_cast<const ulonglong2 *>(flp);
            P.inv_last = static_cast<const ulonglong2 *>(ilp);
        }
        ts->roots.push_back(host[i].root);
        ts->inv_roots.push_back(host[i].inv_root);
    }
    ts->pm = all_pm;
    ts->ntt_arith = all_pm ? kArithPm : (use_mont ? kArithMont : kArithShoup);
    void *pd = nullptr;
    PFHE_HIP(counted_malloc(&pd, count * sizeof(NttPrime)));
    ts->allocations.push_back(pd);
    PFHE_HIP(hipMemcpy(pd, ts->primes.data(), count * sizeof(NttPrime), hipMemcpyHostToDevice));
    ts->primes_dev = static_cast<const NttPrime *>(pd);
    void *md = nullptr;
    PFHE_HIP(counted_malloc(&md, count * sizeof(u64)));
    ts->allocations.push_back(md);
    std::vector<u64> mods(moduli, moduli + count);
    PFHE_HIP(hipMemcpy(md, mods.data(), count * sizeof(u64), hipMemcpyHostToDevice));
    ts->moduli_dev = static_cast<const u64 *>(md);
    out = std::move(ts);
    return PFHE_OK;
}

// len must be a positive-or-zero multiple of the unit (L*N words)
static int check_len(const TableSet &t, size_t len, u64 &units) {
    const size_t unit = t.n * t.L;
    if (len % unit != 0) {
        set_last_error("slice length is not a multiple of the polynomial length");
        return PFHE_ERR_BAD_LENGTH;
    }
    units = len / unit;
    return PFHE_OK;
}

int transform_dev(const TableSet &t, u64 *data, size_t len, bool inverse, bool lazy, hipStream_t s) {
    if (!data && len) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(data);
    u64 units = 0;
    PFHE_TRY(check_len(t, len, units));
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    const u64 npolys = units * t.L;
    return inverse ? ntt_inverse_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, data, npolys, lazy, s, t.tune)
                   : ntt_forward_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, data, npolys, lazy, s, t.tune);
}

// Host-pointer form of the transforms (table.rs:541-563 takes `&mut [T]` in place): the slice is staged through a
// pooled context (pfhe_staging.hpp: no allocation in steady state).  Polynomials are independent, so a long slice in
// memory the CALLER pinned is cut into pieces of whole units and pipelined over the context's two streams: one carries the
// copies in, the other waits for each piece, transforms it and copies it back — the copy back of piece i overlaps the
// copy in of piece i + 1 (the link is full duplex); pageable copies block the calling thread and go as one piece.
size_t stage_chunk_bytes() { return stage_knobs().chunk_bytes; }
size_t stage_bounce_max() { return stage_knobs().bounce_max; }
bool stage_zero_copy() { return stage_knobs().zero_copy; }

int transform_host(const TableSet &t, u64 *host, size_t len, bool inverse, bool lazy) {
    if (!host && len) return PFHE_ERR_BAD_ARGUMENT;
    u64 units = 0;
    PFHE_TRY(check_len(t, len, units));
    if (len == 0) return PFHE_OK;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    HostStage st(t.device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *dv = nullptr;
    PFHE_TRY(st.alloc(len * sizeof(u64), &dv));
    u64 *d = static_cast<u64 *>(dv);
    const size_t unit = t.n * t.L;
    // A slice of at most one bounce buffer takes no copy engine: the CPU copies it into the pool's pinned buffer (memory
    // the caller pinned is used as it is), the kernels read and write that buffer over the link themselves (first pass
    // host -> device scratch, last pass device scratch -> host; single-pass rings in place on the mapped memory), the CPU
    // copies the result back.  Two kernel launches and one synchronisation instead of copy, two kernels, copy
    // (tools/perf_host_slice.py); PFHE_STAGE_ZERO_COPY=0 keeps the copy engines.
    if (stage_zero_copy() && len * sizeof(u64) <= stage_bounce_max() && aligned16(host)) {
        u64 *mapped = static_cast<u64 *>(st.map(host, len * sizeof(u64)));
        const bool caller_mapped = mapped != nullptr;
        void *bounce = nullptr;
        if (!mapped) {
            void *bdev = nullptr;
            bounce = st.bounce(len * sizeof(u64), &bdev);
            if (bounce) {
                std::memcpy(bounce, host, len * sizeof(u64));
                mapped = static_cast<u64 *>(bdev);
            }
        }
        if (mapped) {
            const int rc = ntt_transform_through_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, mapped, d, units * t.L, inverse, lazy,
                                                     st.stream(), t.tune);
            if (rc != PFHE_ERR_UNSUPPORTED) {
                st.touch();
                stage_path_note(caller_mapped ? kPathMappedCaller : kPathMappedBounce);
                PFHE_TRY(rc);
                PFHE_TRY(st.finish());
                if (bounce) std::memcpy(host, bounce, len * sizeof(u64));
                return PFHE_OK;
            }
        }
    }
    const bool pinned = st.pin(host, len * sizeof(u64));
    // Long PAGEABLE slices: a pageable copy blocks the thread that issues it, so one thread alone cannot use both
    // directions of the link.  The context's HELPER THREAD (parked between calls, pfhe_staging.hpp) copies back while this
    // one copies in: the slice is cut into up to eight pieces of at least 6 MiB; this thread, piece by piece, copies in and
    // launches the transform on the context's first stream and records an event; the helper waits for each event and copies
    // that piece back on the second stream.  16 RNS polynomials of 2^16: 0.95 -> 0.79 ms, 64: 3.67 -> 2.54 ms
    // (tools/perf_host_slice.py).
    // Two rules keep the two threads apart.  (1) The helper sleeps on a condition variable until a piece is ready — no
    // spinning beside the copying thread.  (2) Adjacent pieces share the page that holds their common boundary (slices are
    // 8-byte aligned, not page aligned), and the runtime pins the caller's pages for the duration of a pageable copy: the
    // helper copies piece k back only once this thread has FINISHED copying piece k + 1 in, so the two threads never have
    // a page in common in flight (the withdrawn registration of round 4 — r04_experiments.txt item 6 — is the reason to be
    // strict about who maps the caller's pages when).
    const StageKnobs &K = stage_knobs();
    if (!pinned && K.helper_thread && units >= 2 && len * sizeof(u64) >= ((size_t)8 << 20) &&
        !st.touches_pinned(host, len * sizeof(u64))) {  // (a partly registered slice: copy_in / download stage it)
        // pieces of at least 6 MiB, at most eight (24 MiB: 2 / 3 / 4 / 6 / 8 pieces 861 / 829 / 808 / 855 / 844 us;
        // 96 MiB: 3.01 / 2.77 / 2.74 / 2.54 / 2.54 ms; one thread: 0.95 / 3.67 ms)
        const size_t pieces = std::max<size_t>(2, std::min<size_t>({K.pieces, (size_t)units, len * sizeof(u64) / ((size_t)6 << 20)}));
        std::vector<hipEvent_t> done(pieces);
        for (hipEvent_t &e : done) PFHE_TRY(st.take_event(&e));
        std::vector<size_t> off(pieces + 1);
        for (size_t i = 0; i <= pieces; ++i) off[i] = (size_t)(units * i / pieces) * unit;
        // shared with the helper's task: must outlive it on EVERY exit path.  `st` was declared before these locals and is
        // therefore destroyed after them, so its destructor's own wait would come too late for an exception thrown between
        // helper_start and helper_wait: the Joiner below, declared after everything the task references, aborts and waits first.
        struct Shared {
            std::mutex mu;
            std::condition_variable cv;
            size_t copied_in = 0;  // pieces whose copy in has returned and whose transform is launched (event recorded)
            bool all_in = false, abort = false;
        } sh;
        const hipStream_t s_in = st.stream(), s_out = st.stream2();
        st.touch();
        stage_path_note(kPathHelper);
        PFHE_TRY(st.helper_start(
            [&]() -> int {
                for (size_t i = 0; i < pieces; ++i) {
                    {
                        std::unique_lock<std::mutex> lk(sh.mu);
                        // piece i is launched AND this thread's neighbour piece i + 1 is no longer being copied in
                        sh.cv.wait(lk, [&] { return sh.abort || sh.all_in || sh.copied_in >= i + 1 + K.helper_lag; });
                        if (sh.abort) return PFHE_OK;
                    }
                    hipError_t e = hipEventSynchronize(done[i]);
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(host + off[i], d + off[i], (off[i + 1] - off[i]) * sizeof(u64), hipMemcpyDeviceToHost, s_out);
                    if (e == hipSuccess) e = hipStreamSynchronize(s_out);
                    if (e != hipSuccess) {
                        (void)hipGetLastError();
                        return PFHE_ERR_HIP;
                    }
                }
                return PFHE_OK;
            },
            [&]() {
                std::lock_guard<std::mutex> lk(sh.mu);
                sh.abort = true;
                sh.cv.notify_all();
            }));
        struct Joiner {
            HostStage &st;
            Shared &sh;
            ~Joiner() {
                if (!st.helper_busy()) return;
                {
                    std::lock_guard<std::mutex> lk(sh.mu);
                    sh.abort = true;
                }
                sh.cv.notify_all();
                (void)st.helper_wait();
            }
        } joiner{st, sh};
        int rc = PFHE_OK;
        for (size_t i = 0; i < pieces && rc == PFHE_OK; ++i) {
            const size_t words = off[i + 1] - off[i];
            if (hipMemcpyAsync(d + off[i], host + off[i], words * sizeof(u64), hipMemcpyHostToDevice, s_in) != hipSuccess) {
                rc = hip_fail(hipGetLastError(), "staged copy", __FILE__, __LINE__);
                break;
            }
            rc = transform_dev(t, d + off[i], words, inverse, lazy, s_in);
            if (rc == PFHE_OK && hipEventRecord(done[i], s_in) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipEventRecord", __FILE__, __LINE__);
            if (rc == PFHE_OK) {
                std::lock_guard<std::mutex> lk(sh.mu);
                sh.copied_in = i + 1;
                sh.all_in = i + 1 == pieces;
                sh.cv.notify_all();
            }
        }
        if (rc != PFHE_OK) {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.abort = true;
            sh.cv.notify_all();
        }
        const int helper_rc = st.helper_wait();
        PFHE_TRY(rc);
        PFHE_TRY(helper_rc);
        return st.finish();
    }
    // otherwise pageable copies go as one piece (they block the calling thread: nothing to pipeline)
    const size_t per = pinned ? std::max<size_t>(1, stage_chunk_bytes() / (unit * sizeof(u64))) : (size_t)units;
    const bool pipelined = per < units;
    const hipStream_t s_in = st.stream(), s_run = pipelined ? st.stream2() : st.stream();
    for (u64 u0 = 0; u0 < units; u0 += per) {
        const size_t words = (size_t)std::min<u64>(per, units - u0) * unit, off = (size_t)u0 * unit;
        PFHE_TRY(st.copy_in(d + off, host + off, words * sizeof(u64), s_in));
        if (pipelined) PFHE_TRY(st.order(s_in, s_run));
        PFHE_TRY(transform_dev(t, d + off, words, inverse, lazy, s_run));
        PFHE_TRY(st.download(host + off, d + off, words * sizeof(u64), s_run));
    }
    return st.finish();
}

int pointwise(const TableSet &t, int mode, u64 *acc, const u64 *a, size_t len_a, const u64 *b, size_t len_b,
              hipStream_t s) {
    if ((!acc || !b || (mode == 1 && !a)) && len_a) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(acc);
    PFHE_REQUIRE_ALIGNED(a);
    PFHE_REQUIRE_ALIGNED(b);
    u64 units = 0;
    PFHE_TRY(check_len(t, len_a, units));
    if (len_b != len_a && len_b != t.n * t.L) {
        set_last_error("multiplicand must have the same length or exactly one polynomial");
        return PFHE_ERR_BAD_LENGTH;
    }
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return mode == 0 ? pointwise_dev(acc, acc, b, nullptr, t.primes_dev, t.L, t.log_n, len_a, len_b, s, 0, t.pm)
                     : pointwise_dev(acc, a, b, acc, t.primes_dev, t.L, t.log_n, len_a, len_b, s, 0, t.pm);
}

// out = a*b (+ c): NttPolynomial::mul_to / mul_add_to (primus_poly/src/ntt/mul.rs:100-107, ntt/mod.rs:169-187)
int pointwise_to(const TableSet &t, u64 *out, const u64 *a, size_t len_a, const u64 *b, size_t len_b, const u64 *c,
                 bool has_c, hipStream_t s) {
    if ((!out || !a || !b || (has_c && !c)) && len_a) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(out);
    PFHE_REQUIRE_ALIGNED(a);
    PFHE_REQUIRE_ALIGNED(b);
    PFHE_REQUIRE_ALIGNED(c);
    u64 units = 0;
    PFHE_TRY(check_len(t, len_a, units));
    if (len_b != len_a && len_b != t.n * t.L) {
        set_last_error("multiplicand must have the same length or exactly one polynomial");
        return PFHE_ERR_BAD_LENGTH;
    }
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return pointwise_dev(out, a, b, has_c ? c : nullptr, t.primes_dev, t.L, t.log_n, len_a, len_b, s, 0, t.pm);
}

// minus_one: the coefficient of limb i is q_i - 1 (DcrtTable::transform_coeff_minus_one_monomial,
// primus_ntt/src/dcrt/mod.rs:124-134); otherwise `coeff` for every limb.
int monomial(const TableSet &t, u64 coeff, size_t degree, u64 *values, size_t len, bool host, hipStream_t s,
             bool minus_one = false) {
    if (!values) return PFHE_ERR_BAD_ARGUMENT;
    if (len != t.n * t.L) {
        set_last_error("monomial output must be exactly one polynomial");
        return PFHE_ERR_BAD_LENGTH;
    }
    // the per-limb scalars travel by value as kernel arguments (no staging buffer: the device form is capturable), at most
    // kMaxMonomialLimbs of them per launch; wider bases take one launch per group of limbs
    std::vector<MonomialScalars> groups((t.L + kMaxMonomialLimbs - 1) / kMaxMonomialLimbs);
    for (u32 i = 0; i < t.L; ++i) {
        const u64 q = t.primes[i].q;
        const u64 ci = minus_one ? q - 1 : coeff;
        if (ci >= q) {
            set_last_error("monomial coefficient must be reduced modulo every modulus");
            return PFHE_ERR_BAD_ARGUMENT;
        }
        MonomialScalars &sc = groups[i / kMaxMonomialLimbs];
        sc.value[i % kMaxMonomialLimbs] = ci;
        sc.quotient[i % kMaxMonomialLimbs] = (u64)(((unsigned __int128)ci << 64) / q);
    }
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    const u64 deg = (u64)degree & (2 * (u64)t.n - 1);
    const auto run = [&](u64 *out) -> int {
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const u32 l0 = (u32)gi * kMaxMonomialLimbs, lg = std::min<u32>(kMaxMonomialLimbs, t.L - l0);
            PFHE_TRY(monomial_dev(out + (size_t)l0 * t.n, t.primes_dev + l0, lg, t.log_n, deg, groups[gi], s));
        }
        return PFHE_OK;
    };
    if (!host)  // device output: launches on the caller's stream, nothing else (capturable)
        return run(values);
    // host output: the caller's stream is not involved (pooled staging context, no allocation in steady state)
    HostStage st(t.device);
    if (!st.ok()) return PFHE_ERR_HIP;
    void *out_dev = nullptr;
    PFHE_TRY(st.alloc(len * sizeof(u64), &out_dev));
    s = st.stream();
    PFHE_TRY(run(static_cast<u64 *>(out_dev)));
    PFHE_TRY(st.download(values, out_dev, len * sizeof(u64)));
    return st.finish();
}

}  // namespace pfhe

using namespace pfhe;

struct pfhe_ntt {
    std::unique_ptr<TableSet> t;
};
struct pfhe_dcrt {
    std::unique_ptr<TableSet> t;
};

namespace pfhe {
int capi_check_device(int device) { return check_device(device); }
const TableSet *capi_table_of(const pfhe_dcrt *t) { return t->t.get(); }
}  // namespace pfhe


extern "C" {

const char *pfhe_status_string(int status) {
    switch (status) {
        case PFHE_OK: return "ok";
        case PFHE_ERR_NO_PRIMITIVE_ROOT: return "there is no primitive root with this degree and modulus";
        case PFHE_ERR_DEGREE_CONVERSION: return "out of range integral type conversion attempted";
        case PFHE_ERR_DEGREE_TOO_LARGE: return "degree should be less than modulus";
        case PFHE_ERR_NTT_TABLE: return "failed to generate the desired ntt table";
        case PFHE_ERR_MODULUS_TOO_LARGE: return "modulus is too large for this NTT table (max 62-bit supported)";
        case PFHE_ERR_EMPTY_BASE: return "RNS base is empty";
        case PFHE_ERR_COPRIME: return "RNS moduli are not pairwise coprime";
        case PFHE_ERR_UNREPRESENTABLE_MODULUS: return "RNS modulus is not representable";
        case PFHE_ERR_BAD_LENGTH: return "slice length does not match the table";
        case PFHE_ERR_BAD_ARGUMENT: return "bad argument";
        case PFHE_ERR_NO_DEVICE: return "no usable HIP device";
        case PFHE_ERR_HIP: return "HIP runtime error";
        case PFHE_ERR_UNSUPPORTED: return "unsupported parameter";
        case PFHE_ERR_NO_INVERSE: return "element has no inverse";
        case PFHE_ERR_BUSY: return "external-product plan is held by another thread";
    }
    return "unknown status";
}

const char *pfhe_last_error(void) { return g_last_error.c_str(); }
const char *pfhe_version(void) { return "libpfhe_hip 0.1.0 gfx950"; }

int pfhe_device_count(int *count) {
    if (!count) return PFHE_ERR_BAD_ARGUMENT;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count = c;
    return PFHE_OK;
}

uint64_t pfhe_debug_alloc_count(void) { return alloc_event_count(); }
uint64_t pfhe_debug_stage_path_count(int which) { return stage_path_count(which); }
int pfhe_staging_release(int device) { return staging_release(device); }

int pfhe_device_malloc(int device, size_t bytes, void **out) {
    if (!out) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    if (bytes == 0) return PFHE_OK;
    PFHE_HIP(counted_malloc(out, bytes));
    return PFHE_OK;
}

int pfhe_device_free(int device, void *ptr) {
    if (!ptr) return PFHE_OK;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    PFHE_HIP(counted_free(ptr));
    return PFHE_OK;
}

int pfhe_memcpy_h2d(int device, void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return PFHE_OK;
    if (!dst || !src) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    PFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    PFHE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PFHE_OK;
}

int pfhe_memcpy_d2h(int device, void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return PFHE_OK;
    if (!dst || !src) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    PFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    PFHE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PFHE_OK;
}

int pfhe_memcpy_d2d(int device, void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return PFHE_OK;
    if (!dst || !src) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    PFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PFHE_OK;
}

int pfhe_memset_dev(int device, void *dst, int byte, size_t bytes, void *stream) {
    if (bytes == 0) return PFHE_OK;
    if (!dst) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    PFHE_HIP(hipMemsetAsync(dst, byte, bytes, (hipStream_t)stream));
    return PFHE_OK;
}

int pfhe_stream_synchronize(int device, void *stream) {
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    PFHE_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PFHE_OK;
}

int pfhe_fill_uniform_dev(int device, uint64_t *dst, size_t len, const uint64_t *moduli, size_t moduli_count,
                          size_t poly_len, uint64_t seed, void *stream) {
    PFHE_GUARD_BEGIN
    if (len == 0) return PFHE_OK;
    if (!dst || !moduli || moduli_count == 0 || poly_len == 0) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_TRY(check_device(device));
    DeviceGuard g(device);
    HostStage st(device);  // the modulus list travels through a pooled staging context (no allocation per call)
    if (!st.ok()) return PFHE_ERR_HIP;
    void *md = nullptr;
    PFHE_TRY(st.upload(moduli, moduli_count * sizeof(u64), &md));
    PFHE_TRY(st.finish());
    PFHE_TRY(fill_uniform_dev((u64 *)dst, len, (const u64 *)md, moduli_count, poly_len, seed, (hipStream_t)stream));
    PFHE_HIP(hipStreamSynchronize((hipStream_t)stream));  // the list must outlive the kernel
    return PFHE_OK;
    PFHE_GUARD_END
}

/* ---------------------------- U64NttTable ---------------------------- */

int pfhe_ntt_create(uint32_t log_n, uint64_t modulus, int device, pfhe_ntt **out) {
    PFHE_GUARD_BEGIN
    if (!out) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<TableSet> t;
    u64 q = modulus;
    PFHE_TRY(make_table_set(log_n, &q, 1, device, t));
    *out = new pfhe_ntt{std::move(t)};
    return PFHE_OK;
    PFHE_GUARD_END
}

void pfhe_ntt_destroy(pfhe_ntt *table) { delete table; }
size_t pfhe_ntt_poly_length(const pfhe_ntt *t) { return t ? t->t->n : 0; }
uint32_t pfhe_ntt_log_n(const pfhe_ntt *t) { return t ? t->t->log_n : 0; }
uint64_t pfhe_ntt_modulus(const pfhe_ntt *t) { return t ? t->t->primes[0].q : 0; }
uint64_t pfhe_ntt_root(const pfhe_ntt *t) { return t ? t->t->roots[0] : 0; }
uint64_t pfhe_ntt_inv_root(const pfhe_ntt *t) { return t ? t->t->inv_roots[0] : 0; }
uint64_t pfhe_ntt_inv_n(const pfhe_ntt *t) { return t ? t->t->primes[0].inv_n : 0; }
int pfhe_ntt_device(const pfhe_ntt *t) { return t ? t->t->device : -1; }

#define PFHE_NTT_HOST(name, INV, LAZY)                                          \
    int name(const pfhe_ntt *table, uint64_t *p, size_t len) {                  \
        PFHE_GUARD_BEGIN                                                        \
        if (!table) return PFHE_ERR_BAD_ARGUMENT;                               \
        return transform_host(*table->t, (u64 *)p, len, INV, LAZY);             \
        PFHE_GUARD_END                                                          \
    }
PFHE_NTT_HOST(pfhe_ntt_transform_slice, false, false)
PFHE_NTT_HOST(pfhe_ntt_inverse_transform_slice, true, false)
PFHE_NTT_HOST(pfhe_ntt_lazy_transform_slice, false, true)
PFHE_NTT_HOST(pfhe_ntt_lazy_inverse_transform_slice, true, true)
#undef PFHE_NTT_HOST

int pfhe_ntt_transform_monomial(const pfhe_ntt *table, uint64_t coeff, size_t degree, uint64_t *values,
                                size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial(*table->t, coeff, degree, (u64 *)values, len, true, nullptr);
    PFHE_GUARD_END
}

int pfhe_ntt_transform_coeff_one_monomial(const pfhe_ntt *table, size_t degree, uint64_t *values, size_t len) {
    return pfhe_ntt_transform_monomial(table, 1, degree, values, len);
}

int pfhe_ntt_transform_coeff_minus_one_monomial(const pfhe_ntt *table, size_t degree, uint64_t *values,
                                                size_t len) {
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pfhe_ntt_transform_monomial(table, table->t->primes[0].q - 1, degree, values, len);
}

int pfhe_ntt_transform_dev(const pfhe_ntt *table, uint64_t *poly_dev, size_t len, int lazy, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform_dev(*table->t, (u64 *)poly_dev, len, false, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_ntt_inverse_transform_dev(const pfhe_ntt *table, uint64_t *values_dev, size_t len, int lazy,
                                   void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform_dev(*table->t, (u64 *)values_dev, len, true, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_ntt_transform_monomial_dev(const pfhe_ntt *table, uint64_t coeff, size_t degree, uint64_t *values_dev,
                                    size_t len, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial(*table->t, coeff, degree, (u64 *)values_dev, len, false, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_ntt_mul_assign_dev(const pfhe_ntt *table, uint64_t *a_dev, size_t len_a, const uint64_t *b_dev,
                            size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise(*table->t, 0, (u64 *)a_dev, nullptr, len_a, (const u64 *)b_dev, len_b, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_ntt_add_mul_assign_dev(const pfhe_ntt *table, uint64_t *acc_dev, const uint64_t *a_dev, size_t len_a,
                                const uint64_t *b_dev, size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise(*table->t, 1, (u64 *)acc_dev, (const u64 *)a_dev, len_a, (const u64 *)b_dev, len_b,
                     (hipStream_t)stream);
    PFHE_GUARD_END
}

/* ---------------------------- U64DcrtTable ---------------------------- */

int pfhe_dcrt_create(uint32_t log_n, const uint64_t *moduli, size_t moduli_count, int device, pfhe_dcrt **out) {
    PFHE_GUARD_BEGIN
    if (!out || (!moduli && moduli_count)) return PFHE_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<TableSet> t;
    PFHE_TRY(make_table_set(log_n, (const u64 *)moduli, moduli_count, device, t));
    *out = new pfhe_dcrt{std::move(t)};
    return PFHE_OK;
    PFHE_GUARD_END
}

void pfhe_dcrt_destroy(pfhe_dcrt *table) { delete table; }
size_t pfhe_dcrt_poly_length(const pfhe_dcrt *t) { return t ? t->t->n : 0; }
size_t pfhe_dcrt_moduli_count(const pfhe_dcrt *t) { return t ? t->t->L : 0; }
size_t pfhe_dcrt_crt_poly_length(const pfhe_dcrt *t) { return t ? t->t->n * t->t->L : 0; }
int pfhe_dcrt_device(const pfhe_dcrt *t) { return t ? t->t->device : -1; }
uint64_t pfhe_dcrt_modulus(const pfhe_dcrt *t, size_t i) { return (t && i < t->t->L) ? t->t->primes[i].q : 0; }
uint64_t pfhe_dcrt_root(const pfhe_dcrt *t, size_t i) { return (t && i < t->t->L) ? t->t->roots[i] : 0; }
uint64_t pfhe_dcrt_inv_n(const pfhe_dcrt *t, size_t i) { return (t && i < t->t->L) ? t->t->primes[i].inv_n : 0; }

#define PFHE_DCRT_HOST(name, INV, LAZY)                                         \
    int name(const pfhe_dcrt *table, uint64_t *p, size_t len) {                 \
        PFHE_GUARD_BEGIN                                                        \
        if (!table) return PFHE_ERR_BAD_ARGUMENT;                               \
        return transform_host(*table->t, (u64 *)p, len, INV, LAZY);             \
        PFHE_GUARD_END                                                          \
    }
PFHE_DCRT_HOST(pfhe_dcrt_transform_slice, false, false)
PFHE_DCRT_HOST(pfhe_dcrt_inverse_transform_slice, true, false)
PFHE_DCRT_HOST(pfhe_dcrt_lazy_transform_slice, false, true)
PFHE_DCRT_HOST(pfhe_dcrt_lazy_inverse_transform_slice, true, true)
#undef PFHE_DCRT_HOST

int pfhe_dcrt_transform_monomial(const pfhe_dcrt *table, uint64_t coeff, size_t degree, uint64_t *values,
                                 size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial(*table->t, coeff, degree, (u64 *)values, len, true, nullptr);
    PFHE_GUARD_END
}

int pfhe_dcrt_transform_coeff_one_monomial(const pfhe_dcrt *table, size_t degree, uint64_t *values, size_t len) {
    return pfhe_dcrt_transform_monomial(table, 1, degree, values, len);
}

int pfhe_dcrt_transform_coeff_minus_one_monomial(const pfhe_dcrt *table, size_t degree, uint64_t *values, size_t len) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return monomial(*table->t, 0, degree, (u64 *)values, len, true, nullptr, /*minus_one=*/true);
    PFHE_GUARD_END
}

int pfhe_dcrt_transform_monomial_dev(const pfhe_dcrt *table, uint64_t coeff, size_t degree, uint64_t *values_dev,
                                     size_t len, int minus_one, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(values_dev);
    return monomial(*table->t, minus_one ? 0 : coeff, degree, (u64 *)values_dev, len, false, (hipStream_t)stream,
                    minus_one != 0);
    PFHE_GUARD_END
}

int pfhe_dcrt_transform_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len, int lazy, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform_dev(*table->t, (u64 *)poly_dev, len, false, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_inverse_transform_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len, int lazy,
                                    void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return transform_dev(*table->t, (u64 *)poly_dev, len, true, lazy != 0, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_mul_assign_dev(const pfhe_dcrt *table, uint64_t *a_dev, size_t len_a, const uint64_t *b_dev,
                             size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise(*table->t, 0, (u64 *)a_dev, nullptr, len_a, (const u64 *)b_dev, len_b, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_add_mul_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev, const uint64_t *a_dev, size_t len_a,
                                 const uint64_t *b_dev, size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise(*table->t, 1, (u64 *)acc_dev, (const u64 *)a_dev, len_a, (const u64 *)b_dev, len_b,
                     (hipStream_t)stream);
    PFHE_GUARD_END
}

static int butterfly_api(const pfhe_dcrt *table, bool factor, uint64_t *a_dev, const uint64_t *rhs_dev, size_t len,
                         const uint64_t *w_dev, size_t len_w, uint64_t *result_dev, void *stream) {
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    const TableSet &t = *table->t;
    const size_t unit = t.n * t.L;
    if (len % unit != 0 || (len_w != unit * (factor ? 2 : 1) && len_w != len * (factor ? 2 : 1))) {
        set_last_error("butterfly: slices must be multiples of L*N words; the multiplicand is one polynomial "
                       "(shared) or matches the batch (ShoupFactor slices count two words per coefficient)");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (len == 0) return PFHE_OK;
    if (!a_dev || !rhs_dev || !w_dev || !result_dev) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(a_dev);
    PFHE_REQUIRE_ALIGNED(rhs_dev);
    PFHE_REQUIRE_ALIGNED(w_dev);
    PFHE_REQUIRE_ALIGNED(result_dev);
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return butterfly_dev(factor, (u64 *)a_dev, (const u64 *)rhs_dev, (const u64 *)w_dev, (u64 *)result_dev, t.primes_dev,
                         t.L, t.log_n, len, factor ? len_w / 2 : len_w, (hipStream_t)stream, t.pm);
}

int pfhe_dcrt_butterfly_mul_dcrt_polynomial_to_dev(const pfhe_dcrt *table, uint64_t *a_dev, const uint64_t *rhs_dev,
                                                   size_t len, const uint64_t *dcrt_poly_dev, size_t len_w,
                                                   uint64_t *result_dev, void *stream) {
    PFHE_GUARD_BEGIN
    return butterfly_api(table, false, a_dev, rhs_dev, len, dcrt_poly_dev, len_w, result_dev, stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_butterfly_mul_factor_to_dev(const pfhe_dcrt *table, uint64_t *a_dev, const uint64_t *rhs_dev, size_t len,
                                          const uint64_t *factor_poly_dev, size_t len_w, uint64_t *result_dev,
                                          void *stream) {
    PFHE_GUARD_BEGIN
    return butterfly_api(table, true, a_dev, rhs_dev, len, factor_poly_dev, len_w, result_dev, stream);
    PFHE_GUARD_END
}

int pfhe_ntt_mul_to_dev(const pfhe_ntt *table, const uint64_t *a_dev, size_t len_a, const uint64_t *b_dev, size_t len_b,
                        uint64_t *out_dev, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise_to(*table->t, (u64 *)out_dev, (const u64 *)a_dev, len_a, (const u64 *)b_dev, len_b, nullptr, false,
                        (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_ntt_mul_add_to_dev(const pfhe_ntt *table, const uint64_t *a_dev, size_t len_a, const uint64_t *b_dev,
                            size_t len_b, const uint64_t *c_dev, uint64_t *out_dev, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise_to(*table->t, (u64 *)out_dev, (const u64 *)a_dev, len_a, (const u64 *)b_dev, len_b,
                        (const u64 *)c_dev, true, (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_dcrt_mul_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, size_t len_a, const uint64_t *b_dev,
                         size_t len_b, uint64_t *out_dev, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise_to(*table->t, (u64 *)out_dev, (const u64 *)a_dev, len_a, (const u64 *)b_dev, len_b, nullptr, false,
                        (hipStream_t)stream);
    PFHE_GUARD_END
}
int pfhe_dcrt_mul_add_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, size_t len_a, const uint64_t *b_dev,
                             size_t len_b, const uint64_t *c_dev, uint64_t *out_dev, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    return pointwise_to(*table->t, (u64 *)out_dev, (const u64 *)a_dev, len_a, (const u64 *)b_dev, len_b,
                        (const u64 *)c_dev, true, (hipStream_t)stream);
    PFHE_GUARD_END
}

int pfhe_dcrt_add_dcrt_glwe_mul_dcrt_polynomial_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev,
                                                           const uint64_t *dcrt_glwe_dev, size_t len,
                                                           const uint64_t *dcrt_poly_dev, size_t len_poly,
                                                           size_t glwe_polys, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table || glwe_polys == 0 || ((!acc_dev || !dcrt_glwe_dev || !dcrt_poly_dev) && len)) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(acc_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_glwe_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_poly_dev);
    const TableSet &t = *table->t;
    const size_t unit = t.n * t.L;
    if (len % (unit * glwe_polys) != 0 || len_poly != len / glwe_polys) {
        set_last_error("expected batch*(k+1) polynomials in acc / glwe and batch polynomials in dcrt_poly");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (len == 0) return PFHE_OK;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return pointwise_dev((u64 *)acc_dev, (const u64 *)dcrt_glwe_dev, (const u64 *)dcrt_poly_dev, (const u64 *)acc_dev,
                         t.primes_dev, t.L, t.log_n, len, len_poly, (hipStream_t)stream, (u64)unit * glwe_polys, t.pm);
    PFHE_GUARD_END
}

int pfhe_dcrt_glwe_mul_dcrt_polynomial_to_dev(const pfhe_dcrt *table, const uint64_t *dcrt_glwe_dev, size_t len,
                                              const uint64_t *dcrt_poly_dev, size_t len_poly, size_t glwe_polys,
                                              uint64_t *result_dev, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table || glwe_polys == 0 || ((!result_dev || !dcrt_glwe_dev || !dcrt_poly_dev) && len)) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(result_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_glwe_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_poly_dev);
    const TableSet &t = *table->t;
    const size_t unit = t.n * t.L;
    if (len % (unit * glwe_polys) != 0 || len_poly != len / glwe_polys) {
        set_last_error("expected batch*(k+1) polynomials in the glwe / result and batch polynomials in dcrt_poly");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (len == 0) return PFHE_OK;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return pointwise_dev((u64 *)result_dev, (const u64 *)dcrt_glwe_dev, (const u64 *)dcrt_poly_dev, nullptr, t.primes_dev,
                         t.L, t.log_n, len, len_poly, (hipStream_t)stream, (u64)unit * glwe_polys, t.pm);
    PFHE_GUARD_END
}

int pfhe_dcrt_transform_num_passes(const pfhe_dcrt *table) {
    return table ? ntt_num_passes(table->t->log_n, table->t->ntt_arith, table->t->tune) : 0;
}

const char *pfhe_dcrt_transform_pass_name(const pfhe_dcrt *table, int inverse, int index) {
    static thread_local char buf[96];
    buf[0] = 0;
    if (table) ntt_pass_name(table->t->log_n, inverse != 0, index, buf, sizeof buf, table->t->ntt_arith, table->t->tune);
    return buf;
}

int pfhe_dcrt_transform_form(const pfhe_dcrt *table, size_t len, int inverse, char *name, size_t cap, int *launches) {
    if (!table || !name || cap == 0 || !launches) return PFHE_ERR_BAD_ARGUMENT;
    const TableSet &t = *table->t;
    if (len % (t.n * t.L) != 0) return PFHE_ERR_BAD_LENGTH;
    *launches = ntt_transform_form(t.L, t.log_n, t.ntt_arith, len / t.n, inverse != 0, t.tune, name, cap);
    return PFHE_OK;
}

int pfhe_dcrt_transform_pass_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len, int inverse, int index,
                                 int lazy, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table || (!poly_dev && len)) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(poly_dev);
    const TableSet &t = *table->t;
    if (len % (t.n * t.L) != 0) return PFHE_ERR_BAD_LENGTH;
    DeviceGuard g(t.device);
    if (!g.ok) return PFHE_ERR_NO_DEVICE;
    return ntt_pass_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, (u64 *)poly_dev, len / t.n, inverse != 0, index,
                        lazy != 0, (hipStream_t)stream, nullptr, 0, t.tune);
    PFHE_GUARD_END
}

int pfhe_dcrt_mul_dcrt_polynomial_dev(const pfhe_dcrt *table, uint64_t *crt_poly_dev, size_t len,
                                      const uint64_t *dcrt_poly_dev, size_t len_b, void *stream) {
    PFHE_GUARD_BEGIN
    if (!table) return PFHE_ERR_BAD_ARGUMENT;
    const TableSet &t = *table->t;
    // every argument is checked BEFORE the forward transform is enqueued: a call that fails leaves the caller's
    // polynomial in coefficient form
    if ((!crt_poly_dev || !dcrt_poly_dev) && len) return PFHE_ERR_BAD_ARGUMENT;
    PFHE_REQUIRE_ALIGNED(crt_poly_dev);
    PFHE_REQUIRE_ALIGNED(dcrt_poly_dev);
    if (len % (t.n * t.L) != 0) {
        set_last_error("slice length is not a multiple of the polynomial length");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (len_b != len && len_b != t.n * t.L) {
        set_last_error("multiplicand must have the same length or exactly one polynomial");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (len != 0) {
        // forward block pass -> product -> inverse block pass in one kernel, between the strided passes
        DeviceGuard g(t.device);
        if (!g.ok) return PFHE_ERR_NO_DEVICE;
        const int rc = ntt_polymul_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, (u64 *)crt_poly_dev, len / t.n,
                                       (const u64 *)dcrt_poly_dev, len_b / t.n, (hipStream_t)stream, t.tune);
        if (rc != PFHE_ERR_UNSUPPORTED) return rc;
    }
    PFHE_TRY(transform_dev(t, (u64 *)crt_poly_dev, len, false, false, (hipStream_t)stream));
    if (t.log_n >= 4 && len != 0) {
        // shapes the fused kernel does not cover: the pointwise product rides on the loads of the inverse transform's first pass
        DeviceGuard g(t.device);
        if (!g.ok) return PFHE_ERR_NO_DEVICE;
        return ntt_inverse_mul_dev(t.primes_dev, t.L, t.log_n, t.ntt_arith, (u64 *)crt_poly_dev, len / t.n,
                                   (const u64 *)dcrt_poly_dev, len_b / t.n, (hipStream_t)stream, t.tune);
    }
    PFHE_TRY(pointwise(t, 0, (u64 *)crt_poly_dev, nullptr, len, (const u64 *)dcrt_poly_dev, len_b,
                       (hipStream_t)stream));
    return transform_dev(t, (u64 *)crt_poly_dev, len, true, false, (hipStream_t)stream);
    PFHE_GUARD_END
}

}  // extern "C"
