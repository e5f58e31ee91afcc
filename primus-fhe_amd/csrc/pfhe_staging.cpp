// pfhe_staging.cpp — per-device pool of staging contexts for the host-pointer entry points (pfhe_staging.hpp).
#include <sys/mman.h>
#include "pfhe_staging.hpp"

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace pfhe {

namespace {
std::atomic<std::uint64_t> g_alloc_events{0};

// a non-negative decimal integer, the whole value; anything else (empty, signs, trailing text, overflow) -> dflt
size_t env_bytes(const char *name, size_t dflt) {
    const char *v = std::getenv(name);
    if (!v || !*v || *v == '-' || *v == '+') return dflt;
    char *end = nullptr;
    errno = 0;
    const unsigned long long x = std::strtoull(v, &end, 10);
    return (end == v || *end != '\0' || errno == ERANGE) ? dflt : (size_t)x;
}
constexpr size_t kAlign = 256;
size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
std::atomic<std::uint64_t> g_paths[kPathCount];
}  // namespace

const StageKnobs &stage_knobs() {
    static const StageKnobs k = [] {
        StageKnobs v;
        // slices of at least register_min bytes are looked up: memory the caller pinned is used as it is
        v.use_register = true;
        v.register_min = (size_t)128 << 10;
        // transfers up to bounce_max go through the pinned bounce buffer (a CPU copy + a true asynchronous DMA, or kernels
        // on that buffer); larger ones are handed to the runtime as they are (it pins the caller's pages in pieces)
        v.bounce_max = env_bytes("PFHE_STAGE_BOUNCE_MAX", (size_t)1 << 20);
        v.cache_max = (size_t)2 << 30;
        const size_t chunk = env_bytes("PFHE_STAGE_CHUNK", 0);
        v.chunk_bytes = chunk ? chunk : (size_t)8 << 20;
        // PFHE_STAGE_ZERO_COPY=0: no kernel ever reads or writes host memory (neither the caller's pinned memory nor
        // the pool's own buffer) — every byte crosses the link through the copy engines
        v.zero_copy = env_bytes("PFHE_STAGE_ZERO_COPY", 1) != 0;
        v.helper_thread = true;
        v.pieces = 8;
        // idle contexts kept per device (each holds its arenas, up to cache_max): a burst of T concurrent callers leaves
        // at most this many behind
        v.idle_max = 4;
        // pieces the helper thread stays behind the copying thread (1: never a page in common in flight)
        v.helper_lag = 1;
        return v;
    }();
    return k;
}

void stage_path_note(StagePath which) { g_paths[which].fetch_add(1, std::memory_order_relaxed); }
std::uint64_t stage_path_count(int which) {
    return which >= 0 && which < kPathCount ? g_paths[which].load(std::memory_order_relaxed) : 0;
}

hipError_t counted_malloc(void **p, size_t bytes) {
    g_alloc_events.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc(p, bytes);
}
hipError_t counted_free(void *p) {
    g_alloc_events.fetch_add(1, std::memory_order_relaxed);
    return hipFree(p);
}
hipError_t counted_host_malloc(void **p, size_t bytes) {
    g_alloc_events.fetch_add(1, std::memory_order_relaxed);
    // coherent (fine-grained): kernels read what the CPU just copied in without any cache of theirs in between
    return hipHostMalloc(p, bytes, hipHostMallocCoherent);
}
hipError_t counted_host_free(void *p) {
    g_alloc_events.fetch_add(1, std::memory_order_relaxed);
    return hipHostFree(p);
}
hipError_t counted_malloc_async(void **p, size_t bytes, hipStream_t s) {
    g_alloc_events.fetch_add(1, std::memory_order_relaxed);
    return hipMallocAsync(p, bytes, s);
}
hipError_t counted_free_async(void *p, hipStream_t s) {
    g_alloc_events.fetch_add(1, std::memory_order_relaxed);
    return hipFreeAsync(p, s);
}
std::uint64_t alloc_event_count() { return g_alloc_events.load(std::memory_order_relaxed); }

// A bump arena over a few blocks.  A call that outgrows the cached block gets a further block (earlier regions stay
// valid); when the context is returned the blocks are merged into ONE block of the bytes the call actually asked for
// (never the sum of the cached block and the overflow blocks), so the next call of the same shape allocates nothing.
// Arenas above `cache_max` are not kept.
struct Arena {
    struct Block {
        void *p;
        size_t cap;
    };
    std::vector<Block> blocks;
    size_t used = 0;    // in blocks.back()
    size_t asked = 0;   // bytes handed out during this call (rounded), over all blocks
    bool pinned = false;

    hipError_t raw_alloc(void **p, size_t bytes) const { return pinned ? counted_host_malloc(p, bytes) : counted_malloc(p, bytes); }
    void raw_free(void *p) const { (void)(pinned ? counted_host_free(p) : counted_free(p)); }
    size_t total() const {
        size_t t = 0;
        for (const Block &b : blocks) t += b.cap;
        return t;
    }
    hipError_t get(size_t bytes, void **out) {
        bytes = round_up(bytes ? bytes : 1, kAlign);
        asked += bytes;
        if (!blocks.empty() && used + bytes <= blocks.back().cap) {
            *out = static_cast<char *>(blocks.back().p) + used;
            used += bytes;
            return hipSuccess;
        }
        void *p = nullptr;
        const hipError_t e = raw_alloc(&p, bytes);
        if (e != hipSuccess) return e;
        blocks.push_back(Block{p, bytes});
        used = bytes;
        *out = p;
        return hipSuccess;
    }
    // end of a call: one block of max(cached block, bytes this call asked for) (or nothing when that exceeds cache_max)
    void recycle(size_t cache_max) {
        used = 0;
        const size_t need = asked;
        asked = 0;
        if (blocks.size() <= 1 && total() <= cache_max) return;
        const size_t want = std::max(blocks.empty() ? (size_t)0 : blocks.front().cap, need);
        for (const Block &b : blocks) raw_free(b.p);
        blocks.clear();
        if (want > cache_max) return;
        void *p = nullptr;
        if (raw_alloc(&p, want) == hipSuccess) blocks.push_back(Block{p, want});
        else (void)hipGetLastError();
    }
    void release() {
        for (const Block &b : blocks) raw_free(b.p);
        blocks.clear();
        used = asked = 0;
    }
};

struct StageCtx {
    int device = 0;
    hipStream_t s[2] = {nullptr, nullptr};
    Arena dev, pin;
    struct Pending {
        void *host;
        const void *bounce;
        size_t bytes;
    };
    std::vector<Pending> pending;
    struct Range {
        char *p;
        size_t bytes;
        bool owned;  // (always false: nothing is registered here, see HostStage::pin)
    };
    std::vector<Range> registered;  // ranges of this call that lie in memory the caller pinned
    std::vector<hipEvent_t> events;  // pooled, timing disabled
    size_t events_used = 0;

    // helper thread: parked on `cv` between tasks
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, task_done = false, quit = false;
    int task_rc = PFHE_OK;

    void worker_loop() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_task || quit; });
            if (quit) return;
            std::function<int()> fn = std::move(task);
            has_task = false;
            lk.unlock();
            int rc;
            try {
                rc = fn();
            } catch (...) {
                rc = PFHE_ERR_HIP;
            }
            fn = nullptr;
            lk.lock();
            task_rc = rc;
            task_done = true;
            cv.notify_all();
        }
    }
    void stop_worker() {
        if (!worker.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        worker.join();
    }
};

namespace {
struct Pool {
    std::mutex mu;
    std::vector<std::vector<StageCtx *>> idle;  // per device
};
// never destroyed: at process exit the HIP runtime may already be gone
Pool &pool() {
    static Pool *p = new Pool();
    return *p;
}
}  // namespace

HostStage::HostStage(int device) {
    Pool &P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if ((size_t)device < P.idle.size() && !P.idle[device].empty()) {
            ctx_ = P.idle[device].back();
            P.idle[device].pop_back();
        }
    }
    if (ctx_) return;
    auto c = std::make_unique<StageCtx>();
    c->device = device;
    c->pin.pinned = true;
    for (hipStream_t &s : c->s) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            for (hipStream_t t : c->s)
                if (t) (void)hipStreamDestroy(t);
            return;
        }
    }
    ctx_ = c.release();
}

// Pins [host, host + bytes) in place for this call.  Memory that already is pinned (hipHostMalloc, or registered by the
// caller) needs nothing; a refusal (read-only mapping, pages held by another registration) sends the caller to the
// bounce / pageable path.
// Memory the CALLER has pinned (hipHostMalloc, hipHostRegister, a torch pinned tensor) is copied from / to by true
// asynchronous DMA; only driver-allocated pinned memory is also handed to kernels as it is (map() below).  Pageable memory
// is NOT registered here: round 4 tried
// (hipHostRegister on the slice for the duration of the call, 1.1 us on this platform, kernels reading and writing the
// mapped range) and got rare wrong words and host-heap corruption under a debugging allocator
// (profiles/r04_experiments.txt, item 6) — pageable slices go through the pool's own pinned buffer instead.
bool HostStage::pin(const void *host, size_t bytes, bool any_size) {
    const StageKnobs &K = stage_knobs();
    char *h = static_cast<char *>(const_cast<void *>(host));
    for (const StageCtx::Range &r : ctx_->registered)
        if (h >= r.p && h + bytes <= r.p + r.bytes) return true;  // pieces of a pinned slice; in-place downloads
    if (bytes == 0 || !K.use_register || (!any_size && bytes < K.register_min)) return false;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, h) != hipSuccess || at.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return false;
    }
    // The whole range must lie inside ONE pinned allocation / registration: pinned first and last bytes can belong to two
    // different ones with pageable pages in between (a slice spanning two registered buffers, a partly registered array),
    // and the mapped range is handed to kernels and copy engines as a whole.  The runtime reports start and size of the
    // range that contains an address (HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR / _RANGE_SIZE: the registration's own for
    // hipHostRegister memory, where hipMemGetAddressRange returns a null base); when it cannot, or the slice sticks out of
    // it, the slice is treated as pageable.
    hipDeviceptr_t base = nullptr;
    size_t extent = 0;
    if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, h) != hipSuccess ||
        hipPointerGetAttribute(&extent, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, h) != hipSuccess || base == nullptr ||
        h < static_cast<char *>(base) || h + bytes > static_cast<char *>(base) + extent) {
        (void)hipGetLastError();
        return false;
    }
    ctx_->registered.push_back(StageCtx::Range{h, bytes, false});
    return true;
}

// Device-side address of caller memory that KERNELS may read and write in place — only memory the DRIVER allocated pinned
// (hipHostMalloc; a torch pinned tensor).  Memory that is merely REGISTERED (hipHostRegister: a userptr mapping of pages the
// host kernel still owns) is not handed to kernels: tools/microbench12_register_hazard.hip — plain HIP, no code of this
// library — registers a heap block, runs a kernel on the mapped range and unregisters it, between pageable hipMemcpy's of the
// same block, and on four MI355X hosts out of five finds 2-16 blocks in 10 000 with wrong words (part of them the values
// from before the kernel, at reused heap addresses under MALLOC_CHECK_=3); never on hipHostMalloc memory, never through the
// copy engines (profiles/r05_microbench12_register_hazard.txt).  That is round 4's "rare wrong words" (r04_experiments.txt
// item 6), reproduced this round from the caller's side inside the full test suite (tools/hazard_suite_probe.sh).
// hipHostGetFlags tells the two apart on this runtime: it succeeds for hipHostMalloc memory and fails for a registration
// (tools/probe_host_kinds.hip; every other attribute — type, device pointer, range — reads the same for both).  A registered
// slice takes the pool's own pinned buffer (short) or the copy engines (long) instead.
// The check FAILS CLOSED.  (1) The discriminator is probed once per process on a block this library registers itself: if
// hipHostGetFlags SUCCEEDS for that registration (some HIP builds report the register flags), it cannot tell the two kinds
// apart here and no caller memory is ever mapped.  (2) A second, independent property must hold as well:
// hipMemGetAddressRange reports a non-null base only for memory the driver allocated (for a registration the base is null,
// see pin()).  (3) PFHE_STAGE_ZERO_COPY=0 turns the path off altogether.
// The probe block is two pages of its OWN anonymous mapping, and it stays mapped and registered for the life of the process:
// pages that have once been registered must never come back as somebody's pageable copy source (the defect located in round
// 5 lives exactly there — profiles/r05_experiments.txt item 7), which a heap block returned to malloc would.
static bool host_flags_tell_registrations_apart() {
    static const bool works = [] {
        void *blk = mmap(nullptr, 2 * 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (blk == MAP_FAILED) return false;
        if (hipHostRegister(blk, 2 * 4096, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            (void)munmap(blk, 2 * 4096);  // never registered: nothing to keep
            return false;
        }
        unsigned flags = 0;
        const bool ok = hipHostGetFlags(&flags, blk) != hipSuccess;  // must be refused for a registration
        (void)hipGetLastError();
        return ok;  // (the 8 KiB stay registered: see above)
    }();
    return works;
}

void *HostStage::map(void *host, size_t bytes) {
    if (!stage_knobs().zero_copy || !host_flags_tell_registrations_apart()) return nullptr;
    if (!pin(host, bytes, true)) return nullptr;
    unsigned flags = 0;
    if (hipHostGetFlags(&flags, host) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    hipDeviceptr_t base = nullptr;
    size_t extent = 0;
    if (hipMemGetAddressRange(&base, &extent, (hipDeviceptr_t)host) != hipSuccess || base == nullptr) {
        (void)hipGetLastError();
        return nullptr;
    }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return d;
}

// a region of the pool's pinned, coherent host buffer and its device-side address
void *HostStage::bounce(size_t bytes, void **dev) {
    void *b = nullptr;
    *dev = nullptr;
    if (ctx_->pin.get(bytes, &b) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipHostGetDevicePointer(dev, b, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return b;
}

// true when the first or the last byte of the range is pinned memory (for a range pin() has refused: it lies PARTLY inside
// registrations; the runtime copies such a range neither as pinned nor as pageable memory — see staged_copy)
bool HostStage::touches_pinned(const void *host, size_t bytes) {
    if (bytes == 0) return false;
    const char *h = static_cast<const char *>(host);
    for (const char *p : {h, h + bytes - 1}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) return true;
        (void)hipGetLastError();
    }
    return false;
}

void HostStage::unpin_all() { ctx_->registered.clear(); }

int HostStage::take_event(hipEvent_t *out) {
    if (ctx_->events_used == ctx_->events.size()) {
        hipEvent_t e = nullptr;
        PFHE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx_->events.push_back(e);
    }
    *out = ctx_->events[ctx_->events_used++];
    return PFHE_OK;
}

int HostStage::order(hipStream_t signaller, hipStream_t waiter) {
    hipEvent_t e = nullptr;
    PFHE_TRY(take_event(&e));
    PFHE_HIP(hipEventRecord(e, signaller));
    PFHE_HIP(hipStreamWaitEvent(waiter, e, 0));
    return PFHE_OK;
}

static void destroy_ctx(StageCtx *c) {
    c->stop_worker();
    DeviceGuard g(c->device);
    c->dev.release();
    c->pin.release();
    for (hipStream_t s : c->s)
        if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    delete c;
}

int HostStage::helper_start(std::function<int()> task, std::function<void()> on_abandon) {
    if (helper_busy_) return PFHE_ERR_BAD_ARGUMENT;
    StageCtx &c = *ctx_;
    if (!c.worker.joinable()) {
        try {
            c.worker = std::thread([ctx = ctx_] { ctx->worker_loop(); });
        } catch (...) {
            return PFHE_ERR_HIP;
        }
    }
    {
        std::lock_guard<std::mutex> lk(c.mu);
        c.task = std::move(task);
        c.has_task = true;
        c.task_done = false;
    }
    on_abandon_ = std::move(on_abandon);
    helper_busy_ = true;
    c.cv.notify_all();
    return PFHE_OK;
}

int HostStage::helper_wait() {
    if (!helper_busy_) return PFHE_OK;
    StageCtx &c = *ctx_;
    std::unique_lock<std::mutex> lk(c.mu);
    c.cv.wait(lk, [&] { return c.task_done; });
    helper_busy_ = false;
    on_abandon_ = nullptr;
    return c.task_rc;
}

HostStage::~HostStage() {
    if (!ctx_) return;
    if (helper_busy_) {  // an error path left while the helper's task may still reference the caller's frame
        if (on_abandon_) on_abandon_();
        (void)helper_wait();
    }
    if (dirty_) {
        for (hipStream_t s : ctx_->s) (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
    }
    unpin_all();
    ctx_->events_used = 0;
    ctx_->pending.clear();
    const StageKnobs &K = stage_knobs();
    ctx_->dev.recycle(K.cache_max);
    ctx_->pin.recycle(K.cache_max);
    Pool &P = pool();
    StageCtx *surplus = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if ((size_t)ctx_->device >= P.idle.size()) P.idle.resize(ctx_->device + 1);
        // idle contexts are capped per device: a burst of concurrent callers does not pin its arenas for the life of the
        // process (StageKnobs::idle_max, four)
        if (P.idle[ctx_->device].size() < K.idle_max) P.idle[ctx_->device].push_back(ctx_);
        else surplus = ctx_;
    }
    if (surplus) destroy_ctx(surplus);
}

hipStream_t HostStage::stream() const { return ctx_->s[0]; }
hipStream_t HostStage::stream2() const { return ctx_->s[1]; }

int HostStage::alloc(size_t bytes, void **dev) {
    *dev = nullptr;
    // whoever takes arena memory is about to queue work on the context's streams: from here on every exit path — a failed
    // launch or copy included — synchronises them before the arena is recycled and the context returns to the pool
    dirty_ = true;
    PFHE_HIP(ctx_->dev.get(bytes, dev));
    return PFHE_OK;
}

int HostStage::copy_in(void *dev, const void *host, size_t bytes, hipStream_t s) {
    if (bytes == 0) return PFHE_OK;
    if (!s) s = ctx_->s[0];
    dirty_ = true;
    if (pin(host, bytes)) {
        stage_path_note(kPathDmaCaller);
        PFHE_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        return PFHE_OK;
    }
    if (bytes <= stage_knobs().bounce_max) {
        stage_path_note(kPathBounceDma);
        void *b = nullptr;
        PFHE_HIP(ctx_->pin.get(bytes, &b));
        std::memcpy(b, host, bytes);
        PFHE_HIP(hipMemcpyAsync(dev, b, bytes, hipMemcpyHostToDevice, s));
        return PFHE_OK;
    }
    stage_path_note(kPathPageable);
    if (hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s) == hipSuccess) return PFHE_OK;
    (void)hipGetLastError();
    return staged_copy(dev, const_cast<void *>(host), bytes, true, s);
}

// Last resort for a range the runtime refuses to copy as one piece — a slice that lies partly inside registrations of the
// caller's (two adjacent registrations, a partly registered array) is neither pinned nor pageable to hipMemcpyAsync and comes
// back as "invalid argument": the CPU moves it through one chunk of the pool's pinned buffer at a time, synchronously.
int HostStage::staged_copy(void *dev, void *host, size_t bytes, bool to_device, hipStream_t s) {
    const size_t chunk = std::min(bytes, std::max<size_t>(stage_knobs().bounce_max, (size_t)64 << 10));
    void *b = nullptr;
    PFHE_HIP(ctx_->pin.get(chunk, &b));
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        if (to_device) {
            std::memcpy(b, static_cast<char *>(host) + off, n);
            PFHE_HIP(hipMemcpyAsync(static_cast<char *>(dev) + off, b, n, hipMemcpyHostToDevice, s));
            PFHE_HIP(hipStreamSynchronize(s));
        } else {
            PFHE_HIP(hipMemcpyAsync(b, static_cast<char *>(dev) + off, n, hipMemcpyDeviceToHost, s));
            PFHE_HIP(hipStreamSynchronize(s));
            std::memcpy(static_cast<char *>(host) + off, b, n);
        }
    }
    return PFHE_OK;
}

int HostStage::upload(const void *host, size_t bytes, void **dev) {
    PFHE_TRY(alloc(bytes, dev));
    return copy_in(*dev, host, bytes);
}

int HostStage::download(void *host, const void *dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return PFHE_OK;
    if (!s) s = ctx_->s[0];
    dirty_ = true;
    if (pin(host, bytes)) {
        PFHE_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        return PFHE_OK;
    }
    if (bytes <= stage_knobs().bounce_max) {
        void *b = nullptr;
        PFHE_HIP(ctx_->pin.get(bytes, &b));
        PFHE_HIP(hipMemcpyAsync(b, dev, bytes, hipMemcpyDeviceToHost, s));
        ctx_->pending.push_back(StageCtx::Pending{host, b, bytes});
        return PFHE_OK;
    }
    if (hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s) == hipSuccess) return PFHE_OK;
    (void)hipGetLastError();
    return staged_copy(const_cast<void *>(dev), host, bytes, false, s);
}

int HostStage::finish() {
    if (!dirty_) return PFHE_OK;
    for (hipStream_t s : ctx_->s) PFHE_HIP(hipStreamSynchronize(s));
    dirty_ = false;
    unpin_all();
    for (const StageCtx::Pending &p : ctx_->pending) std::memcpy(p.host, p.bounce, p.bytes);
    ctx_->pending.clear();
    return PFHE_OK;
}

int staging_release(int device) {
    Pool &P = pool();
    std::vector<StageCtx *> victims;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (size_t d = 0; d < P.idle.size(); ++d) {
            if (device >= 0 && (size_t)device != d) continue;
            victims.insert(victims.end(), P.idle[d].begin(), P.idle[d].end());
            P.idle[d].clear();
        }
    }
    for (StageCtx *c : victims) destroy_ctx(c);
    return (int)victims.size();
}

}  // namespace pfhe
