// pfhe_staging.cpp — per-device pool of staging contexts for the host-pointer entry points (pfhe_staging.hpp).
#include "pfhe_staging.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace pfhe {

namespace {
std::atomic<std::uint64_t> g_alloc_events{0};

size_t env_bytes(const char *name, size_t dflt) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const unsigned long long x = std::strtoull(v, &end, 10);
    return end == v ? dflt : (size_t)x;
}
constexpr size_t kAlign = 256;
size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace

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
// valid); when the context is returned the blocks are merged into ONE of the summed size, so the next call of the same
// shape allocates nothing.  Arenas above `cache_max` are not kept.
struct Arena {
    struct Block {
        void *p;
        size_t cap;
    };
    std::vector<Block> blocks;
    size_t used = 0;  // in blocks.back()
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
    // end of a call: one block of the summed size (or nothing when that exceeds cache_max)
    void recycle(size_t cache_max) {
        used = 0;
        if (blocks.size() <= 1 && total() <= cache_max) return;
        const size_t want = total();
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
        used = 0;
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
};

namespace {
struct Pool {
    std::mutex mu;
    std::vector<std::vector<StageCtx *>> idle;  // per device
    size_t bounce_max, cache_max, register_min;
    bool use_register;
    Pool() {
        // slices of at least register_min bytes are looked up: memory the caller pinned is used as it is
        // (PFHE_STAGE_REGISTER=0: never)
        use_register = env_bytes("PFHE_STAGE_REGISTER", 1) != 0;
        register_min = env_bytes("PFHE_STAGE_REGISTER_MIN", (size_t)128 << 10);
        // transfers up to bounce_max go through the pinned bounce buffer (a CPU copy + a true asynchronous DMA);
        // larger ones are handed to the runtime as they are (it pins the caller's pages in pieces)
        bounce_max = env_bytes("PFHE_STAGE_BOUNCE_MAX", (size_t)1 << 20);
        cache_max = env_bytes("PFHE_STAGE_CACHE_MAX", (size_t)2 << 30);
    }
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
// asynchronous DMA and may be handed to kernels as it is.  Pageable memory is NOT registered here: round 4 tried
// (hipHostRegister on the slice for the duration of the call, 1.1 us on this platform, kernels reading and writing the
// mapped range) and got rare wrong words and host-heap corruption under a debugging allocator
// (profiles/r04_experiments.txt, item 6) — pageable slices go through the pool's own pinned buffer instead.
bool HostStage::pin(const void *host, size_t bytes, bool any_size) {
    const Pool &P = pool();
    char *h = static_cast<char *>(const_cast<void *>(host));
    for (const StageCtx::Range &r : ctx_->registered)
        if (h >= r.p && h + bytes <= r.p + r.bytes) return true;  // pieces of a pinned slice; in-place downloads
    if (!P.use_register || (!any_size && bytes < P.register_min)) return false;
    hipPointerAttribute_t at{}, last{};
    if (hipPointerGetAttributes(&at, h) != hipSuccess || at.type != hipMemoryTypeHost ||
        hipPointerGetAttributes(&last, h + bytes - 1) != hipSuccess || last.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return false;
    }
    ctx_->registered.push_back(StageCtx::Range{h, bytes, false});
    return true;
}

void *HostStage::map(void *host, size_t bytes) {
    if (!pin(host, bytes, true)) return nullptr;
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

HostStage::~HostStage() {
    if (!ctx_) return;
    if (dirty_) {
        for (hipStream_t s : ctx_->s) (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
    }
    unpin_all();
    ctx_->events_used = 0;
    ctx_->pending.clear();
    Pool &P = pool();
    ctx_->dev.recycle(P.cache_max);
    ctx_->pin.recycle(P.cache_max);
    std::lock_guard<std::mutex> lk(P.mu);
    if ((size_t)ctx_->device >= P.idle.size()) P.idle.resize(ctx_->device + 1);
    P.idle[ctx_->device].push_back(ctx_);
}

hipStream_t HostStage::stream() const { return ctx_->s[0]; }
hipStream_t HostStage::stream2() const { return ctx_->s[1]; }

int HostStage::alloc(size_t bytes, void **dev) {
    *dev = nullptr;
    PFHE_HIP(ctx_->dev.get(bytes, dev));
    return PFHE_OK;
}

int HostStage::copy_in(void *dev, const void *host, size_t bytes, hipStream_t s) {
    if (bytes == 0) return PFHE_OK;
    if (!s) s = ctx_->s[0];
    dirty_ = true;
    if (pin(host, bytes)) {
        PFHE_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        return PFHE_OK;
    }
    if (bytes <= pool().bounce_max) {
        void *b = nullptr;
        PFHE_HIP(ctx_->pin.get(bytes, &b));
        std::memcpy(b, host, bytes);
        PFHE_HIP(hipMemcpyAsync(dev, b, bytes, hipMemcpyHostToDevice, s));
        return PFHE_OK;
    }
    PFHE_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
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
    if (bytes <= pool().bounce_max) {
        void *b = nullptr;
        PFHE_HIP(ctx_->pin.get(bytes, &b));
        PFHE_HIP(hipMemcpyAsync(b, dev, bytes, hipMemcpyDeviceToHost, s));
        ctx_->pending.push_back(StageCtx::Pending{host, b, bytes});
        return PFHE_OK;
    }
    PFHE_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
    return PFHE_OK;
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
    for (StageCtx *c : victims) {
        DeviceGuard g(c->device);
        c->dev.release();
        c->pin.release();
        for (hipStream_t s : c->s)
            if (s) (void)hipStreamDestroy(s);
        for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
        delete c;
    }
    return (int)victims.size();
}

}  // namespace pfhe
