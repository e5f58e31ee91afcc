// pfhe_staging.hpp — staging of the HOST-pointer (`*_slice`, `*_to`) entry points.
//
// The reference's methods take `&self, &mut [T]` and work in place without allocating (table.rs:541-563; SURVEY §8b
// "no hidden allocation per call").  The host-pointer forms of the C ABI therefore must not hipMalloc / hipFree per
// call either: every call borrows a STAGING CONTEXT from a per-device pool — two private non-blocking streams, a cached
// device arena grown on demand, a pinned host bounce buffer for small transfers, a few events — and returns it on exit.
// In steady state (same or smaller sizes as before) a call makes no allocation of any kind: pfhe_debug_alloc_count()
// does not move.  Contexts are handed out under a mutex, one per concurrent call, so any number of threads may call
// through one table handle at the same time (NttTable: Send + Sync).
//
// How the bytes travel (tools/microbench10_host.hip, profiles/r04_microbench10_host_path.txt): a slice of one piece is
// copied by the CPU into the pool's pinned buffer (9 us per 512 KiB), transformed by kernels that read and write that
// buffer over the link themselves (no copy engine: each copy-engine operation costs ~10 us of latency), and copied back
// (10-18 us); longer slices are handed to the runtime's pageable copies (it pins the caller's pages in pieces).  Memory the
// CALLER pinned is used as it is.  Registering the caller's pageable memory for the duration of a call — 1.1 us on this
// platform, and 45 us per 2^16-point transform — was built and withdrawn: rare wrong words and heap corruption
// (profiles/r04_experiments.txt, item 6).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

#include "pfhe_common.hpp"

namespace pfhe {

// Every device / pinned allocation and free the library makes goes through these (pfhe_debug_alloc_count).
hipError_t counted_malloc(void **p, size_t bytes);
hipError_t counted_free(void *p);
hipError_t counted_host_malloc(void **p, size_t bytes);
hipError_t counted_host_free(void *p);
hipError_t counted_malloc_async(void **p, size_t bytes, hipStream_t s);
hipError_t counted_free_async(void *p, hipStream_t s);
std::uint64_t alloc_event_count();

// Knobs of the staging layer.  Three can be set from the environment, parsed ONCE by one parser (a malformed value gives the
// built-in default): PFHE_STAGE_ZERO_COPY (0: copy engines only, no kernel touches host memory), PFHE_STAGE_BOUNCE_MAX,
// PFHE_STAGE_CHUNK; the rest are constants (until round 5 each had a switch of its own).
struct StageKnobs {
    size_t bounce_max, cache_max, register_min, chunk_bytes, pieces, idle_max, helper_lag;
    bool use_register, zero_copy, helper_thread;
};
const StageKnobs &stage_knobs();

// Which way the bytes of host-pointer calls travelled (tests assert the path they mean to exercise):
enum StagePath : int {
    kPathMappedCaller = 0,  // kernels read / wrote memory the CALLER allocated pinned (hipHostMalloc, torch pinned) — never a hipHostRegister registration (HostStage::map)
    kPathMappedBounce = 1,  // kernels read / wrote the pool's own pinned buffer (CPU copies either side)
    kPathDmaCaller = 2,     // copy engines on memory the caller pinned
    kPathPageable = 3,      // the runtime's pageable copies
    kPathHelper = 4,        // long pageable slice, copy back on the context's helper thread
    kPathBounceDma = 5,     // a CPU copy into the pool's pinned buffer + a copy-engine transfer from it (small uploads)
    kPathCount = 6
};
void stage_path_note(StagePath which);
std::uint64_t stage_path_count(int which);

struct StageCtx;

// One host-pointer call.  Usage: HostStage st(device); st.upload / st.alloc ...; launches on st.stream();
// st.download(...); st.finish().  All copies are asynchronous on the context's stream; finish() waits once and
// completes the bounce copies.  The destructor waits too if finish() was not reached (error paths).
class HostStage {
  public:
    explicit HostStage(int device);
    ~HostStage();
    HostStage(const HostStage &) = delete;
    HostStage &operator=(const HostStage &) = delete;
    bool ok() const { return ctx_ != nullptr; }
    hipStream_t stream() const;
    hipStream_t stream2() const;  // second stream of the context (chunk pipelines)
    // a region of the device arena (256-byte aligned); contents undefined
    int alloc(size_t bytes, void **dev);
    // region + asynchronous copy of `bytes` from host memory
    int upload(const void *host, size_t bytes, void **dev);
    // asynchronous copy into an existing region (on stream `s`, default: stream())
    int copy_in(void *dev, const void *host, size_t bytes, hipStream_t s = nullptr);
    // asynchronous copy back to host memory, complete after finish()
    int download(void *host, const void *dev, size_t bytes, hipStream_t s = nullptr);
    // a range the runtime refuses as one piece (partly registered): chunk by chunk through the pool's pinned buffer
    int staged_copy(void *dev, void *host, size_t bytes, bool to_device, hipStream_t s);
    bool touches_pinned(const void *host, size_t bytes);
    // waits for everything queued on the context's streams and completes the downloads
    int finish();
    // stream `waiter` waits for everything queued so far on stream `signaller` (pooled events, no allocation)
    int order(hipStream_t signaller, hipStream_t waiter);
    // a pooled event (timing disabled), the context's until the end of the call
    int take_event(hipEvent_t *out);

    // True when [host, host + bytes) lies inside ONE pinned allocation or registration of the CALLER's (hipHostMalloc,
    // hipHostRegister, a torch pinned tensor): copy engines then read / write it directly (true asynchronous DMA).  KERNELS
    // are handed such memory only through map(), which accepts driver-allocated memory and refuses registrations.
    // Pageable memory is never registered here (see the note on top): it goes through the bounce buffer or the runtime's
    // pageable path.  copy_in / download call it themselves.
    bool pin(const void *host, size_t bytes, bool any_size = false);
    // a region of the pool's pinned, coherent host buffer (valid until the end of the call) and its device-side address
    void *bounce(size_t bytes, void **dev);
    // the device-side address of a range in caller-pinned memory (kernels then read / write it over the link
    // themselves); null for pageable memory
    void *map(void *host, size_t bytes);
    // marks the call as having queued work on the context's streams (kernels on mapped memory)
    void touch() { dirty_ = true; }

    // The context's HELPER THREAD (started on first use, parked on a condition variable between calls, never spawned per
    // call): helper_start hands it one task; helper_wait blocks until the task has returned and yields its status.  At
    // most one task per call.  If a call leaves without helper_wait (an error path, an exception), the destructor runs
    // `on_abandon` (which must make the task return promptly) and waits for the task before the context goes back to the
    // pool — the task may reference the caller's stack.  A task that throws yields PFHE_ERR_HIP; nothing reaches
    // std::terminate.
    int helper_start(std::function<int()> task, std::function<void()> on_abandon);
    int helper_wait();
    bool helper_busy() const { return helper_busy_; }

  private:
    void unpin_all();
    StageCtx *ctx_ = nullptr;
    bool dirty_ = false;
    bool helper_busy_ = false;
    std::function<void()> on_abandon_;
};

// frees the idle contexts of `device` (-1: every device); returns how many were released
int staging_release(int device);

}  // namespace pfhe
