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
// How the bytes travel (tools/microbench10_host.hip, profiles/r04_microbench10_host_path.txt): slices of at least
// 128 KiB are pinned IN PLACE for the duration of the call (hipHostRegister + hipHostUnregister cost 1.1 us together on
// this platform, against 9 + 18 us of CPU copies into and out of a bounce buffer for 512 KiB) and copied by true
// asynchronous DMA; smaller ones, and memory the runtime refuses to pin (read-only mappings, pages another call holds),
// go through the pinned bounce buffer; PFHE_STAGE_REGISTER=0 switches the pinning off.
#pragma once
#include <cstdint>

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
    // waits for everything queued on the context's streams and completes the downloads
    int finish();
    // stream `waiter` waits for everything queued so far on stream `signaller` (pooled events, no allocation)
    int order(hipStream_t signaller, hipStream_t waiter);

    // Pins [host, host + bytes) in place until the end of the call (hipHostRegister) so that copies from / to it are
    // true asynchronous DMA.  False when the range is too small to be worth it or the runtime refuses (read-only
    // mapping, pages held by another registration): copies of it then go through the bounce buffer / the runtime's
    // pageable path.  copy_in / download call it themselves; a caller that copies a slice piece by piece pins it whole.
    bool pin(const void *host, size_t bytes, bool any_size = false);
    // pin() + the device-side address of the pinned range (kernels then read / write the caller's memory over the
    // link themselves: no copy engine, no bounce); null when the range cannot be pinned
    void *map(void *host, size_t bytes);
    // marks the call as having queued work on the context's streams (kernels on mapped memory)
    void touch() { dirty_ = true; }

  private:
    void unpin_all();
    StageCtx *ctx_ = nullptr;
    bool dirty_ = false;
};

// frees the idle contexts of `device` (-1: every device); returns how many were released
int staging_release(int device);

}  // namespace pfhe
