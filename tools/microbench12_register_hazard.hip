// microbench12_register_hazard.hip — is "register a heap block for one call, use it, unregister it" safe on this runtime when
// the same process ALSO copies from / to pageable memory at the same (reused) addresses?  No libpfhe code: plain HIP only.
//
// Background (profiles/r05_experiments.txt item 7): round 4's library registered the caller's pageable slice
// for a call and, rarely and only inside the full test suite, produced wrong words / a damaged heap; this round the same
// happens when the CALLER registers per call inside that suite (tools/hazard_suite_probe.sh), while tools/stress_host_slice.py
// (registered calls only, no pageable copies of the same blocks) never fails.  What the suite has and the stress tool does
// not: pageable hipMemcpy's of heap blocks (torch's .cuda() / .cpu(), the library's own pageable path), for which the
// runtime pins the pages itself.  This program mixes the three uses on heap blocks whose addresses are reused:
//   P  pageable: hipMemcpy H2D -> kernel on device memory -> hipMemcpy D2H            (runtime pins / stages by itself)
//   Z  zero-copy: hipHostRegister -> kernel on the mapped host pointer, in place -> hipHostUnregister
//   D  DMA: hipHostRegister -> hipMemcpyAsync H2D -> kernel -> hipMemcpyAsync D2H -> sync -> hipHostUnregister
// and checks every word.  Build: hipcc --offload-arch=gfx950 -O2 -pthread -o microbench12 microbench12_register_hazard.hip
//   H  (control) every block from hipHostMalloc, zero-copy kernel on it — memory the driver allocated and really pinned
// Run:   ./microbench12 [iterations=20000] [modes=PZD] [seed=1] [max_mib=24] [churn_threads=2]      (also under MALLOC_CHECK_=3)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            printf("HIP error %d (%s) at line %d, iteration %ld\n", (int)e_, hipGetErrorString(e_), __LINE__, g_it); \
            fflush(stdout);                                                                            \
            return 2;                                                                                  \
        }                                                                                              \
    } while (0)

static long g_it = 0;

__global__ void scramble(uint64_t *p, size_t n, uint64_t k) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = p[i] * 0x9E3779B97F4A7C15ull + k;
}

static uint64_t mix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// host threads that grow / trim the heap and map / unmap large blocks meanwhile, checking their own blocks for damage
static void churn(std::atomic<int> *stop, std::atomic<long> *damaged, uint64_t seed) {
    std::vector<std::pair<unsigned char *, size_t>> held;
    uint64_t s = seed;
    while (!stop->load()) {
        const size_t bytes = (mix(s) % 3 == 0) ? (size_t)(mix(s) % (24u << 20)) + 1 : (size_t)(mix(s) % 200000) + 1;
        unsigned char *b = (unsigned char *)malloc(bytes);
        if (!b) continue;
        memset(b, (int)(bytes & 0xFF), bytes);
        held.push_back({b, bytes});
        if (held.size() > 24) {
            const size_t j = mix(s) % held.size();
            const unsigned char want = (unsigned char)(held[j].second & 0xFF);
            for (size_t i = 0; i < held[j].second; i += 509)
                if (held[j].first[i] != want) {
                    damaged->fetch_add(1);
                    break;
                }
            free(held[j].first);
            held[j] = held.back();
            held.pop_back();
        }
    }
    for (auto &h : held) free(h.first);
}

// page-migration counters of the host kernel (/proc/vmstat): a registration is a userptr mapping kept valid through MMU
// notifiers, not a hard pin — the kernel may still migrate the pages (compaction, THP collapse, NUMA balancing)
static void vmstat(const char *when) {
    FILE *f = fopen("/proc/vmstat", "r");
    if (!f) return;
    char name[128];
    unsigned long long v;
    printf("vmstat %s:", when);
    while (fscanf(f, "%127s %llu", name, &v) == 2)
        for (const char *k : {"pgmigrate_success", "numa_pages_migrated", "numa_hint_faults", "thp_collapse_alloc", "thp_split_page",
                              "compact_migrate_scanned", "thp_fault_alloc"})
            if (!strcmp(name, k)) printf(" %s=%llu", name, v);
    printf("\n");
    fclose(f);
}

int main(int argc, char **argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 20000;
    const char *modes = argc > 2 ? argv[2] : "PZD";
    uint64_t s = argc > 3 ? (uint64_t)atoll(argv[3]) : 1;
    const size_t max_bytes = (size_t)(argc > 4 ? atol(argv[4]) : 24) << 20;
    const int churners = argc > 5 ? atoi(argv[5]) : 2;
    const size_t nmodes = strlen(modes);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint64_t *dev;
    CK(hipMalloc(&dev, max_bytes + 8192));  // sizes are rounded up by up to 4 KiB below
    std::atomic<int> stop{0};
    std::atomic<long> damaged{0};
    std::vector<std::thread> ts;
    for (int i = 0; i < churners; ++i) ts.emplace_back(churn, &stop, &damaged, 1000 + i);
    long bad = 0, count[3] = {0, 0, 0}, refused = 0;
    vmstat("before");
    for (g_it = 0; g_it < iters; ++g_it) {
        // sizes as the test suite's: many of one polynomial (8 KiB ... 1.5 MiB), some long (up to max)
        size_t bytes = (mix(s) % 4 == 0) ? (size_t)(mix(s) % max_bytes) : (size_t)(mix(s) % (3u << 19));
        bytes = (bytes + 4096) & ~(size_t)7;
        const size_t n = bytes / 8;
        const bool host_malloc = strchr(modes, 'H') != nullptr;  // every block from hipHostMalloc instead of malloc (control)
        uint64_t *h = nullptr;
        if (host_malloc)
            CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
        else
            h = (uint64_t *)malloc(bytes);
        if (!h) return 3;
        const uint64_t tag = mix(s);
        for (size_t i = 0; i < n; ++i) h[i] = tag + i;
        // the same block goes through one to four uses before it is freed (a numpy array of the suite is copied by torch,
        // transformed in place through a registration, copied again ...); every use is checked
        const int uses = 1 + (int)(mix(s) % 4);
        uint64_t expect_mul = 1, expect_add = 0;  // h[i] == (tag + i) * expect_mul + expect_add
        for (int u = 0; u < uses; ++u) {
            const uint64_t k = mix(s);
            char m = modes[mix(s) % nmodes];
            bool registered = false;
            if (m == 'H') m = 'Z';
            if (m != 'P' && !host_malloc) {
                if (hipHostRegister(h, bytes, hipHostRegisterDefault) == hipSuccess)
                    registered = true;
                else {
                    (void)hipGetLastError();
                    ++refused;
                    m = 'P';
                }
            }
            const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
            if (m == 'P') {
                // the NULL stream, as torch's .cuda() / .cpu() on a numpy array
                CK(hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(scramble, dim3(grid), dim3(256), 0, 0, dev, n, k);
                CK(hipGetLastError());
                CK(hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost));
                ++count[0];
            } else if (m == 'Z') {
                uint64_t *mapped = nullptr;
                CK(hipHostGetDevicePointer((void **)&mapped, h, 0));
                hipLaunchKernelGGL(scramble, dim3(grid), dim3(256), 0, st, mapped, n, k);
                CK(hipGetLastError());
                CK(hipStreamSynchronize(st));
                ++count[1];
            } else {
                CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(scramble, dim3(grid), dim3(256), 0, st, dev, n, k);
                CK(hipGetLastError());
                CK(hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                ++count[2];
            }
            if (registered) CK(hipHostUnregister(h));
            const uint64_t prev_mul = expect_mul, prev_add = expect_add;
            expect_mul *= 0x9E3779B97F4A7C15ull;
            expect_add = expect_add * 0x9E3779B97F4A7C15ull + k;
            size_t wrong = 0, first = 0, last = 0, stale = 0;
            for (size_t i = 0; i < n; ++i)
                if (h[i] != (tag + i) * expect_mul + expect_add) {
                    if (!wrong) first = i;
                    last = i;
                    ++wrong;
                    if (h[i] == (tag + i) * prev_mul + prev_add) ++stale;  // the word as it was BEFORE this use: the write is missing
                }
            if (wrong) {
                ++bad;
                u = uses;  // the block is wrong from here on: one report per block
                const uintptr_t a0 = (uintptr_t)(h + first), a1 = (uintptr_t)(h + last);
                printf("MISMATCH iteration %ld use %d mode %c bytes %zu: %zu wrong words in [%zu, %zu] = host %p..%p (page offsets 0x%lx..0x%lx, %s), "
                       "%zu of them still hold the value from before this use\n",
                       g_it, u, m, bytes, wrong, first, last, (void *)a0, (void *)a1, (unsigned long)(a0 & 4095), (unsigned long)(a1 & 4095),
                       (a0 >> 12) == (a1 >> 12) ? "one 4 KiB page" : "several pages", stale);
                fflush(stdout);
            }
        }
        if (host_malloc)
            CK(hipHostFree(h));
        else
            free(h);
    }
    stop.store(1);
    for (auto &t : ts) t.join();
    vmstat("after");
    printf("%ld iterations (pageable %ld, zero-copy %ld, registered DMA %ld; %ld registrations refused), %ld mismatching blocks, "
           "%ld damaged churn blocks, modes %s\n",
           iters, count[0], count[1], count[2], refused, bad, damaged.load(), modes);
    return bad || damaged.load() ? 1 : 0;
}
