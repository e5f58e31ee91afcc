// microbench2.hip — memory-system calibration with high memory-level parallelism.
//  (1) streaming read+write (in place), 8 x 16 B in flight per thread, vs working-set size
//  (2) "fused two-phase" pattern: every workgroup owns a 512 KiB chunk, transforms it twice
//      (read+write, barrier, read+write) — does the second phase hit L2 / Infinity Cache?
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using u64 = unsigned long long;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1);} } while (0)

// each thread handles 8 vectors at stride `span` (coalesced across threads)
__global__ __launch_bounds__(256) void rmw8(ulonglong2 *p, size_t nvec) {
    const size_t span = nvec / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += (size_t)gridDim.x * blockDim.x) {
        ulonglong2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * span];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k].x += k; v[k].y ^= v[k].x; }
#pragma unroll
        for (int k = 0; k < 8; ++k) p[i + k * span] = v[k];
    }
}
__global__ __launch_bounds__(256) void rd8(const ulonglong2 *p, size_t nvec, u64 *sink) {
    const size_t span = nvec / 8;
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += (size_t)gridDim.x * blockDim.x) {
        ulonglong2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * span];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ __launch_bounds__(256) void wr8(ulonglong2 *p, size_t nvec) {
    const size_t span = nvec / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 8; ++k) p[i + k * span] = ulonglong2{i, (u64)k};
    }
}

// one workgroup per 512 KiB chunk (persistent over chunks): PHASES passes of read+modify+write
template <int PHASES>
__global__ __launch_bounds__(1024) void chunk_phases(ulonglong2 *p, size_t nchunks) {
    constexpr size_t CH = (512 << 10) / 16;  // vectors per chunk
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        ulonglong2 *q = p + c * CH;
#pragma unroll
        for (int ph = 0; ph < PHASES; ++ph) {
            // phase ph: thread t touches vectors t + 1024*k (k<32): 32 x 16 B per thread, in 4 groups of 8
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ulonglong2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = q[threadIdx.x + 1024 * (g * 8 + k)];
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[k].x += ph; v[k].y ^= v[k].x; }
#pragma unroll
                for (int k = 0; k < 8; ++k) q[(threadIdx.x ^ (ph * 37)) % 1024 + 1024 * (g * 8 + k)] = v[k];
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

template <class F>
static float time_ms(F &&f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    u64 *sink; CK(hipMalloc(&sink, 8));
    const size_t sizes_mb[] = {16, 32, 64, 128, 192, 256, 512, 2048, 6144};
    for (size_t mb : sizes_mb) {
        size_t bytes = mb << 20; ulonglong2 *buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) break;
        CK(hipMemset(buf, 1, bytes));
        size_t nvec = bytes / 16; int reps = mb <= 256 ? 50 : 8;
        for (int blocks : {2048, 8192}) {
            float r = time_ms([&] { hipLaunchKernelGGL(rd8, dim3(blocks), dim3(256), 0, 0, buf, nvec, sink); }, reps);
            float w = time_ms([&] { hipLaunchKernelGGL(wr8, dim3(blocks), dim3(256), 0, 0, buf, nvec); }, reps);
            float m = time_ms([&] { hipLaunchKernelGGL(rmw8, dim3(blocks), dim3(256), 0, 0, buf, nvec); }, reps);
            std::printf("%5zu MiB grid %5d: read %7.1f GB/s  write %7.1f GB/s  rmw %7.1f GB/s (r+w)\n", mb, blocks,
                        bytes / r * 1e-6, bytes / w * 1e-6, 2.0 * bytes / m * 1e-6);
        }
        CK(hipFree(buf));
    }
    {
        size_t bytes = 6144ull << 20; ulonglong2 *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
        size_t nchunks = bytes / (512 << 10);
        for (int blocks : {256, 512}) {
            float t1 = time_ms([&] { hipLaunchKernelGGL(chunk_phases<1>, dim3(blocks), dim3(1024), 0, 0, buf, nchunks); }, 5);
            float t2 = time_ms([&] { hipLaunchKernelGGL(chunk_phases<2>, dim3(blocks), dim3(1024), 0, 0, buf, nchunks); }, 5);
            std::printf("chunk phases (6 GiB, %d WGs of 1024): 1 phase %.3f ms (%.0f GB/s r+w), 2 phases %.3f ms => second phase costs %.3f ms (%.0f GB/s)\n",
                        blocks, t1, 2.0 * bytes / t1 * 1e-6, t2, t2 - t1, 2.0 * bytes / (t2 - t1) * 1e-6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
