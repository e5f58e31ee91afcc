// microbench10_host.hip — where the time of a host-pointer (`*_slice`) call goes: CPU copies into / out of pinned
// memory, SDMA copies, kernel launch + synchronise, kernels that read / write pinned host memory directly (zero-copy),
// hipHostRegister, and two host threads copying in opposite directions (is the link used full duplex?).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench10_host.hip -o tools/microbench10 -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); }

__global__ void empty_kernel() {}
// dst[i] = src[i] + 1 on 16-byte vectors: src / dst may be pinned host memory
__global__ void copy_kernel(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ulonglong2 v = src[i];
        v.x += 1;
        dst[i] = v;
    }
}

template <class F>
static double timeit(int reps, F &&f) {
    f();
    f();
    double best = 1e30, sum = 0;
    for (int r = 0; r < reps; ++r) {
        const auto t0 = clk::now();
        f();
        const double t = us_since(t0);
        best = t < best ? t : best;
        sum += t;
    }
    std::printf("(avg %.1f) ", sum / reps);
    return best;
}

int main() {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (size_t bytes : {(size_t)32 << 10, (size_t)512 << 10, (size_t)1536 << 10, (size_t)24 << 20}) {
        std::printf("---- %zu KiB ----\n", bytes >> 10);
        void *pin_in, *pin_out, *dev, *dev2;
        CK(hipHostMalloc(&pin_in, bytes, hipHostMallocDefault));
        CK(hipHostMalloc(&pin_out, bytes, hipHostMallocDefault));
        CK(hipMalloc(&dev, bytes));
        CK(hipMalloc(&dev2, bytes));
        std::vector<char> user(bytes, 1), user2(bytes, 2);
        const size_t nvec = bytes / 16;
        const dim3 grid((unsigned)((nvec + 255) / 256)), block(256);
        const int reps = bytes > ((size_t)4 << 20) ? 20 : 200;
        std::printf("memcpy user -> pinned              : %8.1f us\n", timeit(reps, [&] { std::memcpy(pin_in, user.data(), bytes); }));
        std::printf("empty kernel + sync                : %8.1f us\n", timeit(reps, [&] {
                        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("two empty kernels + sync           : %8.1f us\n", timeit(reps, [&] {
                        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
                        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("H2D async (pinned) + sync          : %8.1f us\n", timeit(reps, [&] {
                        CK(hipMemcpyAsync(dev, pin_in, bytes, hipMemcpyHostToDevice, s));
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("D2H async (pinned) + sync          : %8.1f us\n", timeit(reps, [&] {
                        CK(hipMemcpyAsync(pin_out, dev, bytes, hipMemcpyDeviceToHost, s));
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("H2D + kernel + D2H (pinned) + sync : %8.1f us\n", timeit(reps, [&] {
                        CK(hipMemcpyAsync(dev, pin_in, bytes, hipMemcpyHostToDevice, s));
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dev, (ulonglong2 *)dev2, nvec);
                        CK(hipMemcpyAsync(pin_out, dev2, bytes, hipMemcpyDeviceToHost, s));
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("kernel pinned -> device + sync     : %8.1f us\n", timeit(reps, [&] {
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)pin_in, (ulonglong2 *)dev, nvec);
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("kernel device -> pinned + sync     : %8.1f us\n", timeit(reps, [&] {
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dev, (ulonglong2 *)pin_out, nvec);
                        CK(hipStreamSynchronize(s));
                    }));
        std::printf("kernel pinned->dev, dev->pinned+sync: %7.1f us\n", timeit(reps, [&] {
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)pin_in, (ulonglong2 *)dev, nvec);
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dev, (ulonglong2 *)pin_out, nvec);
                        CK(hipStreamSynchronize(s));
                    }));
        {   // CPU copy out of a pinned buffer the GPU has just written (its lines are not in the CPU's caches)
            double best = 1e30;
            for (int r = 0; r < 20; ++r) {
                hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dev, (ulonglong2 *)pin_out, nvec);
                CK(hipStreamSynchronize(s));
                const auto t0 = clk::now();
                std::memcpy(user.data(), pin_out, bytes);
                const double t = us_since(t0);
                best = t < best ? t : best;
            }
            std::printf("memcpy pinned(GPU-written) -> user : %8.1f us\n", best);
        }
        std::printf("hipMemcpy pageable H2D             : %8.1f us\n", timeit(reps, [&] { CK(hipMemcpy(dev, user.data(), bytes, hipMemcpyHostToDevice)); }));
        std::printf("hipMemcpy pageable D2H             : %8.1f us\n", timeit(reps, [&] { CK(hipMemcpy(user.data(), dev, bytes, hipMemcpyDeviceToHost)); }));
        std::printf("hipHostRegister + Unregister       : %8.1f us\n", timeit(reps > 50 ? 50 : reps, [&] {
                        CK(hipHostRegister(user.data(), bytes, hipHostRegisterDefault));
                        CK(hipHostUnregister(user.data()));
                    }));
        {   // hipHostRegister on the caller's pageable buffer: zero-copy kernels and true asynchronous DMA on it
            std::vector<char> fresh(bytes + 4096, 3);
            char *up = fresh.data() + 64;  // not page aligned, like a caller's slice
            std::printf("register + kernel user->dev + kernel dev->user + sync + unregister: %8.1f us\n", timeit(reps > 50 ? 50 : reps, [&] {
                            CK(hipHostRegister(up, bytes, hipHostRegisterDefault));
                            void *dp = nullptr;
                            CK(hipHostGetDevicePointer(&dp, up, 0));
                            hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dp, (ulonglong2 *)dev, nvec);
                            hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dev, (ulonglong2 *)dp, nvec);
                            CK(hipStreamSynchronize(s));
                            CK(hipHostUnregister(up));
                        }));
            // correctness: every byte went 3 -> +1 (per 64-bit x lane) twice per rep
            unsigned long long first;
            std::memcpy(&first, up, 8);
            std::printf("   (first word after the reps: %llx)\n", first);
            std::printf("register + H2D + kernel + D2H + sync + unregister                 : %8.1f us\n", timeit(reps > 50 ? 50 : reps, [&] {
                            CK(hipHostRegister(up, bytes, hipHostRegisterDefault));
                            CK(hipMemcpyAsync(dev, up, bytes, hipMemcpyHostToDevice, s));
                            hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dev, (ulonglong2 *)dev2, nvec);
                            CK(hipMemcpyAsync(up, dev2, bytes, hipMemcpyDeviceToHost, s));
                            CK(hipStreamSynchronize(s));
                            CK(hipHostUnregister(up));
                        }));
            // a buffer never touched by the GPU before, registered once each (first-touch cost of the pinning)
            double sum = 0;
            for (int r = 0; r < 10; ++r) {
                std::vector<char> cold(bytes + 4096, 5);
                const auto t0 = clk::now();
                CK(hipHostRegister(cold.data() + 64, bytes, hipHostRegisterDefault));
                void *dp = nullptr;
                CK(hipHostGetDevicePointer(&dp, cold.data() + 64, 0));
                hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)dp, (ulonglong2 *)dp, nvec);
                CK(hipStreamSynchronize(s));
                CK(hipHostUnregister(cold.data() + 64));
                sum += us_since(t0);
            }
            std::printf("fresh buffer: register + in-place kernel + sync + unregister       : %8.1f us (avg of 10)\n", sum / 10);
            // duplex on registered memory: pieces alternate between two streams
            CK(hipHostRegister(up, bytes, hipHostRegisterDefault));
            for (size_t pieces : {(size_t)1, (size_t)2, (size_t)4, (size_t)8}) {
                const size_t pb = bytes / pieces / 4096 * 4096;
                if (pb == 0) continue;
                std::printf("registered, %zu pieces on 2 streams, H2D + kernel + D2H each      : %8.1f us\n", pieces, timeit(reps > 50 ? 50 : reps, [&] {
                                for (size_t i = 0; i < pieces; ++i) {
                                    hipStream_t st = (i & 1) ? s2 : s;
                                    char *d1 = (char *)dev + i * pb, *d2 = (char *)dev2 + i * pb;
                                    CK(hipMemcpyAsync(d1, up + i * pb, pb, hipMemcpyHostToDevice, st));
                                    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((pb / 16 + 255) / 256)), block, 0, st, (const ulonglong2 *)d1, (ulonglong2 *)d2, pb / 16);
                                    CK(hipMemcpyAsync(up + i * pb, d2, pb, hipMemcpyDeviceToHost, st));
                                }
                                CK(hipStreamSynchronize(s));
                                CK(hipStreamSynchronize(s2));
                            }));
            }
            CK(hipHostUnregister(up));
        }
        // duplex: thread A copies user -> dev on s, thread B copies dev2 -> user2 on s2 (pageable both)
        std::printf("pageable H2D || pageable D2H, 2 thr: %8.1f us\n", timeit(reps, [&] {
                        std::thread tb([&] {
                            CK(hipMemcpyAsync(user2.data(), dev2, bytes, hipMemcpyDeviceToHost, s2));
                            CK(hipStreamSynchronize(s2));
                        });
                        CK(hipMemcpyAsync(dev, user.data(), bytes, hipMemcpyHostToDevice, s));
                        CK(hipStreamSynchronize(s));
                        tb.join();
                    }));
        std::printf("pinned H2D || pinned D2H, 2 streams: %8.1f us\n", timeit(reps, [&] {
                        CK(hipMemcpyAsync(dev, pin_in, bytes, hipMemcpyHostToDevice, s));
                        CK(hipMemcpyAsync(pin_out, dev2, bytes, hipMemcpyDeviceToHost, s2));
                        CK(hipStreamSynchronize(s));
                        CK(hipStreamSynchronize(s2));
                    }));
        std::printf("kernel pinned->dev || kernel dev->pinned, 2 streams: %8.1f us\n", timeit(reps, [&] {
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s, (const ulonglong2 *)pin_in, (ulonglong2 *)dev, nvec);
                        hipLaunchKernelGGL(copy_kernel, grid, block, 0, s2, (const ulonglong2 *)dev2, (ulonglong2 *)pin_out, nvec);
                        CK(hipStreamSynchronize(s));
                        CK(hipStreamSynchronize(s2));
                    }));
        CK(hipHostFree(pin_in));
        CK(hipHostFree(pin_out));
        CK(hipFree(dev));
        CK(hipFree(dev2));
    }
    return 0;
}
