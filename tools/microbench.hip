// microbench.hip — MI355X calibration for the NTT design (SURVEY.md §7 "measure this first"):
//   (1) integer multiply issue rates that bound a 64-bit Shoup butterfly,
//   (2) register-resident Harvey butterfly rate (the VALU ceiling of the NTT),
//   (3) in-place streaming bandwidth vs working-set size (HBM vs Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using u64 = unsigned long long;
using u32 = unsigned int;

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e = (x);                                                               \
        if (e != hipSuccess) {                                                            \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int OP>
__global__ __launch_bounds__(256) void alu_kernel(u64 *out, u32 seed) {
    u32 a[ILP], b = seed | 1u;
    u64 c[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
        c[i] = ((u64)a[i] << 32) | (a[i] * 7u);
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) a[i] = a[i] * b + 1u;                          // v_mul_lo_u32 (+add)
            if (OP == 1) a[i] = __umulhi(a[i], b) + a[i];                // v_mul_hi_u32 (+add)
            if (OP == 2) c[i] = (u64)(u32)c[i] * (u64)b + c[i];          // v_mad_u64_u32
            if (OP == 3) c[i] = c[i] * (c[i] | 1ull) + 3ull;             // 64-bit mul lo
            if (OP == 4) c[i] = __umul64hi(c[i], c[i] | 5ull) + c[i];    // 64-bit mul hi
            if (OP == 5) c[i] = c[i] + (c[i] >> 7);                      // 64-bit add + shift (baseline)
            if (OP == 6) a[i] = a[i] + (a[i] >> 3);                      // 32-bit add + shift (baseline)
            if (OP == 7) a[i] = __mul24(a[i], b) + a[i];                 // v_mul_u32_u24 / mad24
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += c[i] + a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ u64 red1(u64 x, u64 m) { u64 y = x - m; return x < y ? x : y; }

// 8 independent Harvey butterflies per iteration on registers, twiddle fixed per lane
__global__ __launch_bounds__(256) void bfly_kernel(u64 *out, u64 q, u64 w, u64 wp) {
    u64 x[16];
    const u64 two_q = 2 * q;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull) % q;
    w += threadIdx.x;
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & (1 << s)) continue;
                u64 &a = x[k], &b = x[k | (1 << s)];
                u64 tx = red1(a, two_q);
                u64 t = w * b - q * __umul64hi(wp, b);
                a = tx + t;
                b = tx + two_q - t;
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void stream_rmw(ulonglong2 *p, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        ulonglong2 v = p[i];
        v.x += 1;
        v.y ^= v.x;
        p[i] = v;
    }
}

__global__ __launch_bounds__(256) void stream_copy(const ulonglong2 *__restrict__ s, ulonglong2 *__restrict__ d, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

template <class F>
static float time_ms(F &&f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s, CUs %d, clock %d kHz, L2 %d KiB\n", prop.name, prop.multiProcessorCount, prop.clockRate,
                prop.l2CacheSize / 1024);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    u64 *out;
    CK(hipMalloc(&out, (size_t)blocks * threads * sizeof(u64)));
    const char *names[] = {"u32 mul_lo+add", "u32 mul_hi+add", "mad_u64_u32", "u64 mul lo (+add)", "u64 mul hi (+add)",
                           "u64 add+shift", "u32 add+shift", "u32 mul24+add"};
    const double lanes = (double)blocks * threads;
#define RUN(OP)                                                                                            \
    {                                                                                                      \
        float ms = time_ms([&] { hipLaunchKernelGGL(alu_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }, 5); \
        double ops = lanes * ITERS * ILP;                                                                  \
        std::printf("ALU %-20s %8.3f ms  %8.2f Gop/s (lane-ops)\n", names[OP], ms, ops / ms * 1e-6);       \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    {
        const u64 q = 2305843009211596801ull, w = 25740574174379ull;
        const u64 wp = (u64)(((unsigned __int128)w << 64) / q);
        float ms = time_ms([&] { hipLaunchKernelGGL(bfly_kernel, dim3(blocks), dim3(threads), 0, 0, out, q, w, wp); }, 5);
        double bf = lanes * (ITERS / 4) * 4 * 8;
        std::printf("Harvey butterflies in registers: %8.3f ms  %8.2f Gbfly/s  => N=2^16 NTT ceiling %.2f M/s\n", ms,
                    bf / ms * 1e-6, bf / ms * 1e-3 / 524288.0);
    }
    // streaming bandwidth vs working set
    const size_t sizes_mb[] = {32, 64, 128, 192, 256, 512, 1024, 4096};
    for (size_t mb : sizes_mb) {
        size_t bytes = mb << 20;
        ulonglong2 *buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) break;
        CK(hipMemset(buf, 1, bytes));
        size_t nvec = bytes / 16;
        int reps = mb <= 256 ? 40 : 10;
        float ms = time_ms([&] { hipLaunchKernelGGL(stream_rmw, dim3(blocks), dim3(threads), 0, 0, buf, nvec); }, reps);
        std::printf("in-place RMW %5zu MiB: %8.3f ms  %8.1f GB/s (read+write)\n", mb, ms, 2.0 * bytes / ms * 1e-6);
        CK(hipFree(buf));
    }
    for (size_t mb : {256ul, 2048ul}) {
        size_t bytes = mb << 20;
        ulonglong2 *s, *d;
        CK(hipMalloc(&s, bytes));
        CK(hipMalloc(&d, bytes));
        CK(hipMemset(s, 1, bytes));
        float ms = time_ms([&] { hipLaunchKernelGGL(stream_copy, dim3(blocks), dim3(threads), 0, 0, s, d, bytes / 16); }, 10);
        std::printf("copy %5zu MiB -> other buffer: %8.3f ms  %8.1f GB/s (read+write)\n", mb, ms, 2.0 * bytes / ms * 1e-6);
        CK(hipFree(s));
        CK(hipFree(d));
    }
    CK(hipFree(out));
    return 0;
}
