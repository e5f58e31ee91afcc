// microbench3.hip — VALU throughput of the pseudo-Mersenne Harvey butterfly vs occupancy
// (waves per SIMD), register-resident, distinct twiddle per butterfly group like the real kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using u64 = unsigned long long; using u32 = unsigned int;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1);} } while (0)
#define DI __device__ __forceinline__
struct C { u64 q, two_q; u32 c, sh, mask; };
DI u64 pm_mul(const C &k, u64 y, u64 w) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u64 lo = (u64)w0 * y0; u64 mid = (u64)w0 * y1 + (lo >> 32); mid += (u64)w1 * y0;
    const u64 hi = (u64)w1 * y1 + (mid >> 32);
    const u32 l0 = (u32)lo, l1 = (u32)mid, h0 = (u32)hi, h1 = (u32)(hi >> 32);
    const u32 f0 = __builtin_amdgcn_alignbit(h0, l1, k.sh), f1 = __builtin_amdgcn_alignbit(h1, h0, k.sh);
    const u64 plo = ((u64)(l1 & k.mask) << 32) | l0;
    const u64 a = (u64)f0 * k.c + plo; const u64 b = (u64)f1 * k.c + (a >> 32);
    const u32 rh = __builtin_amdgcn_alignbit((u32)(b >> 32), (u32)b, k.sh);
    const u64 rl = ((u64)((u32)b & k.mask) << 32) | (u32)a;
    return (u64)rh * k.c + rl;
}
DI u64 pm_red(const C &k, u64 x) { const u32 x1 = (u32)(x >> 32); return (u64)(x1 >> k.sh) * k.c + (((u64)(x1 & k.mask) << 32) | (u32)x); }
template <int MAXW>
__global__ __launch_bounds__(256, MAXW) void bf(u64 *out, const u64 *tw, C k, int iters) {
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = out[threadIdx.x * 16 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 3; s >= 0; --s)
#pragma unroll
            for (int u = 0; u < (16 >> (s + 1)); ++u) {
                const u64 w = tw[(threadIdx.x & 63) * 16 + (8 >> s) + u + (it & 7)];
#pragma unroll
                for (int v = 0; v < (1 << s); ++v) {
                    const int k0 = (u << (s + 1)) | v, k1 = k0 | (1 << s);
                    const u64 tx = pm_red(k, x[k0]); const u64 t = pm_mul(k, x[k1], w);
                    x[k0] = tx + t; x[k1] = tx + k.two_q - t;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = x[i];
}
template <int MAXW> static void run(u64 *out, u64 *tw, C k, int cus) {
    for (int wgs : {1, 2, 3, 4, 6, 8}) {
        if (wgs > MAXW) continue;
        const int iters = 2000; hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(bf<MAXW>, dim3(cus * wgs), dim3(256), 0, 0, out, tw, k, 10); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); hipLaunchKernelGGL(bf<MAXW>, dim3(cus * wgs), dim3(256), 0, 0, out, tw, k, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
        double bflies = (double)cus * wgs * 256 * iters * 32;
        std::printf("maxwaves/SIMD %d, resident waves/SIMD %d: %8.1f Gbfly/s -> N=2^16 NTT ceiling %.2f M/s\n", MAXW, wgs,
                    bflies / ms * 1e-6, bflies / ms * 1e-3 / 524288.0);
    }
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); int cus = p.multiProcessorCount;
    u64 *out, *tw; CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&tw, 1 << 20)); CK(hipMemset(out, 1, 1 << 20)); CK(hipMemset(tw, 3, 1 << 20));
    C k{2305843009211596801ull, 2 * 2305843009211596801ull, 2097151u, 29u, (1u << 29) - 1};
    run<8>(out, tw, k, cus); run<4>(out, tw, k, cus); run<2>(out, tw, k, cus);
    return 0;
}
