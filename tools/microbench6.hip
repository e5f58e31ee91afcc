// microbench6.hip — register-resident pseudo-Mersenne butterflies on gfx950: the generated asm sequences of
// csrc/pfhe_pm_asm.hpp (one butterfly or two interleaved per block; twiddles in VGPRs or SGPRs) against the same
// arithmetic left to the compiler and against round 1's full-product multiply.  Every variant is checked against
// host arithmetic (values compared mod q), then timed at 4 waves per SIMD (the block pass's occupancy).
// Build: hipcc --offload-arch=gfx950 -O3 -I../primus-fhe_amd/csrc -o microbench6 microbench6.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using u64 = unsigned long long;
using u32 = unsigned int;
namespace pfhe {
using ::u32;
using ::u64;
}
#include "pfhe_pm_asm.hpp"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                           \
        }                                                                           \
    } while (0)

struct Tw {
    u64 w, w2;
};
struct Pm {
    u64 q, q3;
    u32 c, c2, sh, mask;
    u32 vsh, vmask, vmask1;  // copies the kernels keep in VGPRs
    // compiler-scheduled forms
    __device__ __forceinline__ u64 fold(u64 x) const {
        const u32 x1 = (u32)(x >> 32);
        const u64 low = ((u64)(x1 & mask) << 32) | (u32)x;
        return (u64)(x1 >> sh) * c + low;
    }
    __device__ __forceinline__ u64 mul_split(u64 y, Tw t) const {  // same maths as the asm, compiler's schedule
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)t.w, w1 = (u32)(t.w >> 32), v0 = (u32)t.w2, v1 = (u32)(t.w2 >> 32);
        const u64 t0 = (u64)y0 * w0;
        u64 t1;
        const bool cy = __builtin_uaddll_overflow((u64)y1 * v0, t0, &t1);
        u64 t2 = (u64)y0 * w1 + (t1 >> 32);
        asm("" : "+v"(t2));
        u64 t3 = (u64)y1 * v1 + t2;
        t3 += (u64)cy << 32;
        const u32 hp = __builtin_amdgcn_alignbit((u32)(t3 >> 32), (u32)t3, sh + 1);
        const u64 lo = ((u64)((u32)t3 & (2 * mask + 1)) << 32) | (u32)t1;
        return (u64)hp * c2 + lo;
    }
    __device__ __forceinline__ u64 mul_full(u64 y, u64 w) const {  // round 1: 128-bit product folded twice
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
        const u64 lo = (u64)w0 * y0;
        u64 mid = (u64)w0 * y1 + (lo >> 32);
        asm("" : "+v"(mid));
        mid += (u64)w1 * y0;
        const u64 hi = (u64)w1 * y1 + (mid >> 32);
        const u32 l0 = (u32)lo, l1 = (u32)mid, h0 = (u32)hi, h1 = (u32)(hi >> 32);
        const u32 f0 = __builtin_amdgcn_alignbit(h0, l1, sh), f1 = __builtin_amdgcn_alignbit(h1, h0, sh);
        const u64 plo = ((u64)(l1 & mask) << 32) | l0;
        const u64 a = (u64)f0 * c + plo;
        const u64 b = (u64)f1 * c + (a >> 32);
        const u32 rh = __builtin_amdgcn_alignbit((u32)(b >> 32), (u32)b, sh);
        const u64 rl = ((u64)((u32)b & mask) << 32) | (u32)a;
        return (u64)rh * c + rl;
    }
};

// VARIANT 0: round-1 butterfly (fold every stage, full product, x + 2q - t)
//         1: split-operand product, compiler-scheduled, fold at odd stages
//         2: asm, one butterfly per block        3: asm, two interleaved
// UNI: one twiddle per stage for the whole wave (SGPRs) instead of one per lane
template <int VARIANT, bool UNI, bool INV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void bf_kernel(u64 *data, const Tw *tw, Pm a,
                                                                                          int iters) {
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = data[(size_t)(blockIdx.x * 256 + threadIdx.x) * 16 + i];
    a.vsh = a.sh;
    a.vmask = a.mask;
    a.vmask1 = 2 * a.mask + 1;
    asm volatile("" : "+v"(a.vsh), "+v"(a.vmask), "+v"(a.vmask1));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 3; s >= 0; --s) {
            Tw w;
            if constexpr (UNI) {
                w = tw[__builtin_amdgcn_readfirstlane((it * 4 + s) & 63)];
            } else {
                w = tw[(threadIdx.x + it * 4 + s) & 255];
            }
            const bool fold = (s & 1) != 0;  // compile-time after unrolling
#pragma unroll
            for (int k = 0; k < 16; k += 1) {
                if (k & (1 << s)) continue;
                const int k1 = k | (1 << s);
                if constexpr (VARIANT == 0) {
                    if constexpr (!INV) {
                        const u64 tx = a.fold(x[k]), t = a.mul_full(x[k1], w.w);
                        x[k] = tx + t;
                        x[k1] = tx + 2 * a.q - t;
                    } else {
                        const u64 tx = x[k] + x[k1], ty = x[k] + 2 * a.q - x[k1];
                        x[k] = a.fold(tx);
                        x[k1] = a.mul_full(ty, w.w);
                    }
                } else if constexpr (VARIANT == 1) {
                    if constexpr (!INV) {
                        const u64 tx = fold ? a.fold(x[k]) : x[k], t = a.mul_split(x[k1], w);
                        x[k] = tx + t;
                        x[k1] = tx + a.q3 - t;
                    } else {
                        const u64 tx = x[k] + x[k1], ty = x[k] + a.q3 - x[k1];
                        x[k] = a.fold(tx);
                        x[k1] = a.mul_split(ty, w);
                    }
                } else if constexpr (VARIANT == 2) {
                    if constexpr (!INV) {
                        if (fold) pfhe::pm_fwd_bfly1<true, UNI>(a, x[k], x[k1], w);
                        else pfhe::pm_fwd_bfly1<false, UNI>(a, x[k], x[k1], w);
                    } else {
                        pfhe::pm_inv_bfly1<UNI>(a, x[k], x[k1], w);
                    }
                } else {
                    // pair butterfly k with the next butterfly of the stage
                    int kb = k + 1;
                    while (kb & (1 << s)) ++kb;
                    // only the first of each pair launches the block
                    int idx = 0;
                    for (int t = 0; t < k; ++t)
                        if (!(t & (1 << s))) ++idx;
                    if (idx & 1) continue;
                    const int kb1 = kb | (1 << s);
                    if constexpr (!INV) {
                        if (fold) pfhe::pm_fwd_bfly2<true, UNI>(a, x[k], x[k1], w, x[kb], x[kb1], w);
                        else pfhe::pm_fwd_bfly2<false, UNI>(a, x[k], x[k1], w, x[kb], x[kb1], w);
                    } else {
                        pfhe::pm_inv_bfly2<UNI>(a, x[k], x[k1], w, x[kb], x[kb1], w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) data[(size_t)(blockIdx.x * 256 + threadIdx.x) * 16 + i] = x[i];
}

using u128 = unsigned __int128;
static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

// host model of one round (4 stages) on canonical values
template <bool INV>
static void host_round(u64 *x, const Tw *w4, u64 q) {
    for (int s = 3; s >= 0; --s)
        for (int k = 0; k < 16; ++k) {
            if (k & (1 << s)) continue;
            const int k1 = k | (1 << s);
            if (!INV) {
                const u64 t = mulmod(x[k1], w4[s].w, q), a = x[k];
                x[k] = (a + t) % q;
                x[k1] = (a + q - t) % q;
            } else {
                const u64 a = x[k], b = x[k1];
                x[k] = (a + b) % q;
                x[k1] = mulmod((a + q - b) % q, w4[s].w, q);
            }
        }
}

template <int VARIANT, bool UNI, bool INV>
static void run(const char *name, const Pm &pm, const std::vector<Tw> &tw, Tw *dtw, int cus) {
    const int blocks = cus * 4, iters_check = 3, iters_time = 2000;
    const size_t n = (size_t)blocks * 256 * 16;
    std::vector<u64> h(n);
    u64 s = 0x243F6A8885A308D3ull;
    for (auto &v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = s % pm.q;
    }
    u64 *d;
    CK(hipMalloc(&d, n * 8));
    CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((bf_kernel<VARIANT, UNI, INV>), dim3(blocks), dim3(256), 0, 0, d, dtw, pm, iters_check);
    CK(hipDeviceSynchronize());
    std::vector<u64> r(n);
    CK(hipMemcpy(r.data(), d, n * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    u64 maxv = 0;
    for (int b : {0, 1, blocks - 1})
        for (int t = 0; t < 256; ++t) {
            u64 x[16];
            const size_t base = ((size_t)b * 256 + t) * 16;
            for (int i = 0; i < 16; ++i) x[i] = h[base + i];
            for (int it = 0; it < iters_check; ++it) {
                Tw w4[4];
                for (int st = 0; st < 4; ++st) w4[st] = UNI ? tw[(it * 4 + st) & 63] : tw[(t + it * 4 + st) & 255];
                host_round<INV>(x, w4, pm.q);
            }
            for (int i = 0; i < 16; ++i) {
                if (r[base + i] % pm.q != x[i]) ++bad;
                if (r[base + i] > maxv) maxv = r[base + i];
            }
        }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bf_kernel<VARIANT, UNI, INV>), dim3(blocks), dim3(256), 0, 0, d, dtw, pm, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((bf_kernel<VARIANT, UNI, INV>), dim3(blocks), dim3(256), 0, 0, d, dtw, pm, iters_time);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bf = (double)blocks * 256 * 32.0 * iters_time;
    std::printf("%-44s %s  mismatches %ld (max value %.3f * 2^61)  %8.3f ms  %7.1f Gbfly/s  => 2^16-point NTT ceiling %.2f M/s\n", name,
                INV ? "inv" : "fwd", bad, (double)maxv / 2305843009213693952.0, ms, bf / ms * 1e-6, bf / ms * 1e-3 / 524288.0);
    CK(hipFree(d));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (u64 q : {2305843009211596801ull, 2305843009208713217ull, 1125899906826241ull}) {
        Pm pm{};
        const int K = 64 - __builtin_clzll(q);
        pm.q = q;
        pm.q3 = 3 * q;
        pm.c = (u32)((1ull << K) - q);
        pm.c2 = 2 * pm.c;
        pm.sh = K - 32;
        pm.mask = (1u << (K - 32)) - 1;
        std::vector<Tw> tw(256);
        u64 s = 88172645463325252ull + q;
        for (auto &t : tw) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            t.w = s % q;
            t.w2 = (u64)(((u128)t.w << 32) % q);
        }
        tw[3].w = q - 1; tw[3].w2 = (u64)(((u128)(q - 1) << 32) % q);
        Tw *dtw;
        CK(hipMalloc(&dtw, tw.size() * sizeof(Tw)));
        CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(Tw), hipMemcpyHostToDevice));
        std::printf("q = %llu (K = %d, c = %u)\n", q, K, pm.c);
        run<0, false, false>("round-1 full product (per-lane twiddles)", pm, tw, dtw, cus);
        run<1, false, false>("split product, compiler (per-lane)", pm, tw, dtw, cus);
        run<2, false, false>("split product, asm x1 (per-lane)", pm, tw, dtw, cus);
        run<3, false, false>("split product, asm x2 (per-lane)", pm, tw, dtw, cus);
        run<0, true, false>("round-1 full product (uniform twiddles)", pm, tw, dtw, cus);
        run<2, true, false>("split product, asm x1 (uniform)", pm, tw, dtw, cus);
        run<3, true, false>("split product, asm x2 (uniform)", pm, tw, dtw, cus);
        run<0, false, true>("round-1 full product (per-lane)", pm, tw, dtw, cus);
        run<1, false, true>("split product, compiler (per-lane)", pm, tw, dtw, cus);
        run<2, false, true>("split product, asm x1 (per-lane)", pm, tw, dtw, cus);
        run<3, false, true>("split product, asm x2 (per-lane)", pm, tw, dtw, cus);
        run<3, true, true>("split product, asm x2 (uniform)", pm, tw, dtw, cus);
        CK(hipFree(dtw));
    }
    return 0;
}
