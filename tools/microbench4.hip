// microbench4.hip — 30-bit modular butterfly on gfx950: integer Barrett-32 (the reference's
// arithmetic) against an exact FP64 formulation (error-free product via FMA, no range reductions),
// register-resident, 4 resident waves per SIMD like the block pass.  Also raw issue rates of the
// FP64 / integer-multiply instructions involved.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
using u64 = unsigned long long; using u32 = unsigned int;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1);} } while (0)
#define DI __device__ __forceinline__

constexpr u32 Q = 1073479681u;

DI u32 once(u32 x, u32 m) { return min(x, x - m); }
DI u32 mul1(u32 y, u32 w, u32 wp) { return w * y - Q * __umulhi(y, wp); }

__global__ __launch_bounds__(256, 4) void bf_int(u32 *out, const u32 *tw, int iters) {
    u32 x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = out[threadIdx.x * 32 + i] % Q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 4; s >= 0; --s)
#pragma unroll
            for (int u = 0; u < (32 >> (s + 1)); ++u) {
                const u32 w = tw[2 * ((threadIdx.x & 63) * 32 + (16 >> s) + u + (it & 7))];
                const u32 wp = tw[2 * ((threadIdx.x & 63) * 32 + (16 >> s) + u + (it & 7)) + 1];
#pragma unroll
                for (int v = 0; v < (1 << s); ++v) {
                    const int k0 = (u << (s + 1)) | v, k1 = k0 | (1 << s);
                    const u32 tx = once(x[k0], 2 * Q), t = mul1(x[k1], w, wp);
                    x[k0] = tx + t; x[k1] = tx + 2 * Q - t;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) out[threadIdx.x * 32 + i] = x[i];
}

// exact y*w mod q in (-q, q): p = rn(y*w), e = y*w - p (exact), c = rint(y * (w/q)), p - c*q exact
DI double fmulmod(double y, double w, double wq) {
    const double p = y * w;
    const double e = __builtin_fma(y, w, -p);
    const double c = __builtin_rint(y * wq);
    return __builtin_fma(-c, (double)Q, p) + e;
}

__global__ __launch_bounds__(256, 4) void bf_f64(double *out, const double *tw, int iters) {
    double x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = out[threadIdx.x * 32 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 4; s >= 0; --s)
#pragma unroll
            for (int u = 0; u < (32 >> (s + 1)); ++u) {
                const double w = tw[2 * ((threadIdx.x & 63) * 32 + (16 >> s) + u + (it & 7))];
                const double wq = tw[2 * ((threadIdx.x & 63) * 32 + (16 >> s) + u + (it & 7)) + 1];
#pragma unroll
                for (int v = 0; v < (1 << s); ++v) {
                    const int k0 = (u << (s + 1)) | v, k1 = k0 | (1 << s);
                    const double t = fmulmod(x[k1], w, wq);
                    const double a = x[k0];
                    x[k0] = a + t; x[k1] = a - t;
                }
            }
        // keep magnitudes bounded between "transforms" (5 stages grow by at most 5q): one cheap fold
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_fma(-__builtin_rint(x[i] * (1.0 / Q)), (double)Q, x[i]);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) out[threadIdx.x * 32 + i] = x[i];
}

// raw instruction issue rates: 16 independent chains per thread
template <int OP>
__global__ __launch_bounds__(256, 4) void rate(double *out, int iters) {
    double a[16]; u32 b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = out[threadIdx.x * 16 + i]; b[i] = (u32)threadIdx.x * 16 + i + 12345; }
    const double c1 = out[0], c2 = out[1];
    const u32 m1 = (u32)c1 | 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], c1, c2);
            if (OP == 1) a[i] = a[i] * c1;
            if (OP == 2) a[i] = a[i] + c2;
            if (OP == 3) a[i] = __builtin_rint(a[i]) + 0.0 * c1;
            if (OP == 4) b[i] = b[i] * m1;
            if (OP == 5) b[i] = __umulhi(b[i], m1) + it;
            if (OP == 6) b[i] = b[i] + m1;
            if (OP == 7) { a[i] = (double)b[i]; b[i] += 1; }
            if (OP == 8) { b[i] = (u32)a[i]; a[i] += 1.0; }
            if (OP == 9) b[i] = min(b[i], b[i] - m1);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = a[i] + b[i];
}

template <class F> static float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); f(2000); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int cus = p.multiProcessorCount;
    const int wgs = cus * 4;
    void *out, *tw; CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&tw, 4 << 20));
    // integer butterfly
    {
        std::vector<u32> h(1 << 18), t(1 << 18);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(i * 2654435761u) % Q;
        for (size_t i = 0; i < t.size(); i += 2) { u32 w = (u32)((i + 7) * 2246822519u) % Q; t[i] = w; t[i + 1] = (u32)(((u64)w << 32) / Q); }
        CK(hipMemcpy(out, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        float ms = timeit([&](int it) { hipLaunchKernelGGL(bf_int, dim3(wgs), dim3(256), 0, 0, (u32 *)out, (const u32 *)tw, it); });
        double bf = (double)wgs * 256 * 2000 * 80;
        std::printf("Barrett-32 integer butterfly: %8.1f Gbfly/s\n", bf / ms * 1e-6);
    }
    {
        std::vector<double> h(1 << 18), t(1 << 18);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((u32)(i * 2654435761u) % Q);
        for (size_t i = 0; i < t.size(); i += 2) { u32 w = (u32)((i + 7) * 2246822519u) % Q; t[i] = w; t[i + 1] = (double)w / (double)Q; }
        CK(hipMemcpy(out, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, t.data(), t.size() * 8, hipMemcpyHostToDevice));
        // exactness of fmulmod against integers on the host formula (same IEEE ops)
        long bad = 0;
        for (int i = 0; i < 200000; ++i) {
            u64 y = ((u64)i * 2654435761u + 12345) % (17ull * Q); u32 w = (u32)((i + 7) * 2246822519u) % Q;
            double r = std::fma(-std::rint((double)y * ((double)w / Q)), (double)Q, (double)y * w) + std::fma((double)y, (double)w, -((double)y * w));
            long long ri = (long long)r; if (std::fabs(r) >= Q || (double)ri != r || ((ri % (long long)Q) + Q) % Q != (long long)((unsigned __int128)y * w % Q)) ++bad;
        }
        std::printf("fmulmod exactness check (host IEEE, y < 17q): %ld mismatches\n", bad);
        float ms = timeit([&](int it) { hipLaunchKernelGGL(bf_f64, dim3(wgs), dim3(256), 0, 0, (double *)out, (const double *)tw, it); });
        double bf = (double)wgs * 256 * 2000 * 80;
        std::printf("FP64 exact butterfly (+fold) : %8.1f Gbfly/s\n", bf / ms * 1e-6);
    }
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64(+add)", "v_mul_lo_u32", "v_mul_hi_u32(+add)", "v_add_u32", "v_cvt_f64_u32(+add)", "v_cvt_u32_f64(+add)", "sub+min"};
    auto go = [&](int op, auto kern) {
        CK(hipMemset(out, 0, 4 << 20));
        float ms = timeit([&](int it) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, (double *)out, it); });
        double ops = (double)wgs * 256 * 2000 * 16;
        std::printf("%-22s %8.1f Gop/s (lane ops)  = %.2f cycles per wave-instruction per SIMD\n", names[op], ops / ms * 1e-6,
                    (double)cus * 4 * 64 * 2.4e9 / (ops / ms * 1e3));
    };
    go(0, rate<0>); go(1, rate<1>); go(2, rate<2>); go(3, rate<3>); go(4, rate<4>); go(5, rate<5>); go(6, rate<6>); go(7, rate<7>); go(8, rate<8>); go(9, rate<9>);
    return 0;
}
