// Sustained shader clock under the transform kernels: s_memtime (shader-clock cycles) against s_memrealtime (the
// constant 100 MHz reference), read by a one-thread kernel before and after a run of launches of the library's own
// kernels (through the C ABI).  The static cycle model (tools/cycle_model.py) needs this clock: the butterfly kernels
// are VALU-issue bound, so their duration is cycles / clock, and the clock the chip sustains under them is not the
// 2.4 GHz peak.
//   hipcc -O2 --offload-arch=gfx950 -Iinclude -o tools/microbench8 tools/microbench8_clock.hip \
//         -Lprimus-fhe_amd -lpfhe_hip -Wl,-rpath,'$ORIGIN/../primus-fhe_amd'
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pfhe.h"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));       \
            std::exit(1);                                                      \
        }                                                                      \
    } while (0)
#define PK(x)                                                       \
    do {                                                            \
        int r_ = (x);                                               \
        if (r_ != 0) {                                              \
            std::fprintf(stderr, "%s: pfhe error %d\n", #x, r_);    \
            std::exit(1);                                           \
        }                                                           \
    } while (0)

__global__ void read_clocks(unsigned long long *out) {
    out[0] = __builtin_amdgcn_s_memtime();
    out[1] = __builtin_amdgcn_s_memrealtime();
}

struct Sample {
    unsigned long long core, ref;
};

int main() {
    const uint64_t q[3] = {2305843009211596801ull, 2305843009210023937ull, 2305843009208713217ull};
    const uint32_t log_n = 16;
    const size_t batch = 4096, L = 3, n = 1u << log_n, words = batch * L * n;
    pfhe_dcrt *t = nullptr;
    PK(pfhe_dcrt_create(log_n, q, L, 0, &t));
    uint64_t *x = nullptr;
    unsigned long long *clk = nullptr;
    CK(hipMalloc(&x, words * sizeof(uint64_t)));
    CK(hipMalloc(&clk, 64 * sizeof(unsigned long long)));
    PK(pfhe_fill_uniform_dev(0, x, words, q, L, n, 1, nullptr));
    CK(hipDeviceSynchronize());
    const auto sample = [&](int slot) { hipLaunchKernelGGL(read_clocks, dim3(1), dim3(1), 0, 0, clk + 2 * slot); };
    const auto report = [&](const char *what, int reps, int s0, int s1, float ms) {
        unsigned long long h[64];
        CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
        const double core = (double)(h[2 * s1] - h[2 * s0]), ref = (double)(h[2 * s1 + 1] - h[2 * s0 + 1]);
        std::printf("%-44s %3d launches  %8.3f ms each   shader clock %7.1f MHz  (ref clock span %.3f ms)\n", what, reps,
                    ms / reps, core / ref * 100.0, ref / 100e3);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Case {
        const char *name;
        int inverse, pass;  // pass -1: the whole transform (default form)
    };
    const Case cases[] = {{"block pass, forward (stand-alone)", 0, 1},   {"strided pass, forward (stand-alone)", 0, 0},
                          {"whole forward transform (default form)", 0, -1}, {"block pass, inverse (stand-alone)", 1, 0},
                          {"whole inverse transform (default form)", 1, -1}};
    for (int round = 0; round < 2; ++round) {  // the first round warms the clocks up
        for (const Case &c : cases) {
            const int reps = 20;
            sample(0);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) {
                if (c.pass >= 0) PK(pfhe_dcrt_transform_pass_dev(t, x, words, c.inverse, c.pass, 0, nullptr));
                else if (c.inverse) PK(pfhe_dcrt_inverse_transform_dev(t, x, words, 0, nullptr));
                else PK(pfhe_dcrt_transform_dev(t, x, words, 0, nullptr));
            }
            CK(hipEventRecord(e1, 0));
            sample(1);
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (round == 1) report(c.name, reps, 0, 1, ms);
            PK(pfhe_fill_uniform_dev(0, x, words, q, L, n, 1, nullptr));
            CK(hipDeviceSynchronize());
        }
    }
    pfhe_dcrt_destroy(t);
    return 0;
}
