// microbench9_energy.hip — what one VALU instruction costs in ENERGY on gfx950: every resident wave (4 per SIMD) issues one
// instruction class from registers for a few seconds while tools/energy_probe.py samples the package power and the shader
// clock; rate x (power above the idle-clocked package) gives joules per lane-instruction.  The transforms sit at the 1400 W
// package cap, so their time is their energy / power: this is the price list behind that.
// Build: hipcc --offload-arch=gfx950 -O2 -o microbench9 microbench9_energy.hip ; run: ./microbench9 OP SECONDS
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using u64 = unsigned long long;
using u32 = unsigned int;
#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                           \
        }                                                                           \
    } while (0)

constexpr int ITERS = 4096, CH = 8, REP = 4;  // instructions per thread = ITERS * CH * REP

#define A8(fmt)                                                                                                            \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                                                   \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),         \
                   "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])          \
                 : "v"(b), "s"(sc), "v"(e)                                                                                 \
                 : "vcc")
// a[i] = %i (32-bit), d[i] = %(8+i) (64-bit), b = %16, sc = %17 (sgpr), e = %18 (64-bit)
#define F_ADD(i) "v_add_u32 %" #i ", %" #i ", %16\n\t"
#define F_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n\t"
#define F_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n\t"
#define F_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %16\n\t"
#define F_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n\t"
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n\t"
#define F_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 29\n\t"
#define F_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n\t"
#define F_MAD64_8(i) "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n\t"
#define F_LSHLADD64_8(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %18\n\t"
#define F_FMA64_8(i) "v_fma_f64 %" #i ", %" #i ", %18, %18\n\t"
#define D8(fmt) A8(fmt)

template <int OP>
__global__ __launch_bounds__(256) void burn(u32 *out, u32 b, u32 sc, u64 e) {
    u32 a[8];
    u64 d[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + b;
        d[i] = ((u64)a[i] << 20) ^ e;
    }
    for (int it = 0; it < ITERS; ++it) {
        for (int r = 0; r < REP; ++r) {
            if constexpr (OP == 0) A8(F_ADD);
            if constexpr (OP == 1) A8(F_AND);
            if constexpr (OP == 2) A8(F_MULLO);
            if constexpr (OP == 3) A8(F_MULHI);
            if constexpr (OP == 4) A8(F_MUL24);
            if constexpr (OP == 5) A8(F_MAD24);
            if constexpr (OP == 6) A8(F_ALIGN);
            if constexpr (OP == 7) A8(F_ADDCO);
            if constexpr (OP == 8) {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                             "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                             "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\t"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                             : "v"(a[0]), "v"(a[1])
                             : "vcc");
            }
            if constexpr (OP == 9) {
                asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\t"
                             "v_lshl_add_u64 %2, %2, 0, %8\n\tv_lshl_add_u64 %3, %3, 0, %8\n\t"
                             "v_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\t"
                             "v_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8\n\t"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                             : "v"(e));
            }
            if constexpr (OP == 10) {
                asm volatile("v_fma_f64 %0, %0, %8, %8\n\tv_fma_f64 %1, %1, %8, %8\n\t"
                             "v_fma_f64 %2, %2, %8, %8\n\tv_fma_f64 %3, %3, %8, %8\n\t"
                             "v_fma_f64 %4, %4, %8, %8\n\tv_fma_f64 %5, %5, %8, %8\n\t"
                             "v_fma_f64 %6, %6, %8, %8\n\tv_fma_f64 %7, %7, %8, %8\n\t"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                             : "v"(e));
            }
            if constexpr (OP == 11) {  // mad64 with small (24-bit) operands: does the multiplier's energy follow the data?
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                             "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                             "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\t"
                             : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                             : "v"(a[0] & 7u), "v"(a[1] & 0xffffffu)
                             : "vcc");
            }
        }
    }
    u32 acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= a[i] ^ (u32)d[i] ^ (u32)(d[i] >> 32);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;  // keeps the work alive
}

static const char *kNames[] = {"v_add_u32", "v_and_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24",
                               "v_alignbit_b32", "v_add_co_u32", "v_mad_u64_u32", "v_lshl_add_u64", "v_fma_f64",
                               "v_mad_u64_u32 (3-bit x 24-bit operands)"};

template <int OP>
static void run(double secs, u32 *out, int grid) {
    hipLaunchKernelGGL(burn<OP>, dim3(grid), dim3(256), 0, nullptr, out, 12345u, 77u, 0x9E3779B97F4A7C15ull);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    std::printf("PHASE %s\n", kNames[OP]);
    std::fflush(stdout);
    do {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(burn<OP>, dim3(grid), dim3(256), 0, nullptr, out, 12345u, 77u, 0x9E3779B97F4A7C15ull);
        CK(hipDeviceSynchronize());
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    const double lane_instr = (double)launches * grid * 256.0 * ITERS * CH * REP;
    std::printf("PHASE_END %s lane_instr_per_s %.4e seconds %.2f\n", kNames[OP], lane_instr / el, el);
    std::fflush(stdout);
}

int main(int argc, char **argv) {
    const int op = argc > 1 ? std::atoi(argv[1]) : 0;
    const double secs = argc > 2 ? std::atof(argv[2]) : 3.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    u32 *out = nullptr;
    CK(hipMalloc(&out, 4096));
    switch (op) {
        case 0: run<0>(secs, out, grid); break;
        case 1: run<1>(secs, out, grid); break;
        case 2: run<2>(secs, out, grid); break;
        case 3: run<3>(secs, out, grid); break;
        case 4: run<4>(secs, out, grid); break;
        case 5: run<5>(secs, out, grid); break;
        case 6: run<6>(secs, out, grid); break;
        case 7: run<7>(secs, out, grid); break;
        case 8: run<8>(secs, out, grid); break;
        case 9: run<9>(secs, out, grid); break;
        case 10: run<10>(secs, out, grid); break;
        case 11: run<11>(secs, out, grid); break;
        default: return 2;
    }
    return 0;
}
