// microbench5.hip — gfx950 VALU issue costs, instruction by instruction (inline asm, so the compiler cannot
// substitute anything), measured in shader cycles (s_memtime, longest loop of any workgroup) at 1, 4 and 8 resident waves per SIMD, plus the
// shader clock the chip actually holds during the loop (s_memtime / s_memrealtime).
//
// Why: the NTT block pass is bound by VALU issue (VALUBusy 98 %).  The butterfly's cost model needs the price of
// every candidate instruction (32x32+64 multiply-add, 64-bit add forms, shifts, selects, 24-bit multiplies, FP64).
//
// Also checks two suspected hazards on results (carry chains back to back, multiply-add carry-out into an add).
// Build: hipcc --offload-arch=gfx950 -O2 -o microbench5 microbench5.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using u64 = unsigned long long;
using u32 = unsigned int;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__);            \
            std::exit(1);                                                                      \
        }                                                                                      \
    } while (0)

constexpr int ITERS = 512;   // loop trips
constexpr int CHAINS = 8;    // independent register sets per thread
constexpr int REP = 4;       // instruction groups per trip
constexpr int PER_WAVE = ITERS * CHAINS * REP;

struct Stamp {
    u64 cyc, real;
};

// One instruction form per OP.  a*: 32-bit VGPRs, d*: 64-bit VGPR pairs, s: SGPR operand.
#define A8(fmt)                                                                                             \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                                    \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),      \
                   "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),      \
                   "+v"(d[6]), "+v"(d[7])                                                                   \
                 : "v"(b), "s"(sc), "v"(e), "s"(sc64)                                                       \
                 : "vcc", "s20", "s21")
// operand numbers: a[i] = %i, d[i] = %(8+i), b = %16, sc = %17, e (64-bit) = %18, sc64 = %19

#define F_ADD(i) "v_add_u32 %" #i ", %" #i ", %16\n\t"
#define F_MOV(i) "v_mov_b32 %" #i ", %16\n\t"
#define F_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n\t"
#define F_SHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n\t"
#define F_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 29\n\t"
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 20\n\t"
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %16\n\t"
#define F_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n\t"
#define F_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n\t"
#define F_BFI(i) "v_bfi_b32 %" #i ", %16, %" #i ", %17\n\t"
#define F_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n\t"
#define F_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n\t"
#define F_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %16, vcc\n\t"
#define F_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n\t"
#define F_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %16\n\t"
#define F_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n\t"
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n\t"
#define F_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %16\n\t"
#define F_MADU16(i) "v_mad_u32_u16 %" #i ", %" #i ", %16, %17\n\t"
#define F_MIN(i) "v_min_u32 %" #i ", %" #i ", %16\n\t"
#define F_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %17\n\t"
#define F_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
#define F_SUB(i) "v_sub_u32 %" #i ", %" #i ", %16\n\t"
#define F_OR(i) "v_or_b32 %" #i ", %" #i ", %16\n\t"
#define F_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n\t"
#define F_NOT(i) "v_not_b32 %" #i ", %" #i "\n\t"
#define F_SHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n\t"
#define F_ADDS(i) "v_add_u32 %" #i ", %17, %" #i "\n\t"
#define F_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %16, %19\n\t"
#define F_MAX(i) "v_max_u32 %" #i ", %" #i ", %16\n\t"
#define F_ADDE64(i) "v_add_u32_e64 %" #i ", %" #i ", %16\n\t"
#define F_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %16, %" #i "\n\t"
// 64-bit destination forms: d[i] = %(8+i)
#define D(i) "%" #i "+8"
#define F_MAD64(i) "v_mad_u64_u32 %[d" #i "], s[20:21], %" #i ", %16, %[d" #i "]\n\t"

template <int OP>
__global__ __launch_bounds__(256) void inst_kernel(u64 *out, Stamp *stamps, u32 seed, u32 sc_in) {
    u32 a[CHAINS];
    u64 d[CHAINS];
    const u32 b = threadIdx.x * 2654435761u + seed;
    const u64 e = ((u64)b << 32) | (b * 7u + 1u);
    const u32 sc = __builtin_amdgcn_readfirstlane(sc_in);
    const u64 sc64 = ((u64)sc << 32) | (sc ^ 0x5555u);
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
        a[i] = b * (2 * i + 3) + i;
        d[i] = ((u64)a[i] << 32) | (a[i] ^ 0x9e3779b9u);
    }
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if constexpr (OP == 0) A8(F_ADD);
            if constexpr (OP == 1) A8(F_MOV);
            if constexpr (OP == 2) A8(F_AND);
            if constexpr (OP == 3) A8(F_SHL);
            if constexpr (OP == 4) A8(F_ALIGN);
            if constexpr (OP == 5) A8(F_BFE);
            if constexpr (OP == 6) A8(F_LSHLADD);
            if constexpr (OP == 7) A8(F_ADD3);
            if constexpr (OP == 8) A8(F_ANDOR);
            if constexpr (OP == 9) A8(F_BFI);
            if constexpr (OP == 10) A8(F_CNDMASK);
            if constexpr (OP == 11) A8(F_ADDCO);
            if constexpr (OP == 12) A8(F_ADDC);
            if constexpr (OP == 13) A8(F_MULLO);
            if constexpr (OP == 14) A8(F_MULHI);
            if constexpr (OP == 15) A8(F_MUL24);
            if constexpr (OP == 16) A8(F_MAD24);
            if constexpr (OP == 17) A8(F_MULHI24);
            if constexpr (OP == 18) A8(F_MADU16);
            if constexpr (OP == 19) A8(F_MIN);
            if constexpr (OP == 20) A8(F_PERM);
            if constexpr (OP == 21) A8(F_FMA32);
            if constexpr (OP == 22) A8(F_DOT4);
            if constexpr (OP == 23) A8(F_SUB);
            if constexpr (OP == 24) A8(F_OR);
            if constexpr (OP == 25) A8(F_XOR);
            if constexpr (OP == 26) A8(F_NOT);
            if constexpr (OP == 27) A8(F_SHR);
            if constexpr (OP == 28) A8(F_ADDS);
            if constexpr (OP == 29) A8(F_CNDS);
            if constexpr (OP == 41) A8(F_MAX);
            if constexpr (OP == 42) A8(F_ADDE64);
            if constexpr (OP == 30) {  // v_mad_u64_u32, all-VGPR operands
#define G(i) "v_mad_u64_u32 %" #i ", s[20:21], %16, %16, %" #i "\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 31) {  // v_mad_u64_u32 with an SGPR multiplicand
#define G(i) "v_mad_u64_u32 %" #i ", s[20:21], %16, %17, %" #i "\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 32) {  // v_mad_u64_u32 with a zero addend (inline constant)
#define G(i) "v_mad_u64_u32 %" #i ", s[20:21], %16, %17, 0\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 33) {  // v_lshl_add_u64 (64-bit add in one instruction)
#define G(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %18\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 34) {  // v_lshl_add_u64 with an SGPR-pair addend
#define G(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %19\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 35) {  // v_lshlrev_b64
#define G(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 36) {  // v_lshrrev_b64
#define G(i) "v_lshrrev_b64 %" #i ", 3, %" #i "\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 37) {  // v_fma_f64
#define G(i) "v_fma_f64 %" #i ", %" #i ", %18, %18\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 38) {  // v_pk_fma_f32 (two f32 lanes per 64-bit pair)
#define G(i) "v_pk_fma_f32 %" #i ", %" #i ", %18, %18\n\t"
                asm volatile(G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 39) {  // mixed: one v_mad_u64_u32 + two v_add_u32 (does the cheap work hide?)
#define G(i, j) "v_mad_u64_u32 %" #i ", s[20:21], %16, %17, %" #i "\n\tv_add_u32 %" #j ", %" #j ", %16\n\tv_and_b32 %" #j ", %" #j ", %17\n\t"
                asm volatile(G(8, 0) G(9, 1) G(10, 2) G(11, 3) G(12, 4) G(13, 5) G(14, 6) G(15, 7)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
            if constexpr (OP == 40) {  // v_add_co_u32 + v_addc_co_u32 back to back (a 64-bit add as a carry chain)
#define G(i, j) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n\tv_addc_co_u32 %" #j ", vcc, %" #j ", %16, vcc\n\t"
                asm volatile(G(0, 1) G(2, 3) G(4, 5) G(6, 7) G(0, 1) G(2, 3) G(4, 5) G(6, 7)
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                               "+v"(a[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                               "+v"(d[6]), "+v"(d[7])
                             : "v"(b), "s"(sc), "v"(e), "s"(sc64)
                             : "vcc", "s20", "s21");
#undef G
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i] + d[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
}

// ---- hazard checks on results ------------------------------------------------------------------
// (1) 64-bit add as v_add_co_u32 / v_addc_co_u32 back to back; (2) carry-out of v_mad_u64_u32 consumed by the very
// next instruction.  LLVM inserts two wait states between a VALU write of an SGPR/VCC and a VALU read of it on
// gfx940+; inline asm is not seen by that pass, so find out whether the hardware interlocks.
__global__ void hazard_kernel(const u64 *in, u64 *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 x = in[2 * i], y = in[2 * i + 1];
    u32 lo, hi;
    asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"((u32)x), "v"((u32)(x >> 32)), "v"((u32)y), "v"((u32)(y >> 32))
                 : "vcc");
    out[3 * i] = ((u64)hi << 32) | lo;
    // x0*y0 + y (carry out) -> immediately added to x1
    u64 p;
    u32 c;
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\tv_addc_co_u32 %1, vcc, 0, %5, vcc"
                 : "=&v"(p), "=&v"(c)
                 : "v"((u32)x), "v"((u32)(x >> 32)), "v"(y), "v"((u32)(y >> 32))
                 : "vcc");
    out[3 * i + 1] = p;
    out[3 * i + 2] = c;
}

struct Row {
    const char *name;
    int op;
};

template <int OP>
static void run_one(const char *name, int waves_per_simd, u64 *out, Stamp *stamps, int cus) {
    // one workgroup of 256 threads = one wave per SIMD; W workgroups per CU = W waves per SIMD
    const int blocks = cus * waves_per_simd;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(inst_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, stamps, 1u, 77u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(inst_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, stamps, 1u, 77u);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<Stamp> h(blocks);
    CK(hipMemcpy(h.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
    // the LONGEST loop is what counts: VALU issue is arbitrated by age, so the oldest wave of a SIMD runs almost as
    // if alone and finishes early; the youngest one sees the SIMD's true throughput
    double cyc = 0, real = 0;
    for (auto &s : h) {
        if ((double)s.cyc > cyc) {
            cyc = (double)s.cyc;
            real = (double)s.real;
        }
    }
    // instructions issued per SIMD during the loop = waves_per_simd * PER_WAVE (mixed forms count their own)
    const double per_inst = cyc / ((double)waves_per_simd * PER_WAVE);
    const double mhz = real > 0 ? cyc / real * 100.0 : 0.0;  // s_memrealtime ticks at 100 MHz
    std::printf("%-34s W=%d  %7.3f cyc/inst/SIMD   loop %9.0f cyc  clock %6.0f MHz  wall %.3f ms\n", name, waves_per_simd,
                per_inst, cyc, mhz, ms);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

#define RUN(OP, NAME)                                                    \
    for (int w : {1, 4, 8}) run_one<OP>(NAME, w, out, stamps, cus);

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs; %d instructions per wave per run (mixed rows: per listed group)\n", prop.gcnArchName, cus,
                PER_WAVE);
    u64 *out;
    Stamp *stamps;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(u64)));
    CK(hipMalloc(&stamps, (size_t)cus * 8 * sizeof(Stamp)));
    RUN(0, "v_add_u32")
    RUN(1, "v_mov_b32")
    RUN(2, "v_and_b32")
    RUN(3, "v_lshlrev_b32")
    RUN(4, "v_alignbit_b32")
    RUN(5, "v_bfe_u32")
    RUN(6, "v_lshl_add_u32")
    RUN(7, "v_add3_u32 (sgpr src)")
    RUN(8, "v_and_or_b32 (sgpr src)")
    RUN(9, "v_bfi_b32 (sgpr src)")
    RUN(10, "v_cndmask_b32 (vcc)")
    RUN(11, "v_add_co_u32")
    RUN(12, "v_addc_co_u32")
    RUN(13, "v_mul_lo_u32")
    RUN(14, "v_mul_hi_u32")
    RUN(15, "v_mul_u32_u24")
    RUN(16, "v_mad_u32_u24 (sgpr src)")
    RUN(17, "v_mul_hi_u32_u24")
    RUN(18, "v_mad_u32_u16 (sgpr src)")
    RUN(19, "v_min_u32")
    RUN(20, "v_perm_b32 (sgpr src)")
    RUN(21, "v_fma_f32 (sgpr src)")
    RUN(22, "v_dot4_u32_u8")
    RUN(23, "v_sub_u32")
    RUN(24, "v_or_b32")
    RUN(25, "v_xor_b32")
    RUN(26, "v_not_b32")
    RUN(27, "v_lshrrev_b32")
    RUN(28, "v_add_u32 (sgpr src)")
    RUN(29, "v_cndmask_b32 (sgpr-pair select)")
    RUN(41, "v_max_u32")
    RUN(42, "v_add_u32_e64 (VOP3 encoding)")
    RUN(30, "v_mad_u64_u32 vgpr*vgpr+vgpr64")
    RUN(31, "v_mad_u64_u32 vgpr*sgpr+vgpr64")
    RUN(32, "v_mad_u64_u32 vgpr*sgpr+0")
    RUN(33, "v_lshl_add_u64 vgpr64")
    RUN(34, "v_lshl_add_u64 sgpr64 addend")
    RUN(35, "v_lshlrev_b64")
    RUN(36, "v_lshrrev_b64")
    RUN(37, "v_fma_f64")
    RUN(38, "v_pk_fma_f32")
    RUN(39, "mad64 + add + and (3 inst/group)")
    RUN(40, "add_co+addc pair (2 inst/group)")

    // hazard checks
    {
        const int n = 1 << 20;
        std::vector<u64> h(2 * n);
        u64 s = 0x9E3779B97F4A7C15ull;
        for (auto &v : h) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            v = s;
        }
        // force plenty of carries
        for (int i = 0; i < n; i += 3) h[2 * i] |= 0xFFFFFFFF00000000ull >> (i % 33);
        u64 *din, *dout;
        CK(hipMalloc(&din, h.size() * 8));
        CK(hipMalloc(&dout, (size_t)3 * n * 8));
        CK(hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(hazard_kernel, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
        CK(hipDeviceSynchronize());
        std::vector<u64> r((size_t)3 * n);
        CK(hipMemcpy(r.data(), dout, r.size() * 8, hipMemcpyDeviceToHost));
        long bad_add = 0, bad_mad = 0, carries = 0;
        for (int i = 0; i < n; ++i) {
            const u64 x = h[2 * i], y = h[2 * i + 1];
            if (r[3 * i] != x + y) ++bad_add;
            const unsigned __int128 p = (unsigned __int128)(u32)x * (u32)(x >> 32) + y;
            const u64 cy = (u64)(p >> 64);
            carries += cy;
            if (r[3 * i + 1] != (u64)p || r[3 * i + 2] != (u32)((u32)(y >> 32) + cy)) ++bad_mad;
        }
        std::printf("hazard check: add_co/addc back to back: %ld mismatches of %d; mad carry-out -> addc: %ld mismatches (%ld carries)\n",
                    bad_add, n, bad_mad, carries);
    }
    return 0;
}
