/*
 * pfhe_oracle_avx512.c — AVX-512 (DQ) restatement of the reference's vectorised forward and inverse NTT
 * for 64-bit primes (TEST / BENCH INFRASTRUCTURE ONLY, see pfhe_oracle.h).
 *
 * On an AVX-512 host U64NttTable dispatches to its HEXL-style kernels
 * (primus_ntt/src/ntt/prime64/table.rs:408-418); this file restates the BIT_SHIFT = 64 path:
 *   butterfly          prime64/avx512/butterfly.rs:10-57   (Harvey, approximate quotient)
 *   mulhi_approx       prime64/avx512/utils/arithmetic.rs:94-127
 *   small_mod          x -> min(x, x - 2q)                  (utils/arithmetic.rs)
 *   stage structure    prime64/avx512/transform.rs:13-260, stages.rs: depth-first splitting down
 *                      to 1024 points, then breadth-first T8 stages and the shuffled T4/T2/T1 stages.
 *   inverse            prime64/avx512/transform.rs:205-423 (depth-first above 1024 points, T1/T2/T4 shuffled
 *                      stages then T8 stages, last stage fused with N^-1 and N^-1*w and the [0,2q) -> [0,q)
 *                      reduction), butterfly.rs:58-117 (inv_butterfly), exact 64-bit mulhi in the last stage
 *                      (utils/arithmetic.rs:19-60)
 * Unlike the reference it does not pre-expand the twiddle tables for T4/T2/T1 (it permutes the
 * loaded twiddles instead); the arithmetic per butterfly is the same.  Canonical outputs are
 * identical to the scalar path's (tests/test_oracle_avx512.py); it exists so that bench.py's
 * cpu_baseline is not handicapped against what the real reference would run on this host.
 * The body (pfhe_oracle_avx512_impl.h) is compiled twice: BIT_SHIFT = 64 (DQ) and BIT_SHIFT = 52 (IFMA, q < 2^50:
 * butterfly.rs:30-35,97-102, utils/arithmetic.rs:62-75,141-165, transform.rs:388-397).
 */
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>

#include "pfhe_oracle.h"

int orc_avx512_available(void) {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
}

int orc_avx512_ifma_available(void) {
    return orc_avx512_available() && __builtin_cpu_supports("avx512ifma");
}

/* BIT_SHIFT = 64 */
#define SHIFTV 64
#define FN(name) name##_dq
#define TGT __attribute__((target("avx512f,avx512dq")))
#define AVAILABLE orc_avx512_available
#include "pfhe_oracle_avx512_impl.h"
#undef SHIFTV
#undef FN
#undef TGT
#undef AVAILABLE

/* BIT_SHIFT = 32 (q < 2^30) */
#define SHIFTV 32
#define FN(name) name##_dq32
#define TGT __attribute__((target("avx512f,avx512dq")))
#define AVAILABLE orc_avx512_available
#include "pfhe_oracle_avx512_impl.h"
#undef SHIFTV
#undef FN
#undef TGT
#undef AVAILABLE

/* BIT_SHIFT = 52 (IFMA) */
#define SHIFTV 52
#define FN(name) name##_ifma
#define TGT __attribute__((target("avx512f,avx512dq,avx512ifma")))
#define AVAILABLE orc_avx512_ifma_available
#include "pfhe_oracle_avx512_impl.h"
#undef SHIFTV
#undef FN
#undef TGT
#undef AVAILABLE

/* the reference's ladder (table.rs:166-232 forward, :236-302 inverse): IFMA (52) when the CPU has it and q < 2^50, else
 * the DQ backend with 32-bit preconditioners for q < 2^30 and 64-bit ones otherwise */
static int pick_shift(const orc_u64_ntt *t, int shift) {
    if (shift == 52 || shift == 64 || shift == 32) return shift;
    if (orc_avx512_ifma_available() && orc_u64_ntt_modulus(t) < (1ull << 50) && orc_u64_ntt_roots_precon52(t) != 0) return 52;
    return orc_u64_ntt_modulus(t) < (1ull << 30) && orc_u64_ntt_roots_precon32(t) != 0 ? 32 : 64;
}

int orc_u64_ntt_forward_avx512_shift(const orc_u64_ntt *t, uint64_t *values, int lazy, int shift) {
    switch (pick_shift(t, shift)) {
        case 52: return orc_u64_ntt_forward_avx512_ifma(t, values, lazy);
        case 32: return orc_u64_ntt_forward_avx512_dq32(t, values, lazy);
        default: return orc_u64_ntt_forward_avx512_dq(t, values, lazy);
    }
}
int orc_u64_ntt_inverse_avx512_shift(const orc_u64_ntt *t, uint64_t *values, int lazy, int shift) {
    switch (pick_shift(t, shift)) {
        case 52: return orc_u64_ntt_inverse_avx512_ifma(t, values, lazy);
        case 32: return orc_u64_ntt_inverse_avx512_dq32(t, values, lazy);
        default: return orc_u64_ntt_inverse_avx512_dq(t, values, lazy);
    }
}
int orc_u64_ntt_forward_avx512(const orc_u64_ntt *t, uint64_t *values, int lazy) {
    return orc_u64_ntt_forward_avx512_shift(t, values, lazy, 0);
}
int orc_u64_ntt_inverse_avx512(const orc_u64_ntt *t, uint64_t *values, int lazy) {
    return orc_u64_ntt_inverse_avx512_shift(t, values, lazy, 0);
}

/* `count` consecutive polynomials in one call (bench.py's CPU legs: keeps the Python dispatch out of the timing) */
int orc_u64_ntt_forward_avx512_batch(const orc_u64_ntt *t, uint64_t *values, size_t count, int lazy, int shift) {
    const size_t n = orc_u64_ntt_n(t);
    for (size_t i = 0; i < count; ++i) {
        const int rc = orc_u64_ntt_forward_avx512_shift(t, values + i * n, lazy, shift);
        if (rc) return rc;
    }
    return ORC_OK;
}
