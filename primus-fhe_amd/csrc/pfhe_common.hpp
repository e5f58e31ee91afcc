// pfhe_common.hpp — shared declarations of libpfhe_hip (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pfhe.h"

namespace pfhe {

using u64 = unsigned long long;  // same width as uint64_t; matches HIP's 64-bit intrinsics
using u32 = unsigned int;
static_assert(sizeof(u64) == 8 && sizeof(uint64_t) == 8, "64-bit words");

void set_last_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PFHE_HIP(expr)                                                      \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) return ::pfhe::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)

#define PFHE_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != PFHE_OK) return rc_; \
    } while (0)

// The kernels move data as 16-byte vectors: every device buffer handed to a *_dev entry point must
// be 16-byte aligned (hipMalloc returns 256-byte aligned memory; only odd sub-slices can violate it).
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
#define PFHE_REQUIRE_ALIGNED(ptr)                                                  \
    do {                                                                           \
        if (!::pfhe::aligned16(ptr)) {                                             \
            ::pfhe::set_last_error(#ptr " must be 16-byte aligned");               \
            return PFHE_ERR_BAD_ARGUMENT;                                          \
        }                                                                          \
    } while (0)

// true while `s` is being captured into a HIP graph: the internal multi-stream pipelines are then
// replaced by their single-stream forms (same kernels, same results)
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// RAII: make `device` current for the scope, restore the previous device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device);
    ~DeviceGuard();
};

// Per-prime constants + device twiddle tables of one U64NttTable.
// Twiddles are stored interleaved {w, floor(w*2^64/q)} so one 16-byte load fetches both.
//   fwd[i] = roots[i]      (roots[brv(k)] = psi^k,            table.rs:347-351)
//   inv[i] = inv_roots[i]  (inv_roots[brv(k)+1] = psi^-(k+1), table.rs:354-358)
struct NttPrime {
    u64 q, two_q;
    u64 inv_n, inv_n_p;      // N^-1 mod q and its Shoup quotient
    u64 inv_n_w, inv_n_w_p;  // N^-1 * inv_roots[N-1] (table.rs:397-400)
    u64 bar_lo, bar_hi;      // floor(2^128 / q) (BarrettModulus ratio)
    const ulonglong2 *fwd;   // device, N entries
    const ulonglong2 *inv;   // device, N entries
    // pseudo-Mersenne fast path: q = 2^pm_k - pm_c (pm_k == 0: prime does not qualify)
    u32 pm_k, pm_pad;
    u64 pm_c;
    const u64 *fwd_w;        // device: u32 tables only ({w, floor(w*2^32/q)} packed in one word, pfhe_u32.hip)
    const u64 *inv_w;
    const u64 *fwd_wn;       // device: u32 tables only: fwd_w with the twiddle negated, {2^32 - w, floor(w*2^32/q)} (B32Arith::mul1_neg)
    // u32 tables: lane-ordered copies (see fwd_last below) for the register pass at word distances 8..1 and the
    // intra-word stage: 31 * (N/32) entries (15 + 16 per group of 16 words), forward ones negated; null for N < 32
    const u64 *fwd_last_w;
    const u64 *inv_last_w;
    const ulonglong2 *fwd_p; // device, N entries {w, w*2^32 mod q}: twiddles of the pseudo-Mersenne path, same
    const ulonglong2 *inv_p; // indexing as fwd / inv (null when the prime does not qualify)
    u64 inv_n_2, inv_n_w_2;  // inv_n * 2^32 mod q, inv_n_w * 2^32 mod q
    u64 q3;                  // 3q (the multiple of q the pseudo-Mersenne butterflies subtract from)
    // Twiddles of the four stages at distances 8, 4, 2, 1 (the register pass of a block pass in which a thread owns 16
    // consecutive coefficients), re-ordered so that the 64 lanes of a wave load 64 consecutive entries: stage at distance
    // 2^j (j = 3..0) has 2^(3-j) twiddles per group g of 16 coefficients; entry ((2^(3-j) - 1) + u) * (N/16) + g holds
    // fwd[(N >> (j+1)) + g * 2^(3-j) + u] (inv_last: inv[1 + N - (N >> j) + g * 2^(3-j) + u]).  15N/16 entries of the
    // kind the arithmetic policy in use wants ({w, Shoup quotient} or {w, w * 2^32 mod q}); null for N < 16.
    const ulonglong2 *fwd_last;
    const ulonglong2 *inv_last;
    // Montgomery-form twiddles of the generic-prime transforms (MontArith; null unless the table takes that path):
    // {w * 2^32 mod q, w * 2^64 mod q}, same indexing as fwd / inv, and their lane-ordered copies
    const ulonglong2 *fwd_m;
    const ulonglong2 *inv_m;
    const ulonglong2 *fwd_last_m;
    const ulonglong2 *inv_last_m;
    u64 inv_n_m, inv_n_m2, inv_n_w_m, inv_n_w_m2;  // N^-1 and N^-1 * w in the same form
    u32 qinv32;                                    // -q^-1 mod 2^32
    u32 mont_qest;                                 // floor(2^(bits(q) - 1) / (floor(q / 2^32) + 1)) (MontArith::canon)
    u64 mont_qf;                                   // the largest multiple of q that is <= 2^63 (MontArith::fold)
};

// MontArith covers odd primes in [2^48, 2^61): forward lazy values stay below 2^63 + 3q < 2^64, and the quotient estimate
// of its closing reduction reads the high word of q only, which is exact enough from 2^48 (smaller primes keep the
// Shoup transforms)
inline bool mont_shape(u64 q) { return (q & 1) && q < (1ull << 61) && q >= (1ull << 48); }

// q = 2^K - c qualifies for PmArith when 40 <= K <= 61 and c < 2^(K-33)
inline bool pm_shape(u64 q, u32 &k, u64 &c) {
    k = 64 - (u32)__builtin_clzll(q);
    if ((q & (q - 1)) == 0) return false;
    c = (1ull << k) - q;
    return k >= 40 && k <= 61 && c < (1ull << (k - 33));
}

struct HostTable {  // host-side result of table construction
    u32 log_n = 0;
    u64 q = 0, root = 0, inv_root = 0, inv_n = 0, inv_n_w = 0;
    u64 bar_lo = 0, bar_hi = 0;
    std::vector<ulonglong2> fwd, inv;  // N each
    std::vector<u64> ordinal;          // psi^k, k < 2N (monomial transforms)
};

// Builds the table; returns a pfhe_status mirroring NttTable::new's errors.
int build_host_table(u32 log_n, u64 q, HostTable &out);

}  // namespace pfhe
