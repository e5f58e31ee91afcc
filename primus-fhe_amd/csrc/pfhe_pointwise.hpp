// pfhe_pointwise.hpp — launchers of the streaming kernels (pfhe_pointwise.hip).
#pragma once
#include "pfhe_common.hpp"

namespace pfhe {
// out = a*b (+ c when c != nullptr); out may alias a and/or c.  b has len_b words: len or one unit;
// with group_words != 0, b holds one unit per `group_words` consecutive words of a (len_b = len / group * unit).
// pm: every prime has the pseudo-Mersenne shape (TableSet::pm) -> folding multiply instead of Barrett.
int pointwise_dev(u64 *out, const u64 *a, const u64 *b, const u64 *c, const NttPrime *primes, u32 L, u32 log_n,
                  u64 len, u64 len_b, hipStream_t s, u64 group_words = 0, bool pm = false);
// (a, b) = (a + s, (a - s) * w); w holds len_w multiplicands (ShoupFactor pairs when `factor`), shared
// cyclically by the batch.
int butterfly_dev(bool factor, u64 *a, const u64 *s, const u64 *w, u64 *b, const NttPrime *primes, u32 L, u32 log_n,
                  u64 len, u64 len_w, hipStream_t st, bool pm = false);
int fill_uniform_dev(u64 *dst, u64 len, const u64 *moduli_dev, u64 count, u64 poly_len, u64 seed,
                     hipStream_t s);
// Per-limb monomial coefficient and its Shoup quotient, passed to the kernel BY VALUE: the device entry points
// then need no allocation, copy or synchronisation and can be captured into a HIP graph.
constexpr u32 kMaxMonomialLimbs = 16;
struct MonomialScalars {
    u64 value[kMaxMonomialLimbs];
    u64 quotient[kMaxMonomialLimbs];
};
int monomial_dev(u64 *out, const NttPrime *primes, u32 L, u32 log_n, u64 degree, const MonomialScalars &sc,
                 hipStream_t s);
}  // namespace pfhe
