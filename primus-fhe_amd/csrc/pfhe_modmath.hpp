// pfhe_modmath.hpp — 64-bit modular arithmetic shared by host table code and device kernels.
//
// Device functions are the GPU counterpart of primus_reduce / primus_modulus / primus_factor:
//   mul_shoup_lazy   = ShoupFactor::lazy_factor_mul_modulo (primus_factor/src/shoup_factor/mod.rs:124)
//                    = mul_mod_lazy (primus_ntt/src/ntt/prime64/scalar/arithmetic.rs:32-35)
//   barrett_reduce128= BarrettModulus::reduce_wide (primus_modulus/src/barrett/mod.rs:99-139)
//   add_mod/sub_mod  = compact::reduce_add / reduce_sub (common/compact/primitive.rs:10-39)
// The modulus and its precomputations are kernel-uniform values (SGPRs) — the GPU analogue of
// #[derive(Barrett)]'s compile-time constants (primus_barrett_derive/src/lib.rs:28-40).
#pragma once

#include <hip/hip_runtime.h>

namespace pfhe {

using u64 = unsigned long long;
using u32 = unsigned int;

#define PFHE_HD __host__ __device__ __forceinline__

PFHE_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

PFHE_HD u64 min_u64(u64 a, u64 b) { return a < b ? a : b; }

// x mod m for x < 2m
PFHE_HD u64 reduce_once(u64 x, u64 m) { return min_u64(x, x - m); }

// w*y mod q in [0,2q) for any 64-bit y, given wp = floor(w*2^64/q), q < 2^62
PFHE_HD u64 mul_shoup_lazy(u64 y, u64 w, u64 wp, u64 q) { return w * y - q * mulhi64(wp, y); }

PFHE_HD u64 mul_shoup(u64 y, u64 w, u64 wp, u64 q) { return reduce_once(mul_shoup_lazy(y, w, wp, q), q); }

PFHE_HD u64 add_mod(u64 a, u64 b, u64 q) { return reduce_once(a + b, q); }
PFHE_HD u64 sub_mod(u64 a, u64 b, u64 q) {
    u64 d = a - b;
    return min_u64(d, d + q);
}

// (hi:lo) mod q, canonical, for hi:lo < q * 2^64 (always true for a*b+c with a,b,c < q < 2^62).
// mu = floor(2^128/q) as (mu_hi:mu_lo).  Quotient estimate = floor((hi:lo)*mu / 2^128) is at
// most 1 short of the true quotient, so one conditional subtraction finishes.
PFHE_HD u64 barrett_reduce128(u64 lo, u64 hi, u64 q, u64 mu_lo, u64 mu_hi) {
    // carries into the top word only
    u64 a_hi = mulhi64(lo, mu_lo);
    u64 b_lo = lo * mu_hi, b_hi = mulhi64(lo, mu_hi);
    u64 c_lo = hi * mu_lo, c_hi = mulhi64(hi, mu_lo);
    u64 s = b_lo + a_hi;
    u64 carry1 = s < a_hi;
    u64 s2 = s + c_lo;
    u64 carry2 = s2 < c_lo;
    u64 qhat = hi * mu_hi + b_hi + c_hi + carry1 + carry2;
    u64 r = lo - qhat * q;
    return reduce_once(r, q);
}

PFHE_HD u64 mul_mod_barrett(u64 a, u64 b, u64 q, u64 mu_lo, u64 mu_hi) {
    return barrett_reduce128(a * b, mulhi64(a, b), q, mu_lo, mu_hi);
}

// (a*b + c) mod q — BarrettModulus::reduce_mul_add (barrett/ops.rs:308-315)
PFHE_HD u64 mul_add_mod_barrett(u64 a, u64 b, u64 c, u64 q, u64 mu_lo, u64 mu_hi) {
    u64 lo = a * b, hi = mulhi64(a, b);
    u64 lo2 = lo + c;
    hi += (lo2 < lo);
    return barrett_reduce128(lo2, hi, q, mu_lo, mu_hi);
}

}  // namespace pfhe
