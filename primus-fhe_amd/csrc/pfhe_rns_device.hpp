// pfhe_rns_device.hpp — device functions of the RNS / gadget steps, shared by pfhe_rns.hip (one
// kernel per reference slice function) and pfhe_extprod.hip (fused kernels).
#pragma once

#include <type_traits>

#include "pfhe_modmath.hpp"
#include "pfhe_rns.hpp"

namespace pfhe {

// constants held by value (RnsDev / BasisDev): the kernel is compiled for the exact limb count; device-table forms
// (RnsWide / BasisWide) are compiled for a rounded-up count and address memory with the run-time value_len
template <class T>
constexpr bool kByValue = std::is_same<T, RnsDev>::value || std::is_same<T, BasisDev>::value;

// limb count a kernel is instantiated for: the exact one up to kMaxLimbs, then the next multiple of four
template <template <int> class F, class... A>
int dispatch_len(u32 len, A &&...a) {
    switch (len <= (u32)kMaxLimbs ? len : (len + 3u) & ~3u) {
        case 1: return F<1>::run(a...);
        case 2: return F<2>::run(a...);
        case 3: return F<3>::run(a...);
        case 4: return F<4>::run(a...);
        case 5: return F<5>::run(a...);
        case 6: return F<6>::run(a...);
        case 7: return F<7>::run(a...);
        case 8: return F<8>::run(a...);
        case 12: return F<12>::run(a...);
        case 16: return F<16>::run(a...);
        case 20: return F<20>::run(a...);
        case 24: return F<24>::run(a...);
        case 28: return F<28>::run(a...);
        case 32: return F<32>::run(a...);
    }
    set_last_error("unsupported big-integer length");
    return PFHE_ERR_UNSUPPORTED;
}

// Mixed-radix form of the CRT lift (RnsDev::garner): digits v0 = r0, v1 = (r1 - v0) / q0 mod q1,
// v2 = ((r2 - v0) / q0 - v1) / q1 mod q2, then x = v0 + q0*v1 + q0*q1*v2 < Q — no comparison with Q, no subtraction.
template <int LEN>
__device__ __forceinline__ void compose_garner(const RnsDev &R, const u64 *r, u64 (&v)[LEN]) {
    typedef unsigned __int128 u128;
    const u64 v0 = r[0];
    // (max q < 2 min q: one conditional subtraction reduces a digit into another modulus' range)
    u64 t = sub_mod(r[1], reduce_once(v0, R.q[1]), R.q[1]);
    const u64 v1 = mul_shoup(t, R.g_inv[1][0], R.g_inv_p[1][0], R.q[1]);
    u128 lo = (u128)R.q[0] * v1 + v0;  // < q0*q1 < 2^124
    u64 w2 = 0;
    if (R.L == 3) {
        t = sub_mod(r[2], reduce_once(v0, R.q[2]), R.q[2]);
        t = mul_shoup(t, R.g_inv[2][0], R.g_inv_p[2][0], R.q[2]);
        t = sub_mod(t, reduce_once(v1, R.q[2]), R.q[2]);
        const u64 v2 = mul_shoup(t, R.g_inv[2][1], R.g_inv_p[2][1], R.q[2]);
        const u128 m0 = (u128)R.g_prod[0] * v2, m1 = (u128)R.g_prod[1] * v2;
        const u128 s0 = (u128)(u64)lo + (u64)m0;
        const u128 s1 = (u128)(u64)(lo >> 64) + (u64)(m0 >> 64) + (u64)m1 + (u64)(s0 >> 64);
        lo = ((u128)(u64)s1 << 64) | (u64)s0;
        w2 = (u64)(m1 >> 64) + (u64)(s1 >> 64);
    }
    v[0] = (u64)lo;
    if constexpr (LEN > 1) v[1] = (u64)(lo >> 64);
    if constexpr (LEN > 2) v[2] = w2;
#pragma unroll
    for (int j = 3; j < LEN; ++j) v[j] = 0;
}

// v (LEN limbs, canonical in [0,Q)) = CRT lift of the residues fetch(0..L)  — base.rs:609-633, the general form:
// v += P_i * (inv_i * r_i mod q_i), minus Q whenever the sum overflows or reaches Q.  RT: RnsDev or RnsWide.
template <int LEN, class RT, class Fetch>
__device__ __forceinline__ void compose_general(const RT &R, Fetch fetch, u64 (&v)[LEN]) {
#pragma unroll
    for (int j = 0; j < LEN; ++j) v[j] = 0;
    for (u32 i = 0; i < R.L; ++i) {
        const u64 t = mul_shoup(fetch(i), R.inv(i), R.inv_p(i), R.modulus(i));
        // v += P_i * t  (LEN limbs + carry word)
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < LEN; ++j) {
            const u64 p = R.punctured(i, j);
            const u64 lo = p * t;
            const u64 hi = mulhi64(p, t);
            u64 s = v[j] + lo;
            u64 c1 = s < lo;
            u64 s2 = s + carry;
            c1 += s2 < carry;
            v[j] = s2;
            carry = hi + c1;
        }
        // if carry != 0 or v >= Q: v -= Q
        bool ge = carry != 0;
        if (!ge) {
            ge = true;  // equal counts as >=
#pragma unroll
            for (int j = LEN - 1; j >= 0; --j) {
                if (v[j] != R.product(j)) {
                    ge = v[j] > R.product(j);
                    break;
                }
            }
        }
        if (ge) {
            u64 borrow = 0;
#pragma unroll
            for (int j = 0; j < LEN; ++j) {
                const u64 qj = R.product(j);
                const u64 d = v[j] - qj;
                const u64 b1 = v[j] < qj;
                const u64 d2 = d - borrow;
                const u64 b2 = d < borrow;
                v[j] = d2;
                borrow = b1 | b2;
            }
        }
    }
}

// the by-value form with its residues already in registers
template <int LEN>
__device__ __forceinline__ void compose(const RnsDev &R, const u64 *r, u64 (&v)[LEN]) {
    if (R.garner) {
        compose_garner<LEN>(R, r, v);
        return;
    }
    compose_general<LEN>(R, [&](u32 i) { return r[i]; }, v);
}

// basis.rs:334-349: if v >= threshold: v += add ; returns the initial carry bit.  BT: BasisDev or BasisWide.
template <int LEN, class BT>
__device__ __forceinline__ u32 init_value_carry(const BT &B, u64 (&v)[LEN]) {
    if (B.mode & 2u) {
        bool ge = true;
#pragma unroll
        for (int j = LEN - 1; j >= 0; --j) {
            if (v[j] != B.split(j)) {
                ge = v[j] > B.split(j);
                break;
            }
        }
        if (ge) {
            u64 carry = 0;
#pragma unroll
            for (int j = 0; j < LEN; ++j) {
                const u64 s = v[j] + B.addend(j);
                const u64 c1 = s < v[j];
                const u64 s2 = s + carry;
                const u64 c2 = s2 < s;
                v[j] = s2;
                carry = c1 | c2;
            }
        }
    }
    u32 carry = 0;
    if (B.mode & 1u) {
        u64 limb = 0;
#pragma unroll
        for (int j = 0; j < LEN; ++j)
            if ((u32)j == B.carry_index) limb = v[j];
        carry = (limb & B.carry_bit_mask) != 0;
    }
    return carry;
}

// ---- big integers in the caller's word type WT (u64, or u32 for the <u32> instantiations of the reference generics) ----
// words per big integer as a kernel addresses memory: compile-time for by-value constants with 64-bit words
template <int LEN, class WT, class CT>
__device__ __forceinline__ u32 words_of(const CT &C) {
    if constexpr (sizeof(WT) == 8 && kByValue<CT>) return (u32)LEN;
    else if constexpr (sizeof(WT) == 8) return C.value_len;
    else return C.value_words;
}
// v (LEN 64-bit limbs) <- the `words` WT words at p; limbs beyond them are zero
template <int LEN, class WT>
__device__ __forceinline__ void load_limbs(const WT *p, u32 words, u64 (&v)[LEN]) {
#pragma unroll
    for (int j = 0; j < LEN; ++j) {
        if constexpr (sizeof(WT) == 8) {
            v[j] = (u32)j < words ? p[j] : 0;
        } else {
            const u64 lo = (u32)(2 * j) < words ? p[2 * j] : 0u, hi = (u32)(2 * j + 1) < words ? p[2 * j + 1] : 0u;
            v[j] = lo | (hi << 32);
        }
    }
}
template <int LEN, class WT>
__device__ __forceinline__ void store_limbs(WT *p, u32 words, const u64 (&v)[LEN]) {
#pragma unroll
    for (int j = 0; j < LEN; ++j) {
        if constexpr (sizeof(WT) == 8) {
            if ((u32)j < words) p[j] = v[j];
        } else {
            if ((u32)(2 * j) < words) p[2 * j] = (WT)v[j];
            if ((u32)(2 * j + 1) < words) p[2 * j + 1] = (WT)(v[j] >> 32);
        }
    }
}
// word j of the moduli product in the caller's word type
template <class WT, class RT>
__device__ __forceinline__ u64 product_word(const RT &R, u32 j) {
    if constexpr (sizeof(WT) == 8) return R.product(j);
    else return (R.product(j >> 1) >> (32u * (j & 1u))) & 0xffffffffull;
}
// window of log_basis bits starting at bit `start` of a big integer stored as WT words — common.rs:132-140
// (log_basis < bits of WT, so the window spans at most two words)
template <class WT>
__device__ __forceinline__ u64 window_words(const WT *v, u32 start, u64 mask, u32 log_basis) {
    constexpr u32 WB = 8 * sizeof(WT);
    const u32 idx = start / WB, shr = start % WB;
    u64 w = (u64)v[idx] >> shr;
    if (shr + log_basis > WB) w |= (u64)v[idx + 1] << (WB - shr);
    return w & mask;
}

// window of log_basis bits starting at bit `start` of the LEN-limb value — common.rs:132-140
template <int LEN>
__device__ __forceinline__ u64 window(const u64 (&v)[LEN], u32 start, u64 mask, u32 log_basis) {
    const u32 idx = start >> 6, shr = start & 63;
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < LEN; ++j) {
        if ((u32)j == idx) lo = v[j];
        if ((u32)j == idx + 1) hi = v[j];
    }
    u64 w = lo >> shr;
    if (shr + log_basis > 64) w |= hi << (64 - shr);
    return w & mask;
}

}  // namespace pfhe
