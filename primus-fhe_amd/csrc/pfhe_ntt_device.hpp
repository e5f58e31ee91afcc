// pfhe_ntt_device.hpp — device building blocks of the NTT kernels (included by .hip files only).
#pragma once

#include <type_traits>

#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"
#if defined(__HIPCC__)
#include "pfhe_mont_asm.hpp"
#include "pfhe_pm_asm.hpp"
#endif

namespace pfhe {

constexpr u32 kMaxSinglePassLog = 14;  // largest N handled by one block pass (128 KiB in LDS)
constexpr int kTwoPassBlockLog = 12;   // block size used under strided passes

struct NttPlan {
    bool tiny = false;
    int n_strided = 0;
    int strided[4] = {0, 0, 0, 0};  // stages per strided pass, in forward order
    int block_log = 0;
};

// Tuning switches of the transforms.  The PFHE_* environment variables are read ONCE, when a table handle is
// created (NttTuning::from_env), and travel inside the handle: nothing on the launch path calls getenv, and two
// tables created under different settings keep their own.
struct NttTuning {
    int pipe_tiles = 0;        // PFHE_PIPE_TILES: tiles of the pipelined form (0: built-in default, batch / 256 MiB)
    bool pipelined = true;     // PFHE_DISABLE_PIPELINED clears it: N = 2^16 runs one launch per pass instead of tiles + 1 launches of ntt_pipe_{fwd,inv}_kernel
    int pipelined_min_mb = 0;  // PFHE_PIPELINED_MIN_MB: smallest batch (MiB of data) that takes the pipelined form (0: built-in default)
    static NttTuning from_env();
};
NttPlan make_ntt_plan(u32 log_n, int arith = 0, const NttTuning &tune = NttTuning());  // arith: see kArith* below

// `arith` selects the arithmetic policy: kArithShoup (any q < 2^62), kArithMont (every prime below 2^61: the
// transforms of generic primes, NttPrime::fwd_m), kArithPm (every prime of the
// table has the pseudo-Mersenne shape, NttPrime::pm_k) or kArithB32 (32-bit tables: `data` holds
// pairs of u32 coefficients and log_n counts 64-bit WORDS, i.e. log2(N) - 1).  A bool converts to
// the first two.
enum : int { kArithShoup = 0, kArithPm = 1, kArithB32 = 2, kArithMont = 3 };
int ntt_forward_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, bool lazy,
                    hipStream_t s, const NttTuning &tune = NttTuning());
// the form a transform of npolys limb-polynomials takes (kernel or form name into buf) and its number of launches
int ntt_transform_form(u32 L, u32 log_n, int arith, u64 npolys, bool inverse, const NttTuning &tune, char *buf, size_t cap);
int ntt_inverse_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, bool lazy,
                    hipStream_t s, const NttTuning &tune = NttTuning());
// transform of a batch in device-visible HOST memory `io` (pinned / registered) without copy engines: the first pass reads
// `io`, the last writes it, the intermediate of two-pass rings stays in `scratch` (device memory of the same size).
// PFHE_ERR_UNSUPPORTED: shape not covered (u32 tables, more than one strided pass).  64-bit policies only.
int ntt_transform_through_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *io, u64 *scratch, u64 npolys,
                              bool inverse, bool lazy, hipStream_t s, const NttTuning &tune = NttTuning());
// inverse transform of data (*) mul, the pointwise product fused into the loads of the first
// (block) pass; `mul` holds mul_polys limb-polynomials (npolys, or one unit of L shared by the batch).
// 64-bit policies only.  `mul` (and `data`) must be CANONICAL, [0, q): the product is A::mul_any, one reduction of a
// 128-bit product whose precondition is a * b < q * 2^64.
int ntt_inverse_mul_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, const u64 *mul,
                        u64 mul_polys, hipStream_t s, const NttTuning &tune = NttTuning());
// data <- INTT(NTT(data) (*) mul) with the forward block pass, the product and the inverse block pass fused in one kernel
// (one HBM round trip for N <= 2^14, three for two-pass rings).  PFHE_ERR_UNSUPPORTED: shape not covered, use
// ntt_forward_dev + ntt_inverse_mul_dev.  64-bit policies only.  `mul` must be CANONICAL, [0, q): the forward half hands
// RAW words (up to 2^63 + 3q for Montgomery tables, 2^63 + 2^32 for pseudo-Mersenne ones) to A::mul_any, whose
// precondition a * b < q * 2^64 then holds only for b < q (a lazily transformed multiplicand in [0, 4q) breaks it).
int ntt_polymul_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, const u64 *mul,
                    u64 mul_polys, hipStream_t s, const NttTuning &tune = NttTuning());
// U32NttTable / U32DcrtTable transforms (log_n = log2 of the polynomial length in coefficients)
int ntt32_transform_dev(const NttPrime *primes, u32 L, u32 log_n, u32 *data, u64 npolys, bool inverse, bool lazy,
                        hipStream_t s, const NttTuning &tune = NttTuning());

int ntt_num_passes(u32 log_n, int arith = 0, const NttTuning &tune = NttTuning());
void ntt_pass_name(u32 log_n, bool inverse, int index, char *buf, size_t cap, int arith = 0,
                   const NttTuning &tune = NttTuning());
int ntt_pass_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, bool inverse,
                 int index, bool lazy, hipStream_t s, const u64 *mul = nullptr, u64 mul_polys = 0,
                 const NttTuning &tune = NttTuning());

#if defined(__HIPCC__)

// Wave-local LDS exchanges without a workgroup barrier, readfirstlane'd twiddle indices and the 64-bit carry masks
// of the inline asm all assume 64-lane wavefronts.
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "libpfhe_hip is written for wave64 targets (gfx950)"
#endif

// Tables are reached through pointers stored in NttPrime, which the compiler would treat as
// generic (flat) pointers: flat loads tick both vmcnt and lgkmcnt and serialise against LDS
// traffic.  Reading through explicit global-address-space pointers yields plain global_load.
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef const u64x2 __attribute__((address_space(1))) *GCVec2Ptr;
typedef u64x2 __attribute__((address_space(1))) *GVec2Ptr;
typedef const u64 __attribute__((address_space(1))) *GCWordPtr;
// constant address space: a load whose address is wave-uniform becomes a scalar load (s_load_dwordx4) whatever the
// alias analysis concludes; with a per-lane address it is an ordinary global load.  The tables are never written
// while a kernel runs.
typedef const u64x2 __attribute__((address_space(4))) *CCVec2Ptr;

// An opaque copy of the thread id (no instruction): addresses computed from it cannot be hoisted above this point, so
// the address registers of a later phase are not live through an earlier one.
__device__ __forceinline__ u32 opaque_tid() {
    u32 t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// x mod m for x < 2m, using the borrow of the subtraction as the select condition
__device__ __forceinline__ u64 csub(u64 x, u64 m) {
    // four instructions (subtract with borrow, two selects on the borrow); the compiler's lowering of the
    // overflow intrinsic compares separately and takes five: -2 % VALU instructions in the pseudo-Mersenne block
    // pass, -6 % in the Shoup one (4.57 -> 4.39 ms per 12 288 NTTs)
    u32 d0, d1;
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), m0 = (u32)m, m1 = (u32)(m >> 32);
    asm("v_sub_co_u32 %0, vcc, %2, %4\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc\n\t"
        "v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
        : "=&v"(d0), "=&v"(d1)
        : "v"(x0), "v"(x1), "v"(m0), "v"(m1)
        : "vcc");
    return ((u64)d1 << 32) | d0;
}

// ------------------------------------------------------------------------------------------
// Arithmetic policies.  Both provide w*y mod q in [0,2q) for any y < 2^63 ("lazy multiply"), which
// is all the Harvey / Gentleman-Sande butterflies need (scalar/arithmetic.rs:32-79).
//
// ShoupArith — any prime q < 2^62: the reference's own scheme, ShoupFactor / mul_mod_lazy
//   (primus_factor/src/shoup_factor/mod.rs:124-131): 10 32-bit multiplies per product.
// PmArith — primes of the shape q = 2^K - c with 40 <= K <= 61 and c < 2^(K-33) (every
//   "largest NTT-friendly prime below a power of two", incl. the reference's 50/60-bit test
//   moduli): 2^K = c (mod q), so the 128-bit product is folded twice with the small constant c:
//   7 multiplies, no precomputed quotient (8-byte twiddles).  Exact integer arithmetic: canonical
//   outputs are identical to the Shoup path's, lazy outputs agree mod q.
// ------------------------------------------------------------------------------------------
struct ShoupArith {
    struct Tw {
        u64 w, wp;
    };
    u64 q, two_q;
    GCVec2Ptr fwd, inv, fwd_last, inv_last;
    u64 inv_n, inv_n_p, inv_n_w, inv_n_w_p;
    u64 bar_lo, bar_hi;

    __device__ __forceinline__ explicit ShoupArith(const NttPrime *__restrict__ P)
        : q(P->q), two_q(P->two_q), fwd((GCVec2Ptr)(const void *)P->fwd), inv((GCVec2Ptr)(const void *)P->inv),
          fwd_last((GCVec2Ptr)(const void *)P->fwd_last), inv_last((GCVec2Ptr)(const void *)P->inv_last),
          inv_n(P->inv_n), inv_n_p(P->inv_n_p), inv_n_w(P->inv_n_w), inv_n_w_p(P->inv_n_w_p), bar_lo(P->bar_lo),
          bar_hi(P->bar_hi) {}
    // lane-ordered twiddles of the stages at distances 8..1 (NttPrime::fwd_last / inv_last)
    static constexpr bool kLastTables = true;
    __device__ __forceinline__ Tw fwd_tw_last(u32 off) const {
        const u64x2 v = fwd_last[off];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw inv_tw_last(u32 off) const {
        const u64x2 v = inv_last[off];
        return Tw{v.x, v.y};
    }
    // a*b mod q for two canonical residues (no precomputed quotient): BarrettModulus::reduce_mul
    __device__ __forceinline__ u64 mul_any(u64 a, u64 b) const { return mul_mod_barrett(a, b, q, bar_lo, bar_hi); }
    __device__ __forceinline__ Tw fwd_tw(u32 i) const {
        const u64x2 v = fwd[i];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw inv_tw(u32 i) const {
        const u64x2 v = inv[i];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw tw_inv_n() const { return Tw{inv_n, inv_n_p}; }
    __device__ __forceinline__ Tw tw_inv_n_w() const { return Tw{inv_n_w, inv_n_w_p}; }
    __device__ __forceinline__ u64 mul_lazy(u64 y, Tw t) const { return t.w * y - q * mulhi64(t.wp, y); }
    // x in [0,4q) -> [0,2q)
    __device__ __forceinline__ u64 reduce_x(u64 x) const { return csub(x, two_q); }
    __device__ __forceinline__ u64 reduce_2q(u64 x) const { return csub(x, q); }              // [0,2q) -> [0,q)
    __device__ __forceinline__ u64 reduce_4q(u64 x) const { return csub(csub(x, two_q), q); }  // [0,4q) -> [0,q)
    static constexpr bool kPacked = false;
    static constexpr bool kWide = false;
    static constexpr bool kMont = false;
};

// v_mad_u64_u32 with its carry-out kept (an SGPR pair: one bit per lane), and the add-with-carry that consumes it.
// The compiler's own form of "64-bit sum that may overflow" is a compare + select + 64-bit add (three more
// instructions per twiddle multiply).
__device__ __forceinline__ u64 mad_u64_carry(u32 a, u32 b, u64 c, u64 &carry) {
    u64 d;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ u32 add_carry(u32 a, u64 carry) {
    u32 d;
    u64 unused;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(d), "=s"(unused) : "v"(a), "s"(carry));
    return d;
}
// a - b on 64-bit values as one borrow chain (the compiler puts an s_nop between the two halves)
__device__ __forceinline__ u64 sub_u64(u64 a, u64 b) {
    u32 d0, d1;
    asm("v_sub_co_u32 %0, vcc, %2, %4\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc"
        : "=&v"(d0), "=&v"(d1)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)d1 << 32) | d0;
}

struct PmArith {
    // twiddle w and w2 = w * 2^32 mod q: y*w = y0*w + y1*w2 (mod q) needs four 32x32 products and its sum
    // stays below 2^(K+33), so ONE fold finishes the reduction (five multiply-adds per twiddle product
    // instead of the eight of the full 128-bit product folded twice)
    struct Tw {
        u64 w, w2;
    };
    u64 q, q3;  // q3 = 3q: the multiple of q the butterflies subtract from
    CCVec2Ptr fwd, inv;
    GCVec2Ptr fwd_last, inv_last;
    Tw inv_n, inv_n_w;
    u32 c, c2, sh;  // q = 2^K - c, c2 = 2c, sh = K - 32
    u32 mask;       // 2^(K-32) - 1
    u32 vsh, vmask, vmask1;  // sh, mask and 2^(K-31) - 1 held in VGPRs for the asm butterflies (pfhe_pm_asm.hpp)

    __device__ __forceinline__ explicit PmArith(const NttPrime *__restrict__ P)
        : q(P->q), q3(P->q3), fwd((CCVec2Ptr)(const void *)P->fwd_p), inv((CCVec2Ptr)(const void *)P->inv_p),
          fwd_last((GCVec2Ptr)(const void *)P->fwd_last), inv_last((GCVec2Ptr)(const void *)P->inv_last),
          inv_n{P->inv_n, P->inv_n_2}, inv_n_w{P->inv_n_w, P->inv_n_w_2}, c((u32)P->pm_c), c2(2 * (u32)P->pm_c),
          sh(P->pm_k - 32), mask((1u << (P->pm_k - 32)) - 1), vsh(sh), vmask(mask), vmask1(2 * mask + 1) {
        asm volatile("" : "+v"(vsh), "+v"(vmask), "+v"(vmask1));  // as uniform values they would live in SGPRs
    }
    __device__ __forceinline__ Tw fwd_tw(u32 i) const {
        const u64x2 v = fwd[i];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw inv_tw(u32 i) const {
        const u64x2 v = inv[i];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw tw_inv_n() const { return inv_n; }
    __device__ __forceinline__ Tw tw_inv_n_w() const { return inv_n_w; }
    static constexpr bool kLastTables = true;
    __device__ __forceinline__ Tw fwd_tw_last(u32 off) const {
        const u64x2 v = fwd_last[off];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw inv_tw_last(u32 off) const {
        const u64x2 v = inv_last[off];
        return Tw{v.x, v.y};
    }

    // Twiddle product, y any 64-bit value: S = y0*w + y1*w2 < 2^32 * 2q < 2^(K+33); 2^(K+1) = 2c (mod q), so
    // S = hp*2^(K+1) + L == hp*2c + L =: T with hp < 2^32, L < 2^(K+1), hp*2c < 2^K: T < 3 * 2^K and T <= 3q
    // (c < 2^(K-33)).  Column 0 (y0*w0 + y1*v0) can exceed 64 bits: its carry-out has weight 2^64 = bit 32 of the
    // column-1 sum and is added there.
    __device__ __forceinline__ u64 mul_lazy(u64 y, Tw t) const {
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)t.w, w1 = (u32)(t.w >> 32), v0 = (u32)t.w2,
                  v1 = (u32)(t.w2 >> 32);
        const u64 t0 = (u64)y0 * w0;
        u64 carry;
        const u64 t1 = mad_u64_carry(y1, v0, t0, carry);
        u64 t2 = (u64)y0 * w1 + (t1 >> 32);
        asm("" : "+v"(t2));  // keeps (t1 >> 32) as the addend of this multiply-add (see the note in mul_full)
        const u64 t3 = (u64)y1 * v1 + t2;
        const u32 t3h = add_carry((u32)(t3 >> 32), carry);
        const u32 hp = __builtin_amdgcn_alignbit(t3h, (u32)t3, sh + 1);     // S >> (K+1)
        const u64 lo = ((u64)((u32)t3 & (2 * mask + 1)) << 32) | (u32)t1;   // S mod 2^(K+1)
        return (u64)hp * c2 + lo;
    }
    // a*b mod~ q in [0,2q) for two data words (no precomputation on b), a < 2^63: the full 128-bit product folded
    // twice.  P = b*a < 2^(K+63); P = phi*2^K + plo == phi*c + plo =: R < 2^(K+33); R = rh*2^K + rl == rh*c + rl
    // < 2^K + 2^32*c < 2q.  All partial sums fit 64 bits for a < 2^63, b < 2^K <= 2^61, c < 2^(K-33).
    __device__ __forceinline__ u64 mul_full(u64 y, u64 w) const {
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
        const u64 lo = (u64)w0 * y0;
        u64 mid = (u64)w0 * y1 + (lo >> 32);
        // Opaque to the optimiser, no instruction: without it LLVM re-associates the sum into
        // mad(w0,y1,0); mad(w1,y0,.); 64-bit add of (lo >> 32) — one more instruction per product than
        // feeding (lo >> 32) to the first multiply-add.  PFHE_NO_MID_BARRIER restores the compiler's form;
        // a volatile barrier also pins the schedule and is 6 % slower.
        asm("" : "+v"(mid));
        mid += (u64)w1 * y0;
        const u64 hi = (u64)w1 * y1 + (mid >> 32);
        const u32 l0 = (u32)lo, l1 = (u32)mid, h0 = (u32)hi, h1 = (u32)(hi >> 32);
        const u32 f0 = __builtin_amdgcn_alignbit(h0, l1, sh);  // (P >> K) low word
        const u32 f1 = __builtin_amdgcn_alignbit(h1, h0, sh);  // (P >> K) high word
        const u64 plo = ((u64)(l1 & mask) << 32) | l0;
        const u64 a = (u64)f0 * c + plo;
        const u64 b = (u64)f1 * c + (a >> 32);
        const u32 rh = __builtin_amdgcn_alignbit((u32)(b >> 32), (u32)b, sh);
        const u64 rl = ((u64)((u32)b & mask) << 32) | (u32)a;
        return (u64)rh * c + rl;
    }
    __device__ __forceinline__ u64 mul_any(u64 a, u64 b) const { return mul_full(a, b); }
    // any 64-bit x -> x mod~ q in [0, 2^K + 2^31) (2^K = c folds the bits above K): three
    // instructions without a carry chain, cheaper than the compare-and-subtract of the generic path
    __device__ __forceinline__ u64 reduce_x(u64 x) const {
        const u32 x1 = (u32)(x >> 32);
        const u64 low = ((u64)(x1 & mask) << 32) | (u32)x;
        return (u64)(x1 >> sh) * c + low;
    }
    __device__ __forceinline__ u64 reduce_2q(u64 x) const { return csub(x, q); }
    // any 64-bit x -> [0,q)
    __device__ __forceinline__ u64 canon(u64 x) const { return csub(reduce_x(x), q); }
    __device__ __forceinline__ u64 reduce_4q(u64 x) const { return canon(x); }
    static constexpr bool kPacked = false;
    // "wide" lazy domain: butterfly values are arbitrary 64-bit representatives bounded by the analysis at
    // fwd_bfly / inv_bfly below, not the reference's [0,4q) / [0,2q)
    static constexpr bool kWide = true;
    static constexpr bool kMont = false;
    // forward butterflies with a fold: the shift and the mask of x are C++ in front of the asm block (pfhe_pm_asm.hpp; 16
    // instructions instead of 17 — x is dead after the butterfly, the compiler masks its high half in place and the copy of its
    // low half into the addend pair disappears)
#ifdef PFHE_PM_FOLD_INSIDE  // A/B build (tools/build_variant.sh foldin -DPFHE_PM_FOLD_INSIDE): round 4's 17-instruction form everywhere
    static constexpr bool kFoldOutside = false;
#else
    static constexpr bool kFoldOutside = true;
#endif
};
// the same arithmetic with the fold inside the asm block: for a kernel at its register limit (extprod_small_kernel: 256
// registers; the outside form costs it 28 bytes of scratch per lane)
struct PmArithFoldInside : PmArith {
    using PmArith::PmArith;
    static constexpr bool kFoldOutside = false;
};
template <class A>
struct FoldInsideOf {
    using type = A;
};
template <>
struct FoldInsideOf<PmArith> {
    using type = PmArithFoldInside;
};


// MontArith — odd primes in [2^48, 2^61) (the transforms of tables whose primes are not all pseudo-Mersenne): one-word
//   Montgomery reduction with a split multiplicand.  Twiddles are stored as {wm = w*2^32 mod q, wm2 = w*2^64 mod q}
//   (NttPrime::fwd_m / inv_m); T = (y0*wm + y1*wm2 + m*q) / 2^32 == y*w (mod q), T < 3q for ANY 64-bit y, with
//   m = (low word of the sum) * (-q^-1 mod 2^32): seven 32 x 32 multiplies where the reference's Shoup product
//   (shoup_factor/mod.rs:124-131) takes ten.  Lazy domains: forward [0, 2^63 + 3q), inverse [0, F), F the largest
//   multiple of q below 2^63 (the members below and pfhe_mont_asm.hpp).  Exact
//   integer arithmetic: canonical outputs are the Shoup path's, lazy outputs agree mod q and honour the reference's
//   [0,4q) / [0,2q) contracts.  Products of two data words keep the Barrett form (mul_any).
// low word of a*b + c in ONE instruction (v_mad_u64_u32): the compiler, asked for 32 bits of that sum, emits
// v_mul_lo_u32 + v_add_u32
static __device__ __forceinline__ u32 mad_lo32(u32 a, u32 b, u64 c) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return (u32)d;
}

struct MontArith {
    struct Tw {
        u64 w, w2;
    };
    u64 q, two_q, q3;
    u64 qf;  // F: the largest multiple of q that is <= 2^63 (NttPrime::mont_qf; 4q <= F)
    CCVec2Ptr fwd, inv;
    GCVec2Ptr fwd_last, inv_last;
    Tw inv_n, inv_n_w;
    u64 bar_lo, bar_hi;
    u32 qinv;          // -q^-1 mod 2^32
    // Forward lazy domain [0, 2^63 + 3q): a forward butterfly folds x by adding 2^64 - F under the mask of its sign bit
    // (halves in VGPRs: v_and_b32 with two VGPR operands issues at twice the rate of one that reads an SGPR; the
    // inverse butterflies' carry chains use them too).  Inverse lazy domain [0, F).
    u32 vnqf_0, vnqf_1;
    // closing reduction (canon): quotient estimate k = floor(v_hi * qest / 2^(32 + qest_sh)) + 1 with
    // qest = floor(2^(bits(q) - 1) / (floor(q / 2^32) + 1)) (NttPrime::mont_qest), qest_sh = bits(q) - 33
    u32 qest, qest_sh;
    u64 qest_one, nq;  // 2^(32 + qest_sh); 2^64 - q

    __device__ __forceinline__ explicit MontArith(const NttPrime *__restrict__ P)
        : q(P->q), two_q(P->two_q), q3(P->q3), qf(P->mont_qf), fwd((CCVec2Ptr)(const void *)P->fwd_m),
          inv((CCVec2Ptr)(const void *)P->inv_m), fwd_last((GCVec2Ptr)(const void *)P->fwd_last_m),
          inv_last((GCVec2Ptr)(const void *)P->inv_last_m), inv_n{P->inv_n_m, P->inv_n_m2},
          inv_n_w{P->inv_n_w_m, P->inv_n_w_m2}, bar_lo(P->bar_lo), bar_hi(P->bar_hi), qinv(P->qinv32),
          vnqf_0((u32)(0 - P->mont_qf)), vnqf_1((u32)((0 - P->mont_qf) >> 32)), qest(P->mont_qest),
          qest_sh(31u - (u32)__builtin_clzll(P->q)), qest_one(1ull << (63 - __builtin_clzll(P->q))), nq(0 - P->q) {
        asm volatile("" : "+v"(vnqf_0), "+v"(vnqf_1));  // as uniform values they would live in SGPRs
    }
    static constexpr bool kLastTables = true;
    static constexpr bool kPacked = false;
    static constexpr bool kWide = false;
    static constexpr bool kMont = true;
    __device__ __forceinline__ Tw fwd_tw(u32 i) const {
        const u64x2 v = fwd[i];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw inv_tw(u32 i) const {
        const u64x2 v = inv[i];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw fwd_tw_last(u32 off) const {
        const u64x2 v = fwd_last[off];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw inv_tw_last(u32 off) const {
        const u64x2 v = inv_last[off];
        return Tw{v.x, v.y};
    }
    __device__ __forceinline__ Tw tw_inv_n() const { return inv_n; }
    __device__ __forceinline__ Tw tw_inv_n_w() const { return inv_n_w; }
#if defined(__HIPCC__)
    __device__ __forceinline__ u64 mul_lazy(u64 y, Tw t) const { return mont_mul1<false>(*this, y, t); }  // [0, 3q); twiddle operands in VGPRs: also right when the prime is a per-lane value
#endif
    __device__ __forceinline__ u64 mul_any(u64 a, u64 b) const { return mul_mod_barrett(a, b, q, bar_lo, bar_hi); }
    __device__ __forceinline__ u64 reduce_x(u64 x) const { return csub(x, qf); }                     // [0,2F) -> [0,F)
    __device__ __forceinline__ u64 reduce_2q(u64 x) const { return csub(x, q); }
    __device__ __forceinline__ u64 canon3(u64 x) const { return csub(csub(x, two_q), q); }            // [0,4q) -> [0,q)
    // [0, 2^63 + 3q) -> [0, q), the end of a forward transform: subtract (k + 1) q for an estimate k of the quotient that
    // is exact or one short — k = floor(v_hi * qest / 2^(32 + qest_sh)) never exceeds floor(v / q) (v_hi * 2^32 <= v,
    // (q_hi + 1) * 2^32 > q) and falls short of v / q by less than 1 + (v / q + 1) / q_hi + 2^-7 < 2 for q >= 2^48
    // (v / q <= 2^63 / q + 3, q_hi >= q / 2^32 - 1: the middle term is below 2^95 / q^2 <= 1/2) — which leaves a value in
    // [-q, q), then add q back under the sign mask.  Nine instructions, four of them of the cheap kind, where three
    // conditional subtractions (4q, 2q, q) took twelve.
    __device__ __forceinline__ u64 canon_with(u64 x, u32 q_lo, u32 q_hi, u64 one) const {
        u64 est, t, carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(est), "=s"(carry) : "v"((u32)(x >> 32)), "s"(qest), "v"(one));
        const u32 k = (u32)(est >> 32) >> qest_sh;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(t), "=s"(carry) : "v"(k), "s"((u32)nq), "v"(x));
        const u32 th = mad_lo32(k, (u32)(nq >> 32), t >> 32);
        const u32 m = (u32)((int)th >> 31);
        return (((u64)th << 32) | (u32)t) + (((u64)(m & q_hi) << 32) | (m & q_lo));
    }
    __device__ __forceinline__ u64 canon(u64 x) const { return canon_with(x, (u32)q, (u32)(q >> 32), qest_one); }
    // the same on every register of a thread (the constant of the estimate moved into VGPRs once)
    template <int E>
    __device__ __forceinline__ void canon_regs(u64 (&x)[E]) const {
        u64 one = qest_one;
        asm volatile("" : "+v"(one));
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = canon_with(x[k], (u32)q, (u32)(q >> 32), one);
    }
    // the fold of a forward butterfly: x - F when bit 63 of x is set (below 2^63 afterwards)
    __device__ __forceinline__ u64 fold(u64 x) const {
        const u32 m = (u32)((int)(u32)(x >> 32) >> 31);
        const u64 a = ((u64)(m & vnqf_1) << 32) | (m & vnqf_0);
        u64 r;
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(a));
        return r;
    }
    // the reference's forward lazy contract is [0, 4q); 2^63 may be many multiples of q, so the canonical value it is
    __device__ __forceinline__ u64 fwd_lazy(u64 x) const { return canon(x); }
    __device__ __forceinline__ u64 reduce_4q(u64 x) const { return canon(x); }
};

// B32Arith — the u32 tables (U32NttTable, q < 2^30): a 64-bit word carries the two adjacent
//   coefficients 2i (low half) and 2i+1 (high half), so that every kernel written for 64-bit words
//   moves and shuffles u32 data at full width.  All stages with butterfly distance >= 2 pair word
//   with word and use one twiddle for both halves; the distance-1 stage is the extra "intra-word"
//   stage of the block pass.  Arithmetic is the reference's Barrett-32 lazy multiply
//   (prime32/scalar/arithmetic.rs:16-51) on each half; packed additions never carry across the
//   halves because every lazy value is < 4q < 2^32.
//   Twiddle tables hold {w, floor(w*2^32/q)} as one 64-bit entry.  In word units the forward index
//   formula is unchanged; the inverse one is off by N/2, which NttPrime::inv_w already includes.
struct B32Arith {
    struct Tw {
        u32 w, wp;
    };
    u32 q, two_q32;
    u64 two_q;  // 2q in both halves
    GCWordPtr fwd, inv, fwd_last, inv_last;
    Tw inv_n, inv_n_w;
    static constexpr bool kPacked = true;
    static constexpr bool kWide = false;
    static constexpr bool kMont = false;
    // lane-ordered twiddles of the last register pass and of the intra-word stage (NttPrime::fwd_last_w / inv_last_w)
    static constexpr bool kLastTables = true;

    __device__ __forceinline__ explicit B32Arith(const NttPrime *__restrict__ P)
        : q((u32)P->q), two_q32((u32)P->two_q), two_q(P->two_q | (P->two_q << 32)),
          fwd((GCWordPtr)(const void *)P->fwd_wn), inv((GCWordPtr)(const void *)P->inv_w),
          fwd_last((GCWordPtr)(const void *)P->fwd_last_w), inv_last((GCWordPtr)(const void *)P->inv_last_w),
          inv_n{(u32)P->inv_n, (u32)P->inv_n_p}, inv_n_w{(u32)P->inv_n_w, (u32)P->inv_n_w_p} {}
    static __device__ __forceinline__ Tw unpack(u64 v) { return Tw{(u32)v, (u32)(v >> 32)}; }
    static __device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
    // forward twiddles come from the NEGATED table (NttPrime::fwd_wn: {2^32 - w, floor(w*2^32/q)}): see mul1_neg
    __device__ __forceinline__ Tw fwd_tw(u32 i) const { return unpack(fwd[i]); }
    __device__ __forceinline__ Tw inv_tw(u32 i) const { return unpack(inv[i]); }
    __device__ __forceinline__ Tw fwd_tw_last(u32 off) const { return unpack(fwd_last[off]); }
    __device__ __forceinline__ Tw inv_tw_last(u32 off) const { return unpack(inv_last[off]); }
    __device__ __forceinline__ Tw tw_inv_n() const { return inv_n; }
    __device__ __forceinline__ Tw tw_inv_n_w() const { return inv_n_w; }
    // low word of a*b + c in ONE instruction (v_mad_u64_u32): the compiler, asked for 32 bits of that sum, emits
    // v_mul_lo_u32 + v_add_u32; passing the 64-bit product y*w as the addend makes the whole lazy product three
    // instructions (v_mul_hi_u32, v_mad_u64_u32, v_mad_u64_u32) instead of four
    static __device__ __forceinline__ u32 mad_lo(u32 a, u32 b, u64 c) { return mad_lo32(a, b, c); }
    // arithmetic.rs:16-20: w*y - q*floor(y*w'/2^32), wrapping, in [0,2q)
    __device__ __forceinline__ u32 mul1(u32 y, Tw t) const { return mad_lo(__umulhi(y, t.wp), 0u - q, (u64)y * t.w); }
    // the same product NEGATED, q*floor(y*w'/2^32) - w*y (wrapping), from a twiddle stored as 2^32 - w: a forward
    // butterfly then is x' = tx - tn, y' = tx + 2q + tn (one v_sub_u32 and one v_add3_u32) — seven instructions per
    // coefficient butterfly where the compiler's form of arithmetic.rs:43-59 took ten
    __device__ __forceinline__ u32 mul1_neg(u32 y, Tw tn) const { return mad_lo(__umulhi(y, tn.wp), q, (u64)y * tn.w); }
    static __device__ __forceinline__ u32 once(u32 x, u32 m) { return min(x, x - m); }  // arithmetic.rs:3-6
    __device__ __forceinline__ u64 mul_lazy(u64 y, Tw t) const { return pack(mul1((u32)y, t), mul1((u32)(y >> 32), t)); }
    __device__ __forceinline__ u64 reduce_x(u64 x) const { return pack(once((u32)x, two_q32), once((u32)(x >> 32), two_q32)); }
    __device__ __forceinline__ u64 reduce_2q(u64 x) const { return pack(once((u32)x, q), once((u32)(x >> 32), q)); }
    __device__ __forceinline__ u64 reduce_4q(u64 x) const { return reduce_2q(reduce_x(x)); }

    // butterflies on both halves with 32-bit operations only (no 64-bit carries)
    __device__ __forceinline__ void fwd1(u32 &x, u32 &y, Tw wn) const {
        const u32 tx = once(x, two_q32), tn = mul1_neg(y, wn);
        x = tx - tn;
        y = tx + two_q32 + tn;
    }
    __device__ __forceinline__ void inv1(u32 &x, u32 &y, Tw w) const {
        const u32 tx = x + y, ty = x + two_q32 - y;
        x = once(tx, two_q32);
        y = mul1(ty, w);
    }
    __device__ __forceinline__ void fwd_bfly(u64 &x, u64 &y, Tw w) const {
        u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
        fwd1(x0, y0, w);
        fwd1(x1, y1, w);
        x = pack(x0, x1);
        y = pack(y0, y1);
    }
    __device__ __forceinline__ void inv_bfly(u64 &x, u64 &y, Tw w) const {
        u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
        inv1(x0, y0, w);
        inv1(x1, y1, w);
        x = pack(x0, x1);
        y = pack(y0, y1);
    }
    __device__ __forceinline__ void final1(u32 &x, u32 &y, bool lazy) const {  // scalar/transform.rs:253-271
        const u32 tx = once(x + y, two_q32), ty = x + two_q32 - y;
        u32 rx = mul1(tx, inv_n), ry = mul1(ty, inv_n_w);
        if (!lazy) {
            rx = once(rx, q);
            ry = once(ry, q);
        }
        x = rx;
        y = ry;
    }
    __device__ __forceinline__ void inv_final_bfly(u64 &x, u64 &y, bool lazy) const {
        u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
        final1(x0, y0, lazy);
        final1(x1, y1, lazy);
        x = pack(x0, x1);
        y = pack(y0, y1);
    }

    // distance-1 stage, forward: word at word index i uses roots[N/2 + i] (n = N/2 words)
    __device__ __forceinline__ u64 fwd_intra(u64 x, u32 n_plus_i) const { return fwd_intra_tw(x, fwd_tw(n_plus_i)); }
    __device__ __forceinline__ u64 fwd_intra_tw(u64 x, Tw wn) const {
        const u32 tx = once((u32)x, two_q32), tn = mul1_neg((u32)(x >> 32), wn);
        return pack(tx - tn, tx + two_q32 + tn);
    }
    // distance-1 stage, inverse: inv_roots[1 + i]; `inv` is biased by n words
    __device__ __forceinline__ u64 inv_intra(u64 x, u32 n, u32 i) const {
        return inv_intra_tw(x, unpack(inv[(long)(1 + i) - (long)n]));
    }
    __device__ __forceinline__ u64 inv_intra_tw(u64 x, Tw w) const {
        const u32 a = (u32)x, b = (u32)(x >> 32);
        return pack(once(a + b, two_q32), mul1(a + two_q32 - b, w));
    }
};

// Harvey forward butterfly, values in [0,4q) — scalar/arithmetic.rs:43-59.
//
// Wide policies (PmArith), in units of U = 2^K (K <= 61, so 8U <= 2^64): a twiddle product is T < 3U and T <= 3q
// for ANY 64-bit multiplicand, and a folded value is X < U + 2^31.  A butterfly that folds x first gives
// x' = X + T < 4U + 2^31 and y' = X + 3q - T < 4U + 2^31; one that does NOT fold an input below 4U + 2^31 gives
// x' < 7U + 2^31 and y' < 7U + 2^31, which still fit 64 bits and are valid inputs of a folding butterfly or of the
// twiddle product.  So every OTHER stage may skip the fold (FOLD = false); two skipping stages must never
// follow each other.  Inputs of a transform are below 4q < 4U by the reference's contract, so its first stage may
// skip as well.  The callers' pattern: a block pass folds at odd distances (2^p, p odd), a strided pass at its
// even register bits (so its last stage folds, whatever follows).
// FIRST (Montgomery tables): the first stage of a transform, whose x needs no fold.
template <bool FOLD = true, bool UNI = false, bool FIRST = false, class A>
__device__ __forceinline__ void fwd_bfly(const A &ar, u64 &x, u64 &y, typename A::Tw w) {
    if constexpr (A::kPacked) {
        ar.fwd_bfly(x, y, w);
    } else if constexpr (A::kWide) {
        pm_fwd_bfly1<FOLD, UNI>(ar, x, y, w);
    } else if constexpr (A::kMont) {
        mont_fwd_bfly1<UNI, !FIRST>(ar, x, y, w);  // (folds x at every stage but the first of a transform: MontArith::fold)
    } else {
        const u64 tx = ar.reduce_x(x);
        const u64 t = ar.mul_lazy(y, w);
        x = tx + t;
        y = tx + ar.two_q - t;
    }
}

// Gentleman-Sande inverse butterfly, values in [0,2q) — scalar/arithmetic.rs:63-79.
// Wide policies: inputs below 3U; x' = fold(x + y) < U + 2^31, y' = (x + 3q - y) * w < 3U.
template <bool UNI = false, class A>
__device__ __forceinline__ void inv_bfly(const A &ar, u64 &x, u64 &y, typename A::Tw w) {
    if constexpr (A::kPacked) {
        ar.inv_bfly(x, y, w);
    } else if constexpr (A::kWide) {
        pm_inv_bfly1<UNI>(ar, x, y, w);
    } else if constexpr (A::kMont) {
        mont_inv_bfly1<UNI>(ar, x, y, w);  // inputs below F: x' < F, y' < 3q
    } else {
        const u64 tx = x + y;
        const u64 ty = x + ar.two_q - y;
        x = ar.reduce_x(tx);
        y = ar.mul_lazy(ty, w);
    }
}

// last inverse stage fused with N^-1 (x) and N^-1*w (y) — scalar/transform.rs:283-318
template <class A>
__device__ __forceinline__ void inv_final_bfly(const A &ar, u64 &x, u64 &y, bool lazy) {
    if constexpr (A::kPacked) {
        ar.inv_final_bfly(x, y, lazy);
    } else if constexpr (A::kWide) {
        const u64 tx = x + y;  // the twiddle product takes any 64-bit value: no fold in front of it
        const u64 ty = sub_u64(x + ar.q3, y);
        const u64 rx = ar.mul_lazy(tx, ar.tw_inv_n());
        const u64 ry = ar.mul_lazy(ty, ar.tw_inv_n_w());
        // lazy results must honour the reference's [0,2q) contract: one fold (< U + 2^31 < 2q)
        x = lazy ? ar.reduce_x(rx) : ar.canon(rx);
        y = lazy ? ar.reduce_x(ry) : ar.canon(ry);
    } else if constexpr (A::kMont) {
        const u64 tx = x + y;  // inputs below F <= 2^63; the product takes any 64-bit value
        const u64 ty = sub_u64(x + ar.qf, y);
        const u64 rx = ar.mul_lazy(tx, ar.tw_inv_n());    // [0, 3q)
        const u64 ry = ar.mul_lazy(ty, ar.tw_inv_n_w());
        // lazy: the reference's [0,2q) contract (one conditional subtraction of 2q); else canonical
        x = lazy ? csub(rx, ar.two_q) : ar.canon3(rx);
        y = lazy ? csub(ry, ar.two_q) : ar.canon3(ry);
    } else {
        const u64 tx = ar.reduce_x(x + y);
        const u64 ty = x + ar.two_q - y;
        u64 rx = ar.mul_lazy(tx, ar.tw_inv_n());
        u64 ry = ar.mul_lazy(ty, ar.tw_inv_n_w());
        if (!lazy) {
            rx = ar.reduce_2q(rx);
            ry = ar.reduce_2q(ry);
        }
        x = rx;
        y = ry;
    }
}

// end of a forward transform: the lazy result honours the reference's [0,4q) contract, the plain one is canonical
// (scalar/transform.rs:104-116)
template <class A>
__device__ __forceinline__ u64 fwd_finish(const A &ar, u64 x, bool lazy) {
    if constexpr (A::kMont) {  // below 2^63 + 3q -> [0,q) (lazy too: inside the reference's [0,4q))
        return lazy ? ar.fwd_lazy(x) : ar.canon(x);
    } else if constexpr (A::kWide) {
        return lazy ? ar.reduce_x(x) : ar.canon(x);
    } else {
        return lazy ? x : ar.reduce_4q(x);
    }
}

// K forward stages on 2^K register-resident coefficients at stride 2^log_s (element index of
// register 0 = ebase): the body of the forward strided pass, shared with the fused decomposition.
// FIRST: the pass holds the first stage of the transform (distance N/2), which reads the transform's inputs — below 4q by
// the reference's contract (scalar/transform.rs:9), so Montgomery tables leave the fold of x out there.
template <class A, int K, int VEC, bool FIRST = false>
__device__ __forceinline__ void strided_forward_regs(const A &ar, u64 (&x)[1 << K][VEC], u32 n, u32 ebase, u32 log_s) {
    constexpr int R = 1 << K;
#ifdef PFHE_SKELETON  // timing-only build (tools/build_variant.sh skel -DPFHE_SKELETON): the data movement without the butterflies
#pragma unroll
    for (int v = 0; v < (R >> 1); ++v)
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            x[v][c] += x[v | (R >> 1)][c];
            x[v | (R >> 1)][c] ^= x[v][c];
        }
    return;
#endif
    if constexpr (A::kMont && FIRST) {
        const typename A::Tw w = ar.fwd_tw((n + ebase) >> (log_s + K));
#pragma unroll
        for (int v = 0; v < (R >> 1); ++v)
#pragma unroll
            for (int c = 0; c < VEC; ++c) fwd_bfly<true, true, true>(ar, x[v][c], x[v | (R >> 1)][c], w);
    }
#pragma unroll
    for (int j = (A::kMont && FIRST) ? K - 2 : K - 1; j >= 0; --j) {
        const u32 base = (n + ebase) >> (log_s + j + 1);
#pragma unroll
        for (int u = 0; u < (R >> (j + 1)); ++u) {
            const typename A::Tw w = ar.fwd_tw(base + u);
#pragma unroll
            for (int v = 0; v < (1 << j); ++v) {
                const int k0 = (u << (j + 1)) | v, k1 = k0 | (1 << j);
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    // (the twiddle index is wave-uniform in every strided kernel: scalar registers)
                    if (j & 1) fwd_bfly<false, true>(ar, x[k0][c], x[k1][c], w);
                    else fwd_bfly<true, true>(ar, x[k0][c], x[k1][c], w);
                }
            }
        }
    }
}

// inverse counterpart: K stages at distances 2^log_s ... 2^(log_s+K-1); FINAL fuses the N^-1 scaling
// of the last stage of the whole transform (scalar/transform.rs:283-318).
template <class A, int K, int VEC, bool FINAL>
__device__ __forceinline__ void strided_inverse_regs(const A &ar, u64 (&x)[1 << K][VEC], u32 n, u32 ebase, u32 log_s,
                                                     bool lazy) {
    constexpr int R = 1 << K;
    constexpr int JTOP = FINAL ? K - 1 : K;
#pragma unroll
    for (int j = 0; j < JTOP; ++j) {
        const u32 p = log_s + j;
        const u32 base = 1 + n - (n >> p) + (ebase >> (p + 1));
#pragma unroll
        for (int u = 0; u < (R >> (j + 1)); ++u) {
            const typename A::Tw w = ar.inv_tw(base + u);
#pragma unroll
            for (int v = 0; v < (1 << j); ++v) {
                const int k0 = (u << (j + 1)) | v, k1 = k0 | (1 << j);
#pragma unroll
                for (int c = 0; c < VEC; ++c) inv_bfly<true>(ar, x[k0][c], x[k1][c], w);
            }
        }
    }
    if constexpr (FINAL) {
#pragma unroll
        for (int v = 0; v < R / 2; ++v)
#pragma unroll
            for (int c = 0; c < VEC; ++c) inv_final_bfly(ar, x[v][c], x[v + R / 2][c], lazy);
    }
}

// A block pass owns blocks of B = 2^LOGB coefficients; every thread keeps E = 2^LOGE of them in registers and runs LOGE
// radix-2 stages per register pass.  LOGE = 4 (16 coefficients, 4 stages) is the general form; LOGE = 3 halves the
// registers a thread needs and doubles the threads per block: more resident waves per SIMD to hide the LDS / twiddle /
// barrier latencies behind, at the price of one more exchange per 12 stages.
template <int LOGB, int LOGE = 4>
struct BlockCfg {
    static_assert(LOGB >= 4 && LOGB <= 14, "block pass handles 2^4 .. 2^14 coefficients");
    static_assert(LOGE == 3 || LOGE == 4, "8 or 16 coefficients per thread");
    static constexpr int B = 1 << LOGB;
    static constexpr int E = 1 << LOGE;                    // coefficients per thread
    static constexpr int TPB = B >> LOGE;                  // threads per block of coefficients
    static_assert(TPB <= 1024, "a block must fit one workgroup");
// smallest workgroup: one wave.  Small workgroups put more independent workgroups on a CU (LDS is
// what limits residency), which overlaps their load / compute / store phases better: measured
// +2..9 % for N = 2^8..2^11 against 256-thread workgroups (measured with a 256-thread minimum).
    static constexpr int kMinWg = 64;
    static constexpr int THREADS = TPB > kMinWg ? TPB : kMinWg;  // workgroup size
    static constexpr int BPW = THREADS / TPB;              // coefficient blocks per workgroup
    static constexpr int LDS_WORDS = B + B / 8;            // 16 words + 2 words of padding
};

// padded LDS index of block-local element e: every 16 words are followed by 2 pad words, which
// keeps 16-byte alignment and de-phases the 128-byte rows read by the pos = 0 register pass.
__device__ __forceinline__ u32 lds_phi(u32 e) { return e + ((e >> 4) << 1); }

// block-local element index of register k of thread lt when register bits sit at [POS+LOGE-1 : POS]
template <int POS, int LOGE = 4>
__device__ __forceinline__ u32 layout(u32 lt, int k) {
    if constexpr (POS == 0) {
        return (lt << LOGE) | (u32)k;
    } else {
        return ((lt >> POS) << (POS + LOGE)) | ((u32)k << POS) | (lt & ((1u << POS) - 1));
    }
}

template <int POS, bool UNIFORM>
__device__ __forceinline__ u32 maybe_uniform(u32 v) {
    if constexpr (UNIFORM && POS >= 6) {
        return __builtin_amdgcn_readfirstlane(v);  // lt >> POS is constant across a wave
    } else {
        return v;
    }
}

// Entry of the lane-ordered tables (NttPrime::fwd_last / inv_last, built for groups of 16 coefficients) for the stage at
// distance 2^j, twiddle u of the thread whose first coefficient is element e0 (a multiple of 2^LOGE): the group of 16 is
// g16 = e0 >> 4, and a thread of 8 coefficients owns half of its twiddles (h = bit 3 of e0 selects which).
template <int LOGE>
__device__ __forceinline__ u32 last_table_off(u32 n, u32 e0, int j, int u) {
    constexpr int S = 4 - LOGE;
    const u32 g = e0 >> LOGE, h = g & ((1u << S) - 1), g16 = g >> S;
    const u32 uu = (h << (LOGE - 1 - j)) + (u32)u;
    return (((8u >> j) - 1) + uu) * (n >> 4) + g16;
}

// forward stages on register bits JHI..JLO (element bits POS+JHI .. POS+JLO); twiddle of the
// butterfly at global element E, distance 2^p: fwd[(N + E) >> (p + 1)]
// EVEN: wide policies fold the x input at EVEN distances 2^p instead of odd ones (see block_forward_core)
template <class A, int POS, int JHI, int JLO, bool UNIFORM, int LOGE = 4, bool EVEN = false>
__device__ __forceinline__ void fwd_regpass(const A &ar, u64 (&x)[1 << LOGE], u32 n_plus_e, u32 n) {
    constexpr int E = 1 << LOGE;
#ifdef PFHE_SKELETON  // timing-only build: see strided_forward_regs
#pragma unroll
    for (int v = 0; v < E / 2; ++v) {
        x[v] += x[v | (E / 2)];
        x[v | (E / 2)] ^= x[v];
    }
    return;
#endif
#pragma unroll
    for (int j = JHI; j >= JLO; --j) {
        const u32 base = maybe_uniform<POS, UNIFORM>(n_plus_e >> (POS + j + 1));
        constexpr bool kUni = UNIFORM && POS >= 6;  // maybe_uniform: the twiddle sits in scalar registers
        typename A::Tw w[E / 2];
        // POS == 0: every lane has twiddles of its own; the lane-ordered tables make the wave's loads contiguous
        const auto load_tw = [&](int u) {
            if constexpr (POS == 0 && A::kLastTables) w[u] = ar.fwd_tw_last(last_table_off<LOGE>(n, n_plus_e - n, j, u));
            else w[u] = ar.fwd_tw(base + u);
        };
        const int ntw = E >> (j + 1);  // twiddles of this stage
        if constexpr (A::kWide || A::kMont) {
            // two butterflies per asm block (independent instruction streams interleaved): butterfly b of the stage
            // has u = b >> j, v = b & (2^j - 1).  Per-lane twiddles (four registers each) are loaded at most four at a
            // time: the eight of the last stage of a 16-coefficient thread at once cost 32 registers at the tightest
            // point of the kernel.
            const int group = (kUni || ntw <= 4) ? ntw : 4;
#pragma unroll
            for (int u0 = 0; u0 < ntw; u0 += group) {
#pragma unroll
                for (int u = u0; u < u0 + group; ++u) load_tw(u);
#pragma unroll
                for (int b = u0 << j; b < ((u0 + group) << j); b += 2) {
                    const int ua = b >> j, va = b & ((1 << j) - 1), ub = (b + 1) >> j, vb = (b + 1) & ((1 << j) - 1);
                    const int a0 = (ua << (j + 1)) | va, a1 = a0 | (1 << j), b0 = (ub << (j + 1)) | vb, b1 = b0 | (1 << j);
                    if constexpr (A::kMont) mont_fwd_bfly2<kUni>(ar, x[a0], x[a1], w[ua], x[b0], x[b1], w[ub]);
                    else if ((((POS + j) & 1) != 0) != EVEN) pm_fwd_bfly2<true, kUni>(ar, x[a0], x[a1], w[ua], x[b0], x[b1], w[ub]);
                    else pm_fwd_bfly2<false, kUni>(ar, x[a0], x[a1], w[ua], x[b0], x[b1], w[ub]);
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < ntw; ++u) load_tw(u);
#pragma unroll
        for (int u = 0; u < ntw; ++u) {
#pragma unroll
            for (int v = 0; v < (1 << j); ++v) {
                const int k0 = (u << (j + 1)) | v, k1 = k0 | (1 << j);
                if ((((POS + j) & 1) != 0) != EVEN) fwd_bfly<true, kUni>(ar, x[k0], x[k1], w[u]);  // distance 2^(POS+j)
                else fwd_bfly<false, kUni>(ar, x[k0], x[k1], w[u]);
            }
        }
    }
}

// inverse stages on register bits JLO..JHI: inv[1 + N - (N >> p) + (E >> (p + 1))]; when
// `final_stage` the top stage (j == JHI) is the last stage of the whole transform.
template <class A, int POS, int JLO, int JHI, bool UNIFORM, int LOGE = 4>
__device__ __forceinline__ void inv_regpass(const A &ar, u64 (&x)[1 << LOGE], u32 n, u32 e_abs, bool final_stage,
                                            bool lazy) {
    constexpr int E = 1 << LOGE;
#pragma unroll
    for (int j = JLO; j <= JHI; ++j) {
        const u32 p = POS + j;
        if (j == JHI && final_stage) {
#pragma unroll
            for (int v = 0; v < E / 2; ++v) inv_final_bfly(ar, x[v], x[v | (E / 2)], lazy);
        } else {
            const u32 base = maybe_uniform<POS, UNIFORM>(1 + n - (n >> p) + (e_abs >> (p + 1)));
            constexpr bool kUni = UNIFORM && POS >= 6;
            typename A::Tw w[E / 2];
            const int ntw = E >> (j + 1);
#pragma unroll
            for (int u = 0; u < ntw; ++u) {
                if constexpr (POS == 0 && A::kLastTables) w[u] = ar.inv_tw_last(last_table_off<LOGE>(n, e_abs, j, u));  // lane-ordered
                else w[u] = ar.inv_tw(base + u);
            }
            if constexpr (A::kWide || A::kMont) {  // two butterflies per asm block, as in fwd_regpass
#pragma unroll
                for (int b = 0; b < E / 2; b += 2) {
                    const int ua = b >> j, va = b & ((1 << j) - 1), ub = (b + 1) >> j, vb = (b + 1) & ((1 << j) - 1);
                    const int a0 = (ua << (j + 1)) | va, a1 = a0 | (1 << j), b0 = (ub << (j + 1)) | vb, b1 = b0 | (1 << j);
                    if constexpr (A::kMont) mont_inv_bfly2<kUni>(ar, x[a0], x[a1], w[ua], x[b0], x[b1], w[ub]);
                    else pm_inv_bfly2<kUni>(ar, x[a0], x[a1], w[ua], x[b0], x[b1], w[ub]);
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < ntw; ++u) {
#pragma unroll
                for (int v = 0; v < (1 << j); ++v) {
                    const int k0 = (u << (j + 1)) | v, k1 = k0 | (1 << j);
                    inv_bfly<kUni>(ar, x[k0], x[k1], w[u]);
                }
            }
        }
    }
}

// Padded LDS index of register k in layout POS = index of register 0 + a compile-time constant: the bit fields of
// layout<POS>(lt, k) are disjoint, so lds_phi(layout(lt, k)) = lds_phi(layout(lt, 0)) + (k << POS) + 2 * ((k << POS) >> 4).
// One address register per layout and immediate offsets in the ds instructions, instead of sixteen computed addresses
// (which the compiler hoisted and, under register pressure, spilled).
template <int POS>
__host__ __device__ constexpr u32 lds_koff(int k) {
    return ((u32)k << POS) + 2u * (((u32)k << POS) >> 4);
}

template <int POS, int LOGE = 4>
__device__ __forceinline__ void lds_get_layout(u64 (&x)[1 << LOGE], const u64 *__restrict__ lds, u32 lt) {
    const u64 *__restrict__ base = lds + lds_phi(layout<POS, LOGE>(lt, 0));
#pragma unroll
    for (int k = 0; k < (1 << LOGE); ++k) x[k] = base[lds_koff<POS>(k)];
}

template <int POS, int LOGE = 4>
__device__ __forceinline__ void lds_put_layout(const u64 (&x)[1 << LOGE], u64 *__restrict__ lds, u32 lt) {
    u64 *__restrict__ base = lds + lds_phi(layout<POS, LOGE>(lt, 0));
#pragma unroll
    for (int k = 0; k < (1 << LOGE); ++k) base[lds_koff<POS>(k)] = x[k];
}

// registers (layout FROM) -> LDS -> registers (layout TO).
// FIRST: the exchange opens a chain, so other threads may still be reading the LDS region (staging of the caller):
// a workgroup barrier precedes the writes.  Later exchanges of a chain need none: a thread overwrites exactly the
// slots it read in the previous exchange (layout FROM is that exchange's layout TO).
// The threads that trade words in one exchange have ids inside one aligned block of 2^max(FROM, TO) threads; up to
// 2^6 that is a single wave, whose LDS accesses execute in program order: no workgroup barrier between the
// writes and the reads either.
// (Round 4 measured the wave-local <4> <-> <0> exchange as an in-register DPP transposition instead: +222 vector
// instructions per thread for 32 LDS instructions fewer, block pass +4.4 % — profiles/r04_exchange_dpp.txt; the code is in
// the commit "Experiment: the <4> <-> <0> exchange of the block cores as a DPP transposition".)
template <int FROM, int TO, bool FIRST, int LOGE = 4>
__device__ __forceinline__ void lds_exchange(u64 (&x)[1 << LOGE], u64 *__restrict__ lds, u32 lt) {
    constexpr bool kLead = FIRST, kWaveLocal = (FROM > TO ? FROM : TO) <= 6;
    if constexpr (kLead) __syncthreads();
    lds_put_layout<FROM, LOGE>(x, lds, lt);
    if constexpr (kWaveLocal) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
    lds_get_layout<TO, LOGE>(x, lds, lt);
}

// Stores of the pipelined kernels' INTERMEDIATE (strided pass -> block pass of the next launch; inverse: the other way
// round): non-temporal, or plain so that tiles small enough stay in the 256 MiB Infinity Cache until they are read.
constexpr bool kPipeIntermediateNt = false;

struct NoLateHook {
    __device__ __forceinline__ void operator()() const {}
};
// before_last runs in front of the last register pass (the pipelined kernel may issue its strided chunk's loads there)
template <class A, int LOGB, int POS, bool FIRST = false, int LOGE = 4, class Late = NoLateHook, bool EVEN = false>
__device__ __forceinline__ void fwd_chain(const A &ar, u64 (&x)[1 << LOGE], u64 *__restrict__ lds, u32 n, u32 eblk, u32 lt,
                                          Late before_last = Late()) {
    constexpr bool UNI = BlockCfg<LOGB, LOGE>::TPB >= 64;  // a wave never straddles two blocks
    if constexpr (POS > 0) {
        constexpr int NPOS = POS >= LOGE ? POS - LOGE : 0;
        constexpr int JHI = POS >= LOGE ? LOGE - 1 : POS - 1;
        asm volatile("" : "+v"(lt));  // this pass's LDS and twiddle addresses are computed here, not hoisted to the kernel's top
        lds_exchange<POS, NPOS, FIRST, LOGE>(x, lds, lt);
        if constexpr (NPOS == 0) before_last();
        fwd_regpass<A, NPOS, JHI, 0, UNI, LOGE, EVEN>(ar, x, n + eblk + layout<NPOS, LOGE>(lt, 0), n);
        fwd_chain<A, LOGB, NPOS, false, LOGE, Late, EVEN>(ar, x, lds, n, eblk, lt, before_last);
    }
}

// forward compute core: x holds layout<LOGB-LOGE> on entry and layout<0> (canonical unless lazy) on exit
// LEAD = false: the caller filled x by lds_get_layout<LOGB-LOGE> from this LDS region (so the first exchange, too,
// only overwrites slots its own thread read last) and needs no barrier in front of it.
// RAW (only with lazy): the output feeds a product with a second data word on chip (PmArith::mul_full, Barrett) instead of
// leaving the kernel, so the reference's [0,4q) contract of lazy outputs does not apply.  Wide policies then fold at EVEN
// distances — the last stage, distance 1, folds its x input — and return x' = X + T, y' = X + 3q - T < 4U + 2^31 as they
// are: below 2^63 + 2^32, which mul_full takes (its partial sums need y1 <= 2^31), and the finishing fold of every
// coefficient (4 instructions each) is gone.  The stage in front of the core must have folded (inputs below 4U + 2^31):
// true for canonical / [0,4q) inputs and behind a strided pass, whose last stage folds.  Montgomery tables skip their
// closing reduction the same way (values below 2^63 + 3q; the Barrett product takes any 64-bit word).
template <class A, int LOGB, bool LEAD = true, int LOGE = 4, class Late = NoLateHook, bool RAW = false>
__device__ __forceinline__ void block_forward_core(const A &ar, u64 (&x)[1 << LOGE], u64 *__restrict__ lds, u32 n,
                                                   u32 eblk, u32 lt, bool lazy, Late before_last = Late()) {
    constexpr int POS0 = LOGB - LOGE, E = 1 << LOGE;
    constexpr bool UNI = BlockCfg<LOGB, LOGE>::TPB >= 64;  // a wave never straddles two blocks
    constexpr bool EVEN = RAW && A::kWide;
    fwd_regpass<A, POS0, LOGE - 1, 0, UNI, LOGE, EVEN>(ar, x, n + eblk + layout<POS0, LOGE>(lt, 0), n);
    fwd_chain<A, LOGB, POS0, LEAD, LOGE, Late, EVEN>(ar, x, lds, n, eblk, lt, before_last);
    if constexpr (RAW && (A::kWide || A::kMont)) return;
    if constexpr (A::kPacked) {  // distance-1 stage between the halves of each word
        if constexpr (LOGE == 4) {  // lane-ordered twiddles: slot 15 + k of group (eblk >> 4) + lt
            const u32 off0 = 15u * (n >> 4) + (eblk >> 4) + lt;
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = ar.fwd_intra_tw(x[k], ar.fwd_tw_last(off0 + (u32)k * (n >> 4)));
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = ar.fwd_intra(x[k], n + eblk + layout<0, LOGE>(lt, k));
        }
    }
    if constexpr (A::kWide) {
        if (!lazy) {  // one uniform branch for the whole thread, two elements per asm block
#pragma unroll
            for (int k = 0; k < E; k += 2) pm_canon2(ar, x[k], x[k + 1]);
        } else
        {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = fwd_finish(ar, x[k], lazy);
        }
    } else if constexpr (A::kMont) {  // below 2^63 + 3q -> [0,q), lazy or not (MontArith::fwd_lazy)
        ar.canon_regs(x);
    } else if (!lazy) {  // [0,4q) -> [0,q): scalar/transform.rs:104-116
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = ar.reduce_4q(x[k]);
    }
}

// before_last runs in front of the last register pass (uniform twiddles: the pass with the fewest live registers)
// HOOK_AT: the hook runs in front of the pass that has HOOK_AT more passes after it (0: the last pass)
template <class A, int LOGB, int POS, bool FIRST = false, int LOGE = 4, class Late = NoLateHook, int HOOK_AT = 0>
__device__ __forceinline__ void inv_chain(const A &ar, u64 (&x)[1 << LOGE], u64 *__restrict__ lds, u32 n, u32 eblk, u32 lt,
                                          bool final_block, bool lazy, Late before_last = Late()) {
    constexpr bool UNI = BlockCfg<LOGB, LOGE>::TPB >= 64;  // a wave never straddles two blocks
    constexpr int DONE = POS + LOGE;  // element bits already processed
    if constexpr (DONE < LOGB) {
        constexpr int NPOS = DONE <= LOGB - LOGE ? DONE : LOGB - LOGE;
        constexpr int JLO = DONE - NPOS;
        constexpr bool LAST = NPOS + LOGE >= LOGB;
        asm volatile("" : "+v"(lt));  // see fwd_chain
        lds_exchange<POS, NPOS, FIRST, LOGE>(x, lds, lt);
        // passes still to come after this one: ceil((LOGB - (NPOS + LOGE)) / LOGE)
        constexpr int AFTER = (LOGB - NPOS - LOGE + LOGE - 1) / LOGE;
        if constexpr (AFTER == HOOK_AT) before_last();
        inv_regpass<A, NPOS, JLO, LOGE - 1, UNI, LOGE>(ar, x, n, eblk + layout<NPOS, LOGE>(lt, 0), LAST && final_block, lazy);
        inv_chain<A, LOGB, NPOS, false, LOGE, Late, HOOK_AT>(ar, x, lds, n, eblk, lt, final_block, lazy, before_last);
    }
}

// inverse compute core: x holds layout<0> on entry and layout<LOGB-LOGE> on exit
// before_last: see inv_chain
template <class A, int LOGB, bool LEAD = true, int LOGE = 4, class Late = NoLateHook, int HOOK_AT = 0>
__device__ __forceinline__ void block_inverse_core(const A &ar, u64 (&x)[1 << LOGE], u64 *__restrict__ lds, u32 n,
                                                   u32 eblk, u32 lt, bool final_block, bool lazy, Late before_last = Late()) {
    constexpr bool UNI = BlockCfg<LOGB, LOGE>::TPB >= 64;  // a wave never straddles two blocks
    if constexpr (A::kPacked) {
        if constexpr (LOGE == 4) {  // lane-ordered twiddles, as in block_forward_core
            const u32 off0 = 15u * (n >> 4) + (eblk >> 4) + lt;
#pragma unroll
            for (int k = 0; k < (1 << LOGE); ++k) x[k] = ar.inv_intra_tw(x[k], ar.inv_tw_last(off0 + (u32)k * (n >> 4)));
        } else {
#pragma unroll
            for (int k = 0; k < (1 << LOGE); ++k) x[k] = ar.inv_intra(x[k], n, eblk + layout<0, LOGE>(lt, k));
        }
    }
    inv_regpass<A, 0, 0, LOGE - 1, UNI, LOGE>(ar, x, n, eblk + layout<0, LOGE>(lt, 0), LOGB == LOGE && final_block, lazy);
    inv_chain<A, LOGB, 0, LEAD, LOGE, Late, HOOK_AT>(ar, x, lds, n, eblk, lt, final_block, lazy, before_last);
}

// ---- coalesced block I/O: E/2 16-byte vectors per thread (vector v = elements 2v, 2v+1), one full KiB per wave
//      instruction, staged through LDS.  Which vectors a thread takes (round 5): WAVE-LOCAL when a block has whole waves —
//      wave w owns vectors [w * 64 * NV, (w + 1) * 64 * NV), exactly the elements its threads hold in layout <0> (thread lt
//      = elements lt * E ... lt * E + E - 1), lane l takes vectors l, l + 64, ... of that range.  The transposition between
//      layout <0> and the I/O vectors then never leaves the wave: no workgroup barrier between lds_put_layout<0> and
//      lds_get_vectors (forward output) or between lds_put_vectors and lds_get_layout<0> (inverse input) — one of the two
//      barriers of a block pass, whose cost is the skew between the workgroup's four waves, each on a SIMD of its own with
//      contention of its own (sync_vectors_layout0 below).  Smaller blocks keep the interleaved assignment lt + TPB * j.
template <int LOGB, int LOGE = 4>
__host__ __device__ constexpr bool wave_local_vectors() {
#ifdef PFHE_NO_WAVE_LOCAL_IO  // A/B build (tools/build_variant.sh): round 4's interleaved assignment and its workgroup barriers
    return false;
#else
    return BlockCfg<LOGB, LOGE>::TPB >= 64;
#endif
}
// WL: the wave-local assignment (default where it exists); a call site may keep the interleaved one (all of a kernel's
// calls must agree)
template <int LOGB, int LOGE = 4, bool WL = wave_local_vectors<LOGB, LOGE>()>
__device__ __forceinline__ u32 vec_index(u32 lt, int j) {
    if constexpr (WL) return ((lt >> 6) << (6 + LOGE - 1)) + (lt & 63u) + 64u * (u32)j;
    else return lt + (u32)BlockCfg<LOGB, LOGE>::TPB * (u32)j;
}
// between an LDS image written in layout <0> and read as I/O vectors, or the other way round
template <int LOGB, int LOGE = 4, bool WL = wave_local_vectors<LOGB, LOGE>()>
__device__ __forceinline__ void sync_vectors_layout0() {
    if constexpr (WL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int LOGB, int LOGE = 4, bool NT = false, bool WL = wave_local_vectors<LOGB, LOGE>()>
__device__ __forceinline__ void load_block_vectors(u64x2 (&v)[1 << (LOGE - 1)], const u64 *gptr, u32 lt) {
    const GCVec2Ptr p = (GCVec2Ptr)(const void *)gptr + vec_index<LOGB, LOGE, WL>(lt, 0);
    constexpr u32 kStep = WL ? 64u : (u32)BlockCfg<LOGB, LOGE>::TPB;
#pragma unroll
    for (int j = 0; j < (1 << (LOGE - 1)); ++j) {
        if constexpr (NT) {  // read-once data of a large batch
            v[j] = __builtin_nontemporal_load(p + kStep * j);
            continue;
        }
        v[j] = p[kStep * j];
    }
}

// Global store of transform output, non-temporal where NT: the kernels of LARGE batches (the pipelined transform, the
// external product's digit polynomials) write every word once and read it again only in a later launch, long after it
// has left the caches.  Measured against plain stores (-DPFHE_PLAIN_STORES), 12 288 limb-polynomials of 2^16: strided
// pass 2.09 vs 2.19 ms (6.2 vs 5.9 TB/s), forward transform 4.87 vs 5.06 ms, inverse 4.83 vs 4.99 ms, NTT -> mul -> INTT
// 10.06 vs 10.33 ms, external product 52.4 vs 51.9 k/s.  Small batches, whose intermediate the next pass finds in the
// Infinity Cache, keep plain stores (192 MiB: 0.181 ms plain, 0.190 ms non-temporal).
template <bool NT, class T>
__device__ __forceinline__ void gstore(T *p, T v) {
    if constexpr (NT) {
        __builtin_nontemporal_store(v, p);
        return;
    }
    *p = v;
}
// (a per-launch choice — `if (flag) non-temporal else plain` — does not survive the compiler: it merges the two stores
// into a plain one; the choice is a template parameter of the kernels that run large batches only)

template <int LOGB, int LOGE = 4, bool NT = false, bool WL = wave_local_vectors<LOGB, LOGE>()>
__device__ __forceinline__ void store_block_vectors(const u64x2 (&v)[1 << (LOGE - 1)], u64 *gptr, u32 lt) {
    const GVec2Ptr p = (GVec2Ptr)(void *)gptr + vec_index<LOGB, LOGE, WL>(lt, 0);
    constexpr u32 kStep = WL ? 64u : (u32)BlockCfg<LOGB, LOGE>::TPB;
#pragma unroll
    for (int j = 0; j < (1 << (LOGE - 1)); ++j) {
        if constexpr (NT) {
            __builtin_nontemporal_store(v[j], p + kStep * j);
            continue;
        }
        p[kStep * j] = v[j];
    }
}

// vector vec_index(lt, j) holds elements 2v, 2v+1: padded index = lds_phi(2 * vec_index(lt, 0)) + constant(j) once the
// element step 2 * kStep is a multiple of 16
template <int LOGB, int LOGE = 4, bool WL = wave_local_vectors<LOGB, LOGE>()>
__host__ __device__ constexpr u32 lds_voff(int j) {
    constexpr u32 kStep = WL ? 64u : (u32)BlockCfg<LOGB, LOGE>::TPB;
    return 2u * kStep * (u32)j + 2u * ((2u * kStep * (u32)j) >> 4);
}

template <int LOGB, int LOGE = 4, bool WL = wave_local_vectors<LOGB, LOGE>()>
__device__ __forceinline__ void lds_put_vectors(const u64x2 (&v)[1 << (LOGE - 1)], u64 *__restrict__ lds, u32 lt) {
    constexpr int NV = 1 << (LOGE - 1), TPB = BlockCfg<LOGB, LOGE>::TPB;
    if constexpr (TPB >= 8) {
        u64 *__restrict__ base = lds + lds_phi(2 * vec_index<LOGB, LOGE, WL>(lt, 0));
#pragma unroll
        for (int j = 0; j < NV; ++j) *reinterpret_cast<u64x2 *>(base + lds_voff<LOGB, LOGE, WL>(j)) = v[j];
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) *reinterpret_cast<u64x2 *>(lds + lds_phi(2 * (lt + TPB * j))) = v[j];
    }
}

template <int LOGB, int LOGE = 4, bool WL = wave_local_vectors<LOGB, LOGE>()>
__device__ __forceinline__ void lds_get_vectors(u64x2 (&v)[1 << (LOGE - 1)], const u64 *__restrict__ lds, u32 lt) {
    constexpr int NV = 1 << (LOGE - 1), TPB = BlockCfg<LOGB, LOGE>::TPB;
    if constexpr (TPB >= 8) {
        const u64 *__restrict__ base = lds + lds_phi(2 * vec_index<LOGB, LOGE, WL>(lt, 0));
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const u64x2 *>(base + lds_voff<LOGB, LOGE, WL>(j));
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const u64x2 *>(lds + lds_phi(2 * (lt + TPB * j)));
    }
}

#endif  // __HIPCC__

}  // namespace pfhe
