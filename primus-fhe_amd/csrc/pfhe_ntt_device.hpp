// pfhe_ntt_device.hpp — device building blocks of the NTT kernels (included by .hip files only).
#pragma once

#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"

namespace pfhe {

constexpr u32 kMaxSinglePassLog = 14;  // largest N handled by one block pass (128 KiB in LDS)
constexpr int kTwoPassBlockLog = 12;   // block size used under strided passes

struct NttPlan {
    bool tiny = false;
    int n_strided = 0;
    int strided[4] = {0, 0, 0, 0};  // stages per strided pass, in forward order
    int block_log = 0;
};
NttPlan make_ntt_plan(u32 log_n);

int ntt_forward_dev(const NttPrime *primes, u32 L, u32 log_n, u64 *data, u64 npolys, bool lazy,
                    hipStream_t s);
int ntt_inverse_dev(const NttPrime *primes, u32 L, u32 log_n, u64 *data, u64 npolys, bool lazy,
                    hipStream_t s);

int ntt_num_passes(u32 log_n);
void ntt_pass_name(u32 log_n, bool inverse, int index, char *buf, size_t cap);
int ntt_pass_dev(const NttPrime *primes, u32 L, u32 log_n, u64 *data, u64 npolys, bool inverse, int index,
                 bool lazy, hipStream_t s);

#if defined(__HIPCC__)

// Twiddle tables are reached through a pointer stored in NttPrime, which the compiler would treat
// as a generic (flat) pointer: flat loads tick both vmcnt and lgkmcnt and serialise against LDS
// traffic and prefetches.  Reading them through an explicit global-address-space pointer yields
// plain global_load_dwordx4.
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef const u64x2 __attribute__((address_space(1))) *TwPtr;
__device__ __forceinline__ TwPtr tw_global(const ulonglong2 *p) { return (TwPtr)(const void *)p; }

// Harvey forward butterfly, values in [0,4q) — scalar/arithmetic.rs:43-59
__device__ __forceinline__ void fwd_bfly(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 two_q) {
    const u64 tx = reduce_once(x, two_q);
    const u64 t = mul_shoup_lazy(y, w, wp, q);
    x = tx + t;
    y = tx + two_q - t;
}

// Gentleman-Sande inverse butterfly, values in [0,2q) — scalar/arithmetic.rs:63-79
__device__ __forceinline__ void inv_bfly(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 two_q) {
    const u64 tx = x + y;
    const u64 ty = x + two_q - y;
    x = reduce_once(tx, two_q);
    y = mul_shoup_lazy(ty, w, wp, q);
}

// last inverse stage fused with N^-1 (x) and N^-1*w (y) — scalar/transform.rs:283-318
template <bool LAZY>
__device__ __forceinline__ void inv_final_bfly(u64 &x, u64 &y, const NttPrime &P) {
    const u64 tx = reduce_once(x + y, P.two_q);
    const u64 ty = x + P.two_q - y;
    u64 rx = mul_shoup_lazy(tx, P.inv_n, P.inv_n_p, P.q);
    u64 ry = mul_shoup_lazy(ty, P.inv_n_w, P.inv_n_w_p, P.q);
    if (!LAZY) {
        rx = reduce_once(rx, P.q);
        ry = reduce_once(ry, P.q);
    }
    x = rx;
    y = ry;
}

template <int LOGB>
struct BlockCfg {
    static_assert(LOGB >= 4 && LOGB <= 14, "block pass handles 2^4 .. 2^14 coefficients");
    static constexpr int B = 1 << LOGB;
    static constexpr int TPB = B / 16;                      // threads per block of coefficients
    static constexpr int THREADS = TPB > 256 ? TPB : 256;   // workgroup size
    static constexpr int BPW = THREADS / TPB;               // coefficient blocks per workgroup
    static constexpr int LDS_WORDS = B + B / 8;             // 16 words + 2 words of padding
};

// padded LDS index of block-local element e: every 16 words are followed by 2 pad words, which
// keeps 16-byte alignment and de-phases the 128-byte rows read by the pos = 0 register pass.
__device__ __forceinline__ u32 lds_phi(u32 e) { return e + ((e >> 4) << 1); }

// block-local element index of register k of thread lt when register bits sit at [POS+3 : POS]
template <int POS>
__device__ __forceinline__ u32 layout(u32 lt, int k) {
    if constexpr (POS == 0) {
        return (lt << 4) | (u32)k;
    } else {
        return ((lt >> POS) << (POS + 4)) | ((u32)k << POS) | (lt & ((1u << POS) - 1));
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

// forward stages on register bits JHI..JLO (element bits POS+JHI .. POS+JLO)
template <int POS, int JHI, int JLO, bool UNIFORM>
__device__ __forceinline__ void fwd_regpass(u64 (&x)[16], TwPtr tw, u32 n_plus_e,
                                            u64 q, u64 two_q) {
#pragma unroll
    for (int j = JHI; j >= JLO; --j) {
        const u32 base = maybe_uniform<POS, UNIFORM>(n_plus_e >> (POS + j + 1));
#pragma unroll
        for (int u = 0; u < (16 >> (j + 1)); ++u) {
#if defined(PFHE_ABL_CONST_TW)
            const u64x2 w = u64x2{q - 12345 - u, two_q + base};
#else
            const u64x2 w = tw[base + u];
#endif
#if !defined(PFHE_ABL_NO_MATH)
#pragma unroll
            for (int v = 0; v < (1 << j); ++v) {
                const int k0 = (u << (j + 1)) | v, k1 = k0 | (1 << j);
                fwd_bfly(x[k0], x[k1], w.x, w.y, q, two_q);
            }
#else
            x[u] += w.x;
#endif
        }
    }
}

// inverse stages on register bits JLO..JHI; when `final_stage` the top stage (j == JHI) is the
// last stage of the whole transform.
template <int POS, int JLO, int JHI, bool UNIFORM, bool LAZY>
__device__ __forceinline__ void inv_regpass(u64 (&x)[16], const NttPrime *__restrict__ P, u32 n, u32 e_abs,
                                            u64 q, u64 two_q, bool final_stage) {
    const TwPtr tw = tw_global(P->inv);
#pragma unroll
    for (int j = JLO; j <= JHI; ++j) {
        const u32 p = POS + j;
        if (j == JHI && final_stage) {
            const NttPrime PP = *P;
#pragma unroll
            for (int v = 0; v < 8; ++v) inv_final_bfly<LAZY>(x[v], x[v | 8], PP);
        } else {
            const u32 base = maybe_uniform<POS, UNIFORM>(1 + n - (n >> p) + (e_abs >> (p + 1)));
#pragma unroll
            for (int u = 0; u < (16 >> (j + 1)); ++u) {
                const u64x2 w = tw[base + u];
#pragma unroll
                for (int v = 0; v < (1 << j); ++v) {
                    const int k0 = (u << (j + 1)) | v, k1 = k0 | (1 << j);
                    inv_bfly(x[k0], x[k1], w.x, w.y, q, two_q);
                }
            }
        }
    }
}

// registers (layout FROM) -> LDS -> registers (layout TO)
template <int FROM, int TO, bool SYNC_BEFORE>
__device__ __forceinline__ void lds_exchange(u64 (&x)[16], u64 *__restrict__ lds, u32 lt) {
#if defined(PFHE_ABL_NO_LDS)
    return;
#endif
    if constexpr (SYNC_BEFORE) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[lds_phi(layout<FROM>(lt, k))] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[lds_phi(layout<TO>(lt, k))];
}

template <int LOGB, int POS, bool FIRST>
__device__ __forceinline__ void fwd_chain(u64 (&x)[16], u64 *__restrict__ lds, TwPtr tw,
                                          u32 n, u32 eblk, u32 lt, u64 q, u64 two_q) {
    constexpr bool UNI = BlockCfg<LOGB>::BPW == 1;
    if constexpr (POS > 0) {
        constexpr int NPOS = POS >= 4 ? POS - 4 : 0;
        constexpr int JHI = POS >= 4 ? 3 : POS - 1;
        lds_exchange<POS, NPOS, !FIRST>(x, lds, lt);
        fwd_regpass<NPOS, JHI, 0, UNI>(x, tw, n + eblk + layout<NPOS>(lt, 0), q, two_q);
        fwd_chain<LOGB, NPOS, false>(x, lds, tw, n, eblk, lt, q, two_q);
    }
}

// forward compute core: x holds layout<LOGB-4> on entry and layout<0> (canonical unless LAZY) on exit.
// LDS_DIRTY: other threads may still be reading the LDS region (sync before the first write).
template <int LOGB, bool LAZY, bool LDS_DIRTY>
__device__ __forceinline__ void block_forward_core(u64 (&x)[16], u64 *__restrict__ lds,
                                                   const NttPrime *__restrict__ P, u32 n, u32 eblk, u32 lt) {
    constexpr int POS0 = LOGB - 4;
    constexpr bool UNI = BlockCfg<LOGB>::BPW == 1;
    const u64 q = P->q, two_q = P->two_q;
    const TwPtr tw = tw_global(P->fwd);
    fwd_regpass<POS0, 3, 0, UNI>(x, tw, n + eblk + layout<POS0>(lt, 0), q, two_q);
    fwd_chain<LOGB, POS0, !LDS_DIRTY>(x, lds, tw, n, eblk, lt, q, two_q);
    if (!LAZY) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = reduce_once(reduce_once(x[k], two_q), q);
    }
}

template <int LOGB, bool LAZY>
__device__ __forceinline__ void block_forward(u64 (&x)[16], u64 *__restrict__ gptr, u64 *__restrict__ lds,
                                              const NttPrime *__restrict__ P, u32 n, u32 eblk, u32 lt,
                                              bool valid) {
    constexpr int POS0 = LOGB - 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = valid ? gptr[layout<POS0>(lt, k)] : 0ull;
    block_forward_core<LOGB, LAZY, false>(x, lds, P, n, eblk, lt);
    if (valid) {
#pragma unroll
        for (int k = 0; k < 16; ++k) gptr[layout<0>(lt, k)] = x[k];
    }
}

template <int LOGB, int POS, bool FIRST, bool LAZY>
__device__ __forceinline__ void inv_chain(u64 (&x)[16], u64 *__restrict__ lds, const NttPrime *__restrict__ P,
                                          u32 n, u32 eblk, u32 lt, u64 q, u64 two_q, bool final_block) {
    constexpr bool UNI = BlockCfg<LOGB>::BPW == 1;
    constexpr int DONE = POS + 4;  // element bits already processed
    if constexpr (DONE < LOGB) {
        constexpr int NPOS = DONE <= LOGB - 4 ? DONE : LOGB - 4;
        constexpr int JLO = DONE - NPOS;
        constexpr bool LAST = NPOS + 4 >= LOGB;
        lds_exchange<POS, NPOS, !FIRST>(x, lds, lt);
        inv_regpass<NPOS, JLO, 3, UNI, LAZY>(x, P, n, eblk + layout<NPOS>(lt, 0), q, two_q,
                                             LAST && final_block);
        inv_chain<LOGB, NPOS, false, LAZY>(x, lds, P, n, eblk, lt, q, two_q, final_block);
    }
}

// inverse compute core: x holds layout<0> on entry and layout<LOGB-4> on exit.
template <int LOGB, bool LAZY, bool LDS_DIRTY>
__device__ __forceinline__ void block_inverse_core(u64 (&x)[16], u64 *__restrict__ lds,
                                                   const NttPrime *__restrict__ P, u32 n, u32 eblk, u32 lt,
                                                   bool final_block) {
    constexpr bool UNI = BlockCfg<LOGB>::BPW == 1;
    const u64 q = P->q, two_q = P->two_q;
    inv_regpass<0, 0, 3, UNI, LAZY>(x, P, n, eblk + layout<0>(lt, 0), q, two_q, LOGB == 4 && final_block);
    inv_chain<LOGB, 0, !LDS_DIRTY, LAZY>(x, lds, P, n, eblk, lt, q, two_q, final_block);
}

template <int LOGB, bool LAZY>
__device__ __forceinline__ void block_inverse(u64 (&x)[16], u64 *__restrict__ gptr, u64 *__restrict__ lds,
                                              const NttPrime *__restrict__ P, u32 n, u32 eblk, u32 lt,
                                              bool valid, bool final_block) {
    constexpr int POSL = LOGB - 4;  // layout of the last register pass
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = valid ? gptr[layout<0>(lt, k)] : 0ull;
    block_inverse_core<LOGB, LAZY, false>(x, lds, P, n, eblk, lt, final_block);
    if (valid) {
#pragma unroll
        for (int k = 0; k < 16; ++k) gptr[layout<POSL>(lt, k)] = x[k];
    }
}

// ---- coalesced block I/O through LDS (persistent kernel): 8 x 16-byte vectors per thread in
//      natural order (vector v = elements 2v, 2v+1), one full KiB per wave instruction ----
typedef u64x2 __attribute__((address_space(1))) *GVecPtr;

template <int LOGB>
__device__ __forceinline__ void load_block_vectors(u64x2 (&v)[8], const u64 *gptr, u32 lt) {
    const TwPtr p = (TwPtr)(const void *)gptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[lt + BlockCfg<LOGB>::TPB * j];
}

template <int LOGB>
__device__ __forceinline__ void store_block_vectors(const u64x2 (&v)[8], u64 *gptr, u32 lt) {
    const GVecPtr p = (GVecPtr)(void *)gptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[lt + BlockCfg<LOGB>::TPB * j] = v[j];
}

template <int LOGB>
__device__ __forceinline__ void lds_put_vectors(const u64x2 (&v)[8], u64 *__restrict__ lds, u32 lt) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u64x2 *>(lds + lds_phi(2 * (lt + BlockCfg<LOGB>::TPB * j))) = v[j];
}

template <int LOGB>
__device__ __forceinline__ void lds_get_vectors(u64x2 (&v)[8], const u64 *__restrict__ lds, u32 lt) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u64x2 *>(lds + lds_phi(2 * (lt + BlockCfg<LOGB>::TPB * j)));
}

template <int POS>
__device__ __forceinline__ void lds_get_layout(u64 (&x)[16], const u64 *__restrict__ lds, u32 lt) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[lds_phi(layout<POS>(lt, k))];
}

template <int POS>
__device__ __forceinline__ void lds_put_layout(const u64 (&x)[16], u64 *__restrict__ lds, u32 lt) {
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[lds_phi(layout<POS>(lt, k))] = x[k];
}

#endif  // __HIPCC__

}  // namespace pfhe
