// pfhe_ntt.hip — batched negacyclic NTT / INTT kernels for gfx950 (MI355X, CDNA4).
//
// What is computed (bit-exact, canonical outputs): U64NttTable::transform_slice /
// inverse_transform_slice and their lazy variants (primus_ntt/src/ntt/prime64/table.rs:541-563,
// scalar/transform.rs:13-319) for every N-word chunk of a batch.  Index conventions are the
// reference's: forward stage with m groups uses roots[m + g]; inverse stage with m groups uses
// inv_roots[1 + N - 2m + g]; output of the forward transform is in bit-reversed order.
//
// How it is computed is NOT the reference's loop nest.  A transform is cut into passes:
//   * "strided" passes: 2^K coefficients at stride S per thread, K stages entirely in
//     registers, fully coalesced 16-byte loads/stores, twiddles wave-uniform (scalar loads);
//   * one "block" pass: a workgroup owns a contiguous block of 2^LOGB coefficients, every
//     thread keeps 16 of them in registers, runs 4 radix-2 stages per register pass and
//     re-shuffles through (padded) LDS between register passes; global traffic is 16-byte
//     coalesced and staged through LDS into/out of the register layouts.
// N <= 2^14 is a single block pass (one HBM read + one HBM write); N = 2^15..2^17 is one strided
// pass + one block pass of 2^12.  In terms of a global element index E (bit p is the butterfly
// distance 2^p) the twiddle index of a forward butterfly is (N + E) >> (p + 1) and of an inverse
// butterfly 1 + N - (N >> p) + (E >> (p + 1)).
//
// Every kernel is instantiated for three arithmetic policies (pfhe_ntt_device.hpp): ShoupArith for
// arbitrary q < 2^62, PmArith for pseudo-Mersenne primes q = 2^K - c, and B32Arith for the u32
// tables (q < 2^30), where a 64-bit word carries two adjacent u32 coefficients.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_ntt_device.hpp"

namespace pfhe {

// ------------------------------------------------------------------------------------------
// tiny transforms (N <= 8): one thread per polynomial, straight loops.  Only there so that the
// whole NttTable domain (log_n >= 0) is served by the device path.
// ------------------------------------------------------------------------------------------
template <bool INV>
__global__ void ntt_tiny_kernel(u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                u64 npolys, u32 lazy) {
    u64 pid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= npolys) return;
    const ShoupArith ar(primes + pid % L);
    const u32 n = 1u << log_n;
    u64 *x = data + pid * n;
    if (!INV) {
        for (u32 p = log_n; p-- > 0;) {
            for (u32 e = 0; e < n; ++e) {
                if (e & (1u << p)) continue;
                fwd_bfly(ar, x[e], x[e | (1u << p)], ar.fwd_tw((n + e) >> (p + 1)));
            }
        }
        if (!lazy && log_n > 0)
            for (u32 e = 0; e < n; ++e) x[e] = ar.reduce_4q(x[e]);
    } else {
        for (u32 p = 0; p + 1 < log_n; ++p) {
            for (u32 e = 0; e < n; ++e) {
                if (e & (1u << p)) continue;
                inv_bfly(ar, x[e], x[e | (1u << p)], ar.inv_tw(1 + n - (n >> p) + (e >> (p + 1))));
            }
        }
        if (log_n > 0) {
            const u32 h = n >> 1;
            for (u32 e = 0; e < h; ++e) inv_final_bfly(ar, x[e], x[e + h], lazy != 0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// strided pass: K stages on 2^K coefficients at stride S = 2^LOGS, VEC adjacent columns per
// thread (16-byte accesses when VEC == 2).
//   forward: stages with butterfly distance S*2^(K-1) ... S       (register bit j <-> bit LOGS+j)
//   inverse: stages with butterfly distance S ... S*2^(K-1); FINAL marks that the top stage is
//            the last stage of the whole transform (fused N^-1 scaling, table.rs:283-318).
// ------------------------------------------------------------------------------------------
// `src`: where the pass reads (the in-place kernels pass `data` itself, which the compiler sees as the same pointer; the
// out-of-place kernels of the host-pointer path read pinned host memory and write device memory, or the reverse).
template <class A, int K, int VEC, bool INV, bool FINAL, bool NT = false>
__device__ __forceinline__ void strided_pass_body(u64 *__restrict__ data, const u64 *src, const NttPrime *__restrict__ primes, u32 L,
                                                  u32 log_n, u32 log_s, u64 gid, u64 total_threads, u32 lazy) {
    constexpr int R = 1 << K;
    if (gid >= total_threads) return;
    // gid -> (pid, hi, col): col indexes VEC-wide column groups inside the stride
    const u32 log_cols = log_s - (VEC == 2 ? 1 : 0);
    const u32 log_hi = log_n - log_s - K;  // number of index bits above the register bits
    const u32 col = (u32)(gid & ((1ull << log_cols) - 1));
    // hi and pid are identical for all 64 lanes of a wave (2^log_cols >= 64 threads per value)
    const u32 hi = __builtin_amdgcn_readfirstlane((u32)((gid >> log_cols) & ((1ull << log_hi) - 1)));
    const u32 pid_lo = __builtin_amdgcn_readfirstlane((u32)(gid >> (log_cols + log_hi)));
    const A ar(primes + pid_lo % L);
    const u32 n = 1u << log_n;
    const u32 ebase = hi << (log_s + K);  // element index of register 0, column 0
    const u64 word0 = (u64)pid_lo * n + ebase + (u64)col * VEC;
    const u64 *sptr = src + word0;

    u64 x[R][VEC];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        // streaming (non-temporal) loads: this pass reads every word exactly once; 2.27 -> 2.16 ms per launch
        if constexpr (VEC == 2) {
            const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(sptr + ((u64)k << log_s)));
            x[k][0] = v.x;
            x[k][1] = v.y;
        } else {
            x[k][0] = __builtin_nontemporal_load(sptr + ((u64)k << log_s));
        }
    }

    if constexpr (!INV) {
        strided_forward_regs<A, K, VEC, FINAL>(ar, x, n, ebase, log_s);  // (forward kernels: FINAL = first pass of the transform)
    } else {
        strided_inverse_regs<A, K, VEC, FINAL>(ar, x, n, ebase, log_s, lazy != 0);
    }

    // The store addresses are recomputed from an opaque copy of the column index: derived from `ptr` they are the load
    // addresses, which the compiler keeps alive through all K stages (2^K 64-bit pairs: 32 of the two-column kernel's
    // 134-136 registers, three waves per SIMD).  Recomputed: 100-106 registers, four waves, and the pass streams faster —
    // forward K = 4: 2.10 -> 2.005 ms per 12.9 GB (6.13 -> 6.43 TB/s), inverse 2.18 -> 2.12 ms; inverse K = 5: 2.22 -> 2.04 ms.
    // The forward one-column K = 5 kernel loses (2.10 -> 2.25 ms) and keeps its addresses.
    constexpr bool kRecompute = VEC == 2 || INV;
    u32 col2 = col;
    if constexpr (kRecompute) asm volatile("" : "+v"(col2));
    u64 *__restrict__ optr = data + (u64)pid_lo * n + ebase + (u64)col2 * VEC;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if constexpr (VEC == 2) {
            gstore<NT>(reinterpret_cast<u64x2 *>(optr + ((u64)k << log_s)), u64x2{x[k][0], x[k][1]});
        } else {
            gstore<NT>(optr + ((u64)k << log_s), x[k][0]);
        }
    }
}

// Register budget: left to the compiler (134-136 registers for the two-column form, three waves per SIMD).  Capping
// it at 128 so that a wave fits beside three block-pass waves costs 4-16 spilled registers, i.e. scratch traffic on an
// HBM-bound kernel (6.34 / 7.5 GiB moved per 6 GiB launch, forward / inverse): 2.17 -> 2.22 ms and 2.18 -> 2.31 ms, for
// no better overlap.  VEC == 1 is the one-column form the five-stage pass takes.
constexpr int kStridedMinWaves = 1;
// NT: non-temporal stores (launch_strided picks it for batches of at least kNtMinBytes)
template <class A, int K, int VEC, bool INV, bool FINAL, bool NT = false>
__global__ __launch_bounds__(256, K <= 4 ? (VEC == 1 ? 5 : kStridedMinWaves) : 1) void ntt_strided_kernel(u64 *__restrict__ data,
                                                          const NttPrime *__restrict__ primes, u32 L,
                                                          u32 log_n, u32 log_s, u64 total_threads, u32 lazy) {
    strided_pass_body<A, K, VEC, INV, FINAL, NT>(data, data, primes, L, log_n, log_s,
                                                 (u64)blockIdx.x * blockDim.x + threadIdx.x, total_threads, lazy);
}
// out-of-place form (host-pointer path, ntt_transform_through_dev): plain stores
template <class A, int K, int VEC, bool INV, bool FINAL>
__global__ __launch_bounds__(256) void ntt_strided_oop_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src,
                                                              const NttPrime *__restrict__ primes, u32 L, u32 log_n, u32 log_s,
                                                              u64 total_threads, u32 lazy) {
    strided_pass_body<A, K, VEC, INV, FINAL, false>(dst, src, primes, L, log_n, log_s,
                                                    (u64)blockIdx.x * blockDim.x + threadIdx.x, total_threads, lazy);
}

// ------------------------------------------------------------------------------------------
// block pass: a workgroup owns BPW contiguous blocks of B = 2^LOGB coefficients (BPW > 1 only
// for B < 1024 so that a workgroup is at least one full wave).
//   forward: the LAST LOGB stages of the transform (distances B/2 ... 1), canonical reduction
//            fused into the final stage (scalar/transform.rs:104-116) unless lazy.
//   inverse: the FIRST LOGB stages (distances 1 ... B/2); when B == N the final stage carries
//            the fused N^-1 / N^-1*w scaling (scalar/transform.rs:283-318).
// ------------------------------------------------------------------------------------------
// LDS limits the block pass to four waves per SIMD; telling the compiler so makes it spend registers on
// instruction-level parallelism instead of chasing a higher occupancy it cannot get (1-3 % measured).
// (x is filled from the staging region by lds_get_layout, so the first exchange needs no barrier in front of it.)

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// after_stage runs once the block's own global loads have landed in LDS (the pipelined kernel issues the loads of its
// strided chunk there, so that they are in flight during the block's stages and do not delay the block's first wait)
// `src`: where the pass reads (`data` itself for the in-place kernels; see strided_pass_body)
// before_store runs in front of the block's global stores (the pipelined kernels touch their prefetched strided data there: see
// ntt_pipe_body)
template <class A, int LOGB, bool INV, bool MUL, int LOGE = 4, class Hook = NoHook, bool NTIO = false, class Hook2 = NoHook>
__device__ __forceinline__ void block_pass_body(u64 *__restrict__ data, const u64 *src, const NttPrime *__restrict__ primes, u32 L,
                                                u32 log_n, u64 total_blocks, u32 lazy, const u64 *__restrict__ mul,
                                                u64 mul_polys, u64 *__restrict__ lds_raw, u64 first_block,
                                                Hook after_stage = Hook(), Hook2 before_store = Hook2()) {
    using Cfg = BlockCfg<LOGB, LOGE>;
    // non-temporal stores and staged loads: the pipelined kernel (large batches only) and the NT instantiations that
    // launch_block picks for batches of at least kNtMinBytes
    constexpr bool kNt = !std::is_same<Hook, NoHook>::value || NTIO;
    constexpr int NV = Cfg::E / 2;  // 16-byte vectors per thread
    // I/O vectors wave-local (pfhe_ntt_device.hpp, vec_index) — except in the pipelined INVERSE kernel of Montgomery tables,
    // which sits at 128 registers and spills four with the wave-local addresses (inverse transform 5.19 against 5.07 ms)
    constexpr bool kWL = wave_local_vectors<LOGB, LOGE>() && !(INV && A::kMont && !std::is_same<Hook, NoHook>::value);
    const u32 tid = threadIdx.x;
    const u32 sub = Cfg::BPW == 1 ? 0u : tid / Cfg::TPB;
    const u32 lt = Cfg::BPW == 1 ? tid : tid % Cfg::TPB;
    u64 blk = first_block * Cfg::BPW + sub;
    const bool valid = blk < total_blocks;
    if (!valid) blk = 0;
    const u32 log_nb = log_n - LOGB;  // blocks per polynomial
    const u64 pid = blk >> log_nb;
    const u32 bi = (u32)(blk & ((1ull << log_nb) - 1));
    const u32 limb = (u32)(pid % L);
    const A ar(primes + limb);
    const u32 n = 1u << log_n;
    const u32 eblk = bi << LOGB;
    u64 *__restrict__ gptr = data + pid * n + eblk;
    const u64 *sgptr = src + pid * n + eblk;
    u64 *__restrict__ lds = lds_raw + (size_t)sub * Cfg::LDS_WORDS;

    // Forward direction: the first register pass wants register k of thread lt = element (k << POS0) + lt, which 8-byte
    // loads deliver directly (a wave instruction reads 512 contiguous bytes): no staging through LDS, one barrier and 24
    // LDS instructions fewer, and the first butterflies start when their own operands have landed.  Only inside the
    // pipelined kernel (Hook given), where it measures 5.15 against 5.19 ms per 12 288 transforms; stand-alone it is
    // neutral at 2^12 (3.22 ms either way) and slower for blocks of 2^14 (0.365 vs 0.357 ms per 4096) and for the u32
    // tables (3.10 vs 2.94 ms).  The mirror image, 8-byte stores after the last INVERSE pass, pays everywhere it was
    // measured (2^16 inverse 5.08 vs 5.20 ms, 2^14 inverse 0.381 vs 0.412 ms) and is the default below.
    constexpr bool kDirectLoad = !INV && !MUL && Cfg::BPW == 1 && (LOGB - LOGE) >= 6 &&
                                 !std::is_same<Hook, NoHook>::value && !A::kPacked
        ;
    u64x2 io[NV];
    u64 x[Cfg::E];
    if constexpr (kDirectLoad) {
#pragma unroll
        for (int k = 0; k < Cfg::E; ++k)
#ifdef PFHE_NOMEM  // timing-only build (tools/build_variant.sh nomem -DPFHE_NOMEM): the arithmetic and LDS traffic without global memory
            x[k] = (u64)lt * 0x9E3779B97F4A7C15ull + (u64)k + lazy;
#else
            x[k] = valid ? __builtin_nontemporal_load(sgptr + ((u32)k << (LOGB - LOGE)) + lt) : 0ull;
#endif
        // (forward: issuing them in front of the last register pass instead — the per-lane-twiddle one — costs 152 registers)
        after_stage();
        block_forward_core<A, LOGB, false, LOGE>(ar, x, lds, n, eblk, lt, lazy != 0);
        lds_put_layout<0, LOGE>(x, lds, lt);
        sync_vectors_layout0<LOGB, LOGE, kWL>();  // wave-local: no workgroup barrier (pfhe_ntt_device.hpp, vec_index)
        lds_get_vectors<LOGB, LOGE, kWL>(io, lds, lt);
        before_store();
#ifdef PFHE_NOMEM
        if (lazy == 0xdeadu)
#endif
        if (valid) store_block_vectors<LOGB, LOGE, kNt, kWL>(io, gptr, lt);
        return;
    }
    // all global traffic as 16-byte vectors in natural order (1 KiB per wave instruction), staged
    // through LDS into / out of the register layouts of the first / last register pass
    if (valid) {
        load_block_vectors<LOGB, LOGE, kNt, kWL>(io, sgptr, lt);
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) io[j] = u64x2{0, 0};
    }
    if constexpr (MUL) {  // fused pointwise product (DcrtPolynomial::mul_assign) on the way in
        // The multiplicand is one RNS polynomial (mul_polys == L, indexed by the limb) or one polynomial per data
        // polynomial (mul_polys == polynomials of this launch, indexed by pid).  The multiplies sit OUTSIDE the
        // `valid` branch on purpose: inside it, the zero high half of the zero-extended 32-bit constant c was
        // defined in the branch, reached the rest of the kernel as a phi the compiler could not fold, and every
        // multiplication by c in the transform became two (5.3 instead of 4.2 ms per 12 288 transforms).
        u64x2 mv[NV];
        if (valid) {
            const u64 mpoly = mul_polys == (u64)L ? (u64)limb : pid;
            load_block_vectors<LOGB, LOGE, kNt, kWL>(mv, mul + mpoly * n + eblk, lt);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) mv[j] = u64x2{0, 0};
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            io[j].x = ar.mul_any(io[j].x, mv[j].x);
            io[j].y = ar.mul_any(io[j].y, mv[j].y);
        }
    }
    lds_put_vectors<LOGB, LOGE, kWL>(io, lds, lt);
    // inverse: layout <0> reads the vectors of the thread's own wave; forward: layout <LOGB - LOGE> reads every wave's
    if constexpr (INV) sync_vectors_layout0<LOGB, LOGE, kWL>();
    else __syncthreads();
    // (inverse: the hook runs in front of the LAST register pass instead — uniform twiddles, the fewest live registers —
    // so that the pipelined kernel's 32 registers of prefetched strided data do not sit through the per-lane-twiddle
    // passes: 128 registers without spills, four waves per SIMD; 4.98 against 5.05 ms per 12 288 inverse transforms,
    // 10.33 against 10.64 ms for NTT -> mul -> INTT)
    if constexpr (!INV) after_stage();
    if constexpr (!INV) {
        lds_get_layout<LOGB - LOGE, LOGE>(x, lds, lt);
        block_forward_core<A, LOGB, false, LOGE>(ar, x, lds, n, eblk, lt, lazy != 0);
        lds_put_layout<0, LOGE>(x, lds, lt);
    } else {
        lds_get_layout<0, LOGE>(x, lds, lt);
        block_inverse_core<A, LOGB, false, LOGE, Hook>(ar, x, lds, n, eblk, lt, log_n == LOGB, lazy != 0, after_stage);
        // mirror of the forward direction's direct loads: the last inverse pass leaves register k of thread lt =
        // element (k << POS0) + lt, stored as 8-byte words (512 contiguous bytes per wave instruction)
        constexpr bool kDirectStore = Cfg::BPW == 1 && (LOGB - LOGE) >= 6 && !A::kPacked
            ;
        if constexpr (kDirectStore) {
            before_store();
            if (valid) {
#pragma unroll
                // (inside the pipelined inverse kernel this is the INTERMEDIATE: see kPipeIntermediateNt)
                for (int k = 0; k < Cfg::E; ++k)
                    gstore<(kNt && (std::is_same<Hook, NoHook>::value || kPipeIntermediateNt))>(
                        gptr + ((u32)k << (LOGB - LOGE)) + lt, x[k]);
            }
            return;
        }
        lds_put_layout<LOGB - LOGE, LOGE>(x, lds, lt);
    }
    if constexpr (!INV) sync_vectors_layout0<LOGB, LOGE, kWL>();  // from layout <0>: wave-local
    else __syncthreads();
    lds_get_vectors<LOGB, LOGE, kWL>(io, lds, lt);
    before_store();
    if (valid) store_block_vectors<LOGB, LOGE, kNt, kWL>(io, gptr, lt);
}


// Three register budgets, by block size (the attributes do not take template arguments, hence three kernels):
//   2^10 .. 2^12: four waves per SIMD (LDS) and at most 104 registers — measured on the 2^12 pass under the strided passes;
//   2^13, 2^14:   four waves per SIMD (one or two workgroups of 512 / 1024 threads per CU), registers left to the compiler;
//   below 2^10:   several polynomials per wave, the primes are per-lane values in VGPRs: no cap at all (the 104-register
//                 cap cost these kernels 20-84 bytes of scratch per lane).
template <class A, int LOGB, bool INV, bool MUL = false, bool NT = false>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4), amdgpu_num_vgpr(104))) void ntt_block_kernel(
    u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u32 log_n, u64 total_blocks, u32 lazy,
    const u64 *__restrict__ mul, u64 mul_polys) {
    static_assert(LOGB >= 10 && LOGB <= 12, "the capped form is for blocks of 2^10 .. 2^12");
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    block_pass_body<A, LOGB, INV, MUL, 4, NoHook, NT>(data, data, primes, L, log_n, total_blocks, lazy, mul, mul_polys, lds_raw,
                                                      blockIdx.x);
}
template <class A, int LOGB, bool INV, bool MUL = false, bool NT = false>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_block_big_kernel(
    u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u32 log_n, u64 total_blocks, u32 lazy,
    const u64 *__restrict__ mul, u64 mul_polys) {
    static_assert(LOGB >= 13, "blocks of 2^13 and 2^14");
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    block_pass_body<A, LOGB, INV, MUL, 4, NoHook, NT>(data, data, primes, L, log_n, total_blocks, lazy, mul, mul_polys, lds_raw,
                                                      blockIdx.x);
}
template <class A, int LOGB, bool INV, bool MUL = false, bool NT = false>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) void ntt_block_small_kernel(
    u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u32 log_n, u64 total_blocks, u32 lazy,
    const u64 *__restrict__ mul, u64 mul_polys) {
    static_assert(LOGB < 10, "blocks below 2^10");
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    block_pass_body<A, LOGB, INV, MUL, 4, NoHook, NT>(data, data, primes, L, log_n, total_blocks, lazy, mul, mul_polys, lds_raw,
                                                      blockIdx.x);
}

// out-of-place form of the 2^12 block pass (host-pointer path, ntt_transform_through_dev)
template <class A, bool INV>
__global__ __launch_bounds__(BlockCfg<12>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_block_oop_kernel(
    u64 *__restrict__ dst, const u64 *__restrict__ src, const NttPrime *__restrict__ primes, u32 L, u32 log_n, u64 total_blocks,
    u32 lazy) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    block_pass_body<A, 12, INV, false, 4, NoHook, false>(dst, src, primes, L, log_n, total_blocks, lazy, nullptr, 0, lds_raw,
                                                         blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// Persistent single-pass transform for N = 2^13 / 2^14 (BASELINE config 2).  A polynomial of 2^14 words fills a CU's LDS
// (144 KiB): one 1024-thread workgroup per CU, whose sixteen waves would all wait for their loads at the same time, then
// all compute, then all store.  Here a workgroup stays resident and walks over polynomials p, p + G, p + 2G, ...; the loads
// of the NEXT polynomial are issued in front of the current one's LAST register pass (forward: two stages for 2^14, per-lane
// twiddles four at a time; inverse: the pass with wave-uniform twiddles) into a second set of 32 registers and land during
// that pass and the write-back, so only the first polynomial of a workgroup is waited for with nothing else to do.
// Forward: 8-byte loads deliver the first register layout directly (register k of thread lt = element (k << POS0) + lt);
// inverse: natural-order 16-byte vectors, staged through LDS into layout 0, and direct 8-byte stores at the end.
// ------------------------------------------------------------------------------------------
// The E loads of a forward register layout: element (k << POS0) + lt, rows 8 << POS0 bytes apart — beyond the 12-bit
// immediate offset of a global load, so every row needs its own 64-bit address.  Row k's address is row k-1's plus an
// OPAQUE scalar stride: one v_lshl_add_u64 each instead of v_add_co_u32 + s_nop + v_addc_co_u32 (43 -> 13 such pairs in
// the 2^14 kernel, VALU 2017 -> 1989: 13.46 -> 13.59 M NTT/s same box; the same chain on the inverse kernel's stores
// measured -0.6 % and is not used — profiles/r06_experiments.txt item 1).
template <int E, int POS0>
__device__ __forceinline__ void persist_row_loads(u64 (&x)[E], const u64 *g) {
    u64 stride = (u64)sizeof(u64) << POS0;
    asm("" : "+s"(stride));
    const char *addr = reinterpret_cast<const char *>(g);
#pragma unroll
    for (int k = 0; k < E; ++k) {
        x[k] = __builtin_nontemporal_load(reinterpret_cast<const u64 *>(addr));
        addr += stride;
    }
}
constexpr int kPersistInvHook = 1;  // passes before the last one at which the inverse direction issues its prefetch (0: 0.332 ms, 1: 0.315 ms per 4096)
template <class A, int LOGB, bool INV>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_persist_kernel(
    u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u64 npolys, u32 lazy) {
    using Cfg = BlockCfg<LOGB>;
    static_assert(Cfg::BPW == 1 && LOGB - 4 >= 6, "one polynomial per workgroup, whole waves per register layout");
    constexpr int NV = Cfg::E / 2, POS0 = LOGB - 4;
    constexpr u32 n = 1u << LOGB;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 p = blockIdx.x;
    if (p >= npolys) return;
    const u64 stride = gridDim.x;
    if constexpr (!INV) {
        u64 x[Cfg::E], xn[Cfg::E];
        {
            persist_row_loads<Cfg::E, POS0>(x, data + p * n + threadIdx.x);
        }
        while (true) {
            const u64 pn = p + stride;
            const bool more = pn < npolys;
            const A ar(primes + p % L);
            const auto prefetch = [&]() {
                if (more) {
                    persist_row_loads<Cfg::E, POS0>(xn, data + pn * n + opaque_tid());
                } else {  // a dead value: the old contents must not count as live
#pragma unroll
                    for (int k = 0; k < Cfg::E; ++k) xn[k] = 0;
                }
            };
            {
                const u32 lt = opaque_tid();
                // (leading barrier of the first exchange: the previous polynomial's write-back may still be read)
                // The next polynomial's loads go out AFTER the last register pass: in front of it (per-lane twiddles) the
                // 32 extra registers cost 12-26 spilled ones and the kernel loses to the plain one (0.402 vs 0.356 ms).
                block_forward_core<A, LOGB, true>(ar, x, lds, n, 0u, lt, lazy != 0);
                prefetch();
                lds_put_layout<0>(x, lds, lt);
            }
            sync_vectors_layout0<LOGB>();  // wave-local (vec_index)
            {
                const u32 lt = opaque_tid();
                u64x2 io[NV];
                lds_get_vectors<LOGB>(io, lds, lt);
                store_block_vectors<LOGB, 4, true>(io, data + p * n, lt);
            }
            if (!more) break;
#pragma unroll
            for (int k = 0; k < Cfg::E; ++k) x[k] = xn[k];
            p = pn;
        }
    } else {
        u64x2 io[NV], ion[NV];
        load_block_vectors<LOGB, 4, true>(io, data + p * n, threadIdx.x);
        bool first = true;
        while (true) {
            const u64 pn = p + stride;
            const bool more = pn < npolys;
            const A ar(primes + p % L);
            const auto prefetch = [&]() {
                if (more) {
                    load_block_vectors<LOGB, 4, true>(ion, data + pn * n, opaque_tid());
                } else {
#pragma unroll
                    for (int j = 0; j < NV; ++j) ion[j] = u64x2{0, 0};
                }
            };
            u64 x[Cfg::E];
            {
                const u32 lt = opaque_tid();
                if (!first) __syncthreads();  // the previous polynomial's last exchange may still be read
                lds_put_vectors<LOGB>(io, lds, lt);
                sync_vectors_layout0<LOGB>();  // wave-local (vec_index)
                lds_get_layout<0>(x, lds, lt);
                block_inverse_core<A, LOGB, false, 4, decltype(prefetch), kPersistInvHook>(ar, x, lds, n, 0u, lt, true, lazy != 0, prefetch);
            }
            {
                u64 *__restrict__ g = data + p * n + opaque_tid();
#pragma unroll
                for (int k = 0; k < Cfg::E; ++k) gstore<true>(g + ((u32)k << POS0), x[k]);
            }
            if (!more) break;
#pragma unroll
            for (int j = 0; j < NV; ++j) io[j] = ion[j];
            p = pn;
            first = false;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pipelined two-pass transform, N = 2^16 (4 strided stages + blocks of 2^12): ONE kernel per tile boundary.  Workgroup
// i runs the block pass on block i of tile A and the strided pass on chunk i (256 columns x 16 rows) of tile B, with
// the strided chunk's loads issued right after the block's own loads have landed and consumed after the block's twelve
// stages: the HBM latency of the strided pass hides behind the block pass's arithmetic INSIDE each wave, instead of
// relying on two kernels of different register footprints sharing a SIMD.  Forward: A = tile k-1, B = tile k
// (strided first); inverse: A = tile k, B = tile k-1 (block first).  Both tiles start at a multiple of L polynomials,
// so block i and chunk i belong to the same limb (16 blocks and 16 chunks per limb-polynomial).
// ------------------------------------------------------------------------------------------
// Register budgets: both directions run at four waves per SIMD without spilling — the forward instantiations in 124
// registers (5.00-5.01 against 5.06 ms per 12 288 transforms at three waves), the inverse ones in 128 once the strided
// chunk's loads are issued in front of the block pass's LAST register pass (block_pass_body; issued after the staging
// they spill 6 registers there: 5.42 ms).
template <class A, int LOGB, bool INV, bool MUL>
__device__ __forceinline__ void ntt_pipe_body(
    u64 *__restrict__ blk_data, u64 blk_total, u64 *__restrict__ str_data, u64 str_total,
    const NttPrime *__restrict__ primes, u32 L, u32 lazy, const u64 *__restrict__ mul, u64 mul_polys) {
    // words per polynomial = 2^K blocks of 2^LOGB = 16 chunks of TPB columns: block i and chunk i share their limb
    constexpr int K = 4, TPB = BlockCfg<LOGB>::TPB;
    static_assert(BlockCfg<LOGB>::BPW == 1 && (1 << LOGB) / TPB == (1 << K), "16 blocks and 16 chunks per polynomial");
    constexpr u32 log_n = LOGB + K, n = 1u << log_n;
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    const u64 chunk = blockIdx.x;
    const bool has_str = chunk < str_total;
    // chunk -> (limb-polynomial, TPB columns): thread t owns column (chunk % 16) * TPB + t, rows k * 2^LOGB
    u64 *__restrict__ sp = str_data + (chunk >> 4) * n + (chunk & 15) * TPB + threadIdx.x;
    u64 sx[1 << K][1];
    const auto issue = [&]() {
        if (has_str) {
#pragma unroll
#ifdef PFHE_NOMEM
            for (int k = 0; k < (1 << K); ++k) sx[k][0] = (u64)threadIdx.x * 0xBF58476D1CE4E5B9ull + (u64)k + lazy;
#else
            for (int k = 0; k < (1 << K); ++k) sx[k][0] = __builtin_nontemporal_load(sp + ((u64)k << LOGB));
#endif
        }
    };
    // The chunk's loads are consumed after the block's stores have been issued.  Loads and stores share the wave's one
    // vector-memory counter and the compiler, with both kinds pending, cannot count past the stores: it would wait for
    // vmcnt(0) — for the block's stores to be ACKNOWLEDGED — in front of the chunk's first butterfly (1.5 k of a workgroup's
    // 42 k cycles in round 2's phase stamps: "wait: block stores + strided loads landed").  Touching the chunk's registers
    // in front of the stores moves the wait to a point where only loads are pending — issued twelve stages earlier, long
    // landed — and the stores are then never waited for.
    // (forward only: the inverse kernels issue the chunk's loads one register pass before the stores — the touch would wait
    // for them there — and the Montgomery instantiation spills four registers with it; measured: no difference either way)
    const auto landed = [&]() {
        if (!INV && has_str) {
#pragma unroll
            for (int k = 0; k < (1 << K); ++k) asm volatile("" : "+v"(sx[k][0]));
        }
    };
    if (chunk < blk_total) {
        block_pass_body<A, LOGB, INV, MUL, 4, decltype(issue), false, decltype(landed)>(
            blk_data, blk_data, primes, L, log_n, blk_total, INV ? 0u : lazy, mul, mul_polys, lds_raw, chunk, issue, landed);
    } else {
        issue();
        landed();  // (both paths reach the chunk's stages with no load pending: the wait counts below the join serve both)
    }
    if (has_str) {
        const A ar(primes + (chunk >> 4) % L);
        if constexpr (!INV) strided_forward_regs<A, K, 1, true>(ar, sx, n, 0u, LOGB);
        else strided_inverse_regs<A, K, 1, true>(ar, sx, n, 0u, LOGB, lazy != 0);  // the only strided pass: final stage
#ifdef PFHE_NOMEM
        if (lazy == 0xdeadu)
#endif
#pragma unroll
        // forward: this is the intermediate (kPipeIntermediateNt); inverse: the final output (always non-temporal)
        for (int k = 0; k < (1 << K); ++k) gstore<(INV || kPipeIntermediateNt)>(sp + ((u64)k << LOGB), sx[k][0]);
    }
}

template <class A, int LOGB>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_pipe_fwd_kernel(
    u64 *__restrict__ blk_data, u64 blk_total, u64 *__restrict__ str_data, u64 str_total,
    const NttPrime *__restrict__ primes, u32 L, u32 lazy) {
    ntt_pipe_body<A, LOGB, false, false>(blk_data, blk_total, str_data, str_total, primes, L, lazy, nullptr, 0);
}
template <class A, int LOGB, bool MUL>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_pipe_inv_kernel(
    u64 *__restrict__ blk_data, u64 blk_total, u64 *__restrict__ str_data, u64 str_total,
    const NttPrime *__restrict__ primes, u32 L, u32 lazy, const u64 *__restrict__ mul, u64 mul_polys) {
    ntt_pipe_body<A, LOGB, true, MUL>(blk_data, blk_total, str_data, str_total, primes, L, lazy, mul, mul_polys);
}

// ------------------------------------------------------------------------------------------
// NTT -> pointwise product -> INTT on one block (CrtRlwe::mul_dcrt_polynomial_to, primus_lattice/src/rlwe/crt.rs:42-65,
// + DcrtPolynomial::into_coeff_form, macros/mod.rs:901-911): the forward transform's block pass, the product and the
// inverse transform's block pass all act on the same contiguous block of 2^LOGB coefficients, so ONE workgroup runs all
// three while the block is on chip.  For N = 2^LOGB (single-pass rings, 2^10 .. 2^14) that is the whole product with one
// HBM read and one write per coefficient; for two-pass rings it sits between the two strided passes: 3 HBM round trips
// (48*N bytes per limb-polynomial) instead of the 4 of transform + fused inverse-mul.
// The forward half leaves lazy values (one fold), the product takes any representative below 2^63 and a canonical
// multiplicand and returns [0, 2q), which the inverse butterflies accept: the canonical final output is the reference's.
// Hooks (pipelined form): after_load runs once the block's own loads are issued and consumed by the first stages;
// mid runs while the block sits in LDS between the halves (no block registers live); late runs in front of the inverse
// half's last register pass (the pass with the fewest live registers).
// ------------------------------------------------------------------------------------------
// gptr / mptr: this block of the data and of the multiplicand; ar: the arithmetic of the block's limb
// NT: non-temporal data loads / stores (large batches); MNT: non-temporal loads of the multiplicand too — only when it is
// per-element (read once); a multiplicand shared by the batch is re-read by every workgroup and must stay cacheable.
template <class A, int LOGB, class HookA = NoHook, class HookM = NoHook, class HookL = NoLateHook, bool NT = false, bool MNT = false>
__device__ __forceinline__ void block_mid_body(const A &ar, u64 *__restrict__ gptr, const u64 *__restrict__ mptr, bool valid,
                                               u32 n, u32 eblk, bool final_block, u64 *__restrict__ lds,
                                               HookA after_load = HookA(), HookM mid = HookM(), HookL late = HookL()) {
    using Cfg = BlockCfg<LOGB>;
    static_assert(Cfg::BPW == 1 && LOGB - 4 >= 6, "one block per workgroup, whole waves per register layout");
    constexpr int NV = Cfg::E / 2, POS0 = LOGB - 4;
    u64 x[Cfg::E];
    {
        const u32 lt = opaque_tid();
        // register k of thread lt = element (k << POS0) + lt: 8-byte loads, 512 contiguous bytes per wave instruction
#pragma unroll
        for (int k = 0; k < Cfg::E; ++k) x[k] = valid ? __builtin_nontemporal_load(gptr + ((u32)k << POS0) + lt) : 0ull;
        after_load();
        block_forward_core<A, LOGB, false, 4, NoLateHook, true>(ar, x, lds, n, eblk, lt, /*lazy=*/true);  // raw: feeds the product
        lds_put_layout<0>(x, lds, lt);
    }
    {
        const u32 lt = opaque_tid();
        u64x2 io[NV], mv[NV];
        if (valid) {  // natural order, 1 KiB per wave instruction
            load_block_vectors<LOGB, 4, MNT>(mv, mptr, lt);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) mv[j] = u64x2{0, 0};
        }
        // layout <0> <-> I/O vectors: both transpositions stay inside the wave (vec_index, pfhe_ntt_device.hpp): the two
        // workgroup barriers that stood here until round 4 are wave-level syncs
        sync_vectors_layout0<LOGB>();
        lds_get_vectors<LOGB>(io, lds, lt);
#pragma unroll
        for (int j = 0; j < NV; ++j) {  // (outside any `valid` branch: see the note in block_pass_body)
            io[j].x = ar.mul_any(io[j].x, mv[j].x);
            io[j].y = ar.mul_any(io[j].y, mv[j].y);
        }
        lds_put_vectors<LOGB>(io, lds, lt);  // the slots this thread just read
    }
    mid();
    sync_vectors_layout0<LOGB>();
    {
        const u32 lt = opaque_tid();
        lds_get_layout<0>(x, lds, lt);
        block_inverse_core<A, LOGB, false, 4, HookL>(ar, x, lds, n, eblk, lt, final_block, /*lazy=*/false, late);
    }
    if (valid) {
        const u32 lt = opaque_tid();
#pragma unroll
        for (int k = 0; k < Cfg::E; ++k) gstore<NT>(gptr + ((u32)k << POS0) + lt, x[k]);
    }
}

template <class A, int LOGB, bool NT, bool MNT = false>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_block_mid_kernel(
    u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u32 log_n, u64 total_blocks,
    const u64 *__restrict__ mul, u64 mul_polys) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    u64 blk = blockIdx.x;
    const bool valid = blk < total_blocks;
    if (!valid) blk = 0;
    const u32 log_nb = log_n - LOGB;
    const u64 pid = blk >> log_nb;
    const u32 limb = (u32)(pid % L);
    const u32 eblk = (u32)(blk & ((1ull << log_nb) - 1)) << LOGB;
    const u32 n = 1u << log_n;
    // The multiplicand is one RNS polynomial (mul_polys == L, indexed by the limb) or one polynomial per data polynomial
    const u64 mpoly = mul_polys == (u64)L ? (u64)limb : pid;
    const A ar(primes + limb);
    block_mid_body<A, LOGB, NoHook, NoHook, NoLateHook, NT, MNT>(ar, data + pid * n + eblk, mul + mpoly * n + eblk, valid, n,
                                                                 eblk, log_n == LOGB, lds_raw);
}

// Persistent form of the middle kernel for N = 2^14 (one 1024-thread workgroup per CU, see ntt_persist_kernel): a resident
// workgroup walks over polynomials and issues the next one's loads in front of the inverse half's second-to-last register
// pass (wave-uniform twiddles from there on), so that only its first polynomial is waited for with nothing else to do.
template <class A, int LOGB>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_persist_mid_kernel(
    u64 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u64 npolys, const u64 *__restrict__ mul, u64 mul_polys) {
    using Cfg = BlockCfg<LOGB>;
    static_assert(Cfg::BPW == 1 && LOGB - 4 >= 6, "one polynomial per workgroup, whole waves per register layout");
    constexpr int NV = Cfg::E / 2, POS0 = LOGB - 4;
    constexpr u32 n = 1u << LOGB;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 p = blockIdx.x;
    if (p >= npolys) return;
    const u64 stride = gridDim.x;
    u64 x[Cfg::E], xn[Cfg::E];
    {
        const u64 *__restrict__ g = data + p * n + threadIdx.x;
#pragma unroll
        for (int k = 0; k < Cfg::E; ++k) x[k] = __builtin_nontemporal_load(g + ((u32)k << POS0));
    }
    while (true) {
        const u64 pn = p + stride;
        const bool more = pn < npolys;
        const u32 limb = (u32)(p % L);
        const A ar(primes + limb);
        const auto prefetch = [&]() {
            if (more) {
                const u64 *__restrict__ g = data + pn * n + opaque_tid();
#pragma unroll
                for (int k = 0; k < Cfg::E; ++k) xn[k] = __builtin_nontemporal_load(g + ((u32)k << POS0));
            } else {  // a dead value: the old contents must not count as live
#pragma unroll
                for (int k = 0; k < Cfg::E; ++k) xn[k] = 0;
            }
        };
        {
            const u32 lt = opaque_tid();
            // (leading barrier of the first exchange: the previous polynomial's last exchange may still be read)
            block_forward_core<A, LOGB, true, 4, NoLateHook, true>(ar, x, lds, n, 0u, lt, /*lazy=*/true);  // raw: feeds the product
            lds_put_layout<0>(x, lds, lt);
        }
        {
            const u32 lt = opaque_tid();
            u64x2 io[NV], mv[NV];
            const u64 mpoly = mul_polys == (u64)L ? (u64)limb : p;
            load_block_vectors<LOGB, 4, false>(mv, mul + mpoly * n, lt);
            sync_vectors_layout0<LOGB>();  // layout <0> <-> I/O vectors: wave-local both ways (vec_index)
            lds_get_vectors<LOGB>(io, lds, lt);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                io[j].x = ar.mul_any(io[j].x, mv[j].x);
                io[j].y = ar.mul_any(io[j].y, mv[j].y);
            }
            lds_put_vectors<LOGB>(io, lds, lt);  // the slots this thread just read
        }
        sync_vectors_layout0<LOGB>();
        {
            const u32 lt = opaque_tid();
            lds_get_layout<0>(x, lds, lt);
            block_inverse_core<A, LOGB, false, 4, decltype(prefetch), kPersistInvHook>(ar, x, lds, n, 0u, lt, true, false, prefetch);
        }
        {
            u64 *__restrict__ g = data + p * n + opaque_tid();
#pragma unroll
            for (int k = 0; k < Cfg::E; ++k) gstore<true>(g + ((u32)k << POS0), x[k]);
        }
        if (!more) break;
#pragma unroll
        for (int k = 0; k < Cfg::E; ++k) x[k] = xn[k];
        p = pn;
    }
}

// Pipelined form for N = 2^16 (4 strided stages, blocks of 2^12), large batches: launch k runs, in workgroup i, the
// middle kernel on block i of tile k-1, the FORWARD strided pass on chunk i (256 columns x 16 rows) of tile k and the
// INVERSE (final) strided pass on chunk i of tile k-2.  The forward chunk's loads are in flight during the block's
// forward half and consumed while the block sits in LDS between the halves; the inverse chunk's loads are issued in
// front of the inverse half's last register pass and consumed after the block's stores.  One set of 32 registers
// serves both chunks.
template <class A, int LOGB>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_pipe_mid_kernel(
    u64 *__restrict__ mid_data, u64 mid_total, u64 *__restrict__ sf_data, u64 sf_total, u64 *__restrict__ si_data,
    u64 si_total, const NttPrime *__restrict__ primes, u32 L, const u64 *__restrict__ mul, u64 mul_polys) {
    constexpr int K = 4, TPB = BlockCfg<LOGB>::TPB;
    static_assert((1 << LOGB) / TPB == (1 << K), "16 blocks and 16 chunks per polynomial");
    constexpr u32 log_n = LOGB + K, n = 1u << log_n;
    extern __shared__ __attribute__((aligned(16))) u64 lds_raw[];
    const u64 chunk = blockIdx.x;
    const bool has_sf = chunk < sf_total, has_si = chunk < si_total;
    const u64 cbase = (chunk >> 4) * n + (chunk & 15) * TPB;  // wave-uniform
    const A ars(primes + (chunk >> 4) % L);  // tiles start at multiples of L polynomials: one limb for all three roles
    u64 sx[1 << K][1];
    const auto issue_f = [&]() {
        if (has_sf) {
            const u64 *__restrict__ sp = sf_data + cbase + opaque_tid();
#pragma unroll
            for (int k = 0; k < (1 << K); ++k) sx[k][0] = __builtin_nontemporal_load(sp + ((u64)k << LOGB));
        }
    };
    const auto finish_f = [&]() {
        if (has_sf) {
            strided_forward_regs<A, K, 1, true>(ars, sx, n, 0u, LOGB);
            u64 *__restrict__ sp = sf_data + cbase + opaque_tid();
#pragma unroll
            for (int k = 0; k < (1 << K); ++k) gstore<kPipeIntermediateNt>(sp + ((u64)k << LOGB), sx[k][0]);
        }
    };
    const auto issue_i = [&]() {
        if (has_si) {
            const u64 *__restrict__ sp = si_data + cbase + opaque_tid();
#pragma unroll
            for (int k = 0; k < (1 << K); ++k) sx[k][0] = sp[(u64)k << LOGB];  // intermediate: plain (Infinity Cache)
        }
    };
    if (chunk < mid_total) {
        const u64 pid = chunk >> 4;
        const u32 eblk = (u32)(chunk & 15) << LOGB;
        const u64 mpoly = mul_polys == (u64)L ? pid % L : pid;
        block_mid_body<A, LOGB>(ars, mid_data + pid * n + eblk, mul + mpoly * n + eblk, true, n, eblk, false, lds_raw, issue_f,
                                finish_f, issue_i);
    } else {
        issue_f();
        finish_f();
        issue_i();
    }
    if (has_si) {
        strided_inverse_regs<A, K, 1, true>(ars, sx, n, 0u, LOGB, /*lazy=*/false);
        u64 *__restrict__ sp = si_data + cbase + opaque_tid();
#pragma unroll
        for (int k = 0; k < (1 << K); ++k) gstore<true>(sp + ((u64)k << LOGB), sx[k][0]);
    }
}

// ------------------------------------------------------------------------------------------
// host-side planner / launchers
// ------------------------------------------------------------------------------------------
namespace {

// Large batches take the instantiations with non-temporal stores / staged loads (pfhe_ntt_device.hpp, gstore): measured
// at 4096 polynomials, block pass of 2^12: 3.18 vs 3.25 ms; N = 2^13: 0.48 vs 0.51 ms; N = 2^14 inverse 0.370 vs 0.392 ms;
// u32 tables 2.90 vs 2.96 ms; strided pass 2.09 vs 2.19 ms.  Small batches, which the next kernel finds in the Infinity
// Cache, keep the plain forms (192 MiB, two passes: 0.181 ms plain, 0.190 ms non-temporal).
constexpr u64 kNtMinBytes = 256ull << 20;

template <class A, int LOGB, bool INV, bool MUL, bool NT>
int launch_block_impl(u64 *data, const NttPrime *primes, u32 L, u32 log_n, u64 npolys, bool lazy, hipStream_t s,
                      const u64 *mul, u64 mul_polys) {
    using Cfg = BlockCfg<LOGB>;
    const u64 total_blocks = npolys << (log_n - LOGB);
    const u64 grid = (total_blocks + Cfg::BPW - 1) / Cfg::BPW;
    if (grid == 0) return PFHE_OK;
    if (grid > 0x7fffffffull) {
        set_last_error("batch too large for one launch");
        return PFHE_ERR_BAD_LENGTH;
    }
    constexpr size_t lds_bytes = (size_t)Cfg::BPW * Cfg::LDS_WORDS * sizeof(u64);
    void (*kern)(u64 *, const NttPrime *, u32, u32, u64, u32, const u64 *, u64);
    if constexpr (LOGB >= 13) kern = ntt_block_big_kernel<A, LOGB, INV, MUL, NT>;
    else if constexpr (LOGB >= 10) kern = ntt_block_kernel<A, LOGB, INV, MUL, NT>;
    else kern = ntt_block_small_kernel<A, LOGB, INV, MUL, NT>;
    if (lds_bytes > 64 * 1024) {
        static thread_local bool configured[64] = {};
        int dev = 0;
        PFHE_HIP(hipGetDevice(&dev));
        const bool cached = dev >= 0 && dev < 64;  // outside the cache: set on every launch
        if (!cached || !configured[dev]) {
            PFHE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            if (cached) configured[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((u32)grid), dim3(Cfg::THREADS), lds_bytes, s, data, primes, L, log_n,
                       total_blocks, lazy ? 1u : 0u, mul, mul_polys);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// CUs of the current device (cached per device)
static int device_cu_count() {
    static thread_local int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev] = v;
    }
    return cached[dev];
}

// persistent single-pass form (ntt_persist_kernel): N = 2^13 / 2^14, at least two polynomials per resident workgroup
template <class A, int LOGB, bool INV>
int launch_persist(u64 *data, const NttPrime *primes, u32 L, u64 npolys, bool lazy, hipStream_t s, u64 resident) {
    using Cfg = BlockCfg<LOGB>;
    constexpr size_t lds_bytes = (size_t)Cfg::LDS_WORDS * sizeof(u64);
    void (*kern)(u64 *, const NttPrime *, u32, u64, u32) = ntt_persist_kernel<A, LOGB, INV>;
    static thread_local bool configured[64] = {};
    int dev = 0;
    PFHE_HIP(hipGetDevice(&dev));
    // (a device index outside the cache sets the attribute on every launch, as polymul_impl does)
    const bool cached = dev >= 0 && dev < 64;
    if (!cached || !configured[dev]) {
        PFHE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes));
        if (cached) configured[dev] = true;
    }
    // equal shares: every workgroup walks ceil(npolys / grid) polynomials, give or take one
    const u64 rounds = (npolys + resident - 1) / resident;
    const u64 grid = (npolys + rounds - 1) / rounds;
    hipLaunchKernelGGL(kern, dim3((u32)grid), dim3(Cfg::THREADS), lds_bytes, s, data, primes, L, npolys, lazy ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class A, int LOGB, bool INV, bool MUL = false>
int launch_block(u64 *data, const NttPrime *primes, u32 L, u32 log_n, u64 npolys, bool lazy, hipStream_t s,
                 const u64 *mul = nullptr, u64 mul_polys = 0) {
    // (N = 2^13, two workgroups per CU, measures slower in this form: 0.33 vs 0.29-0.32 ms per 8192 polynomials)
    if constexpr (LOGB == 14 && !MUL && !A::kPacked) {
        const u64 resident = (u64)device_cu_count();
        if (log_n == LOGB && npolys >= 2 * resident)
            return launch_persist<A, LOGB, INV>(data, primes, L, npolys, lazy, s, resident);
    }
    if constexpr (LOGB >= 11) {
        if ((npolys << log_n) * sizeof(u64) >= kNtMinBytes)
            return launch_block_impl<A, LOGB, INV, MUL, true>(data, primes, L, log_n, npolys, lazy, s, mul, mul_polys);
    }
    return launch_block_impl<A, LOGB, INV, MUL, false>(data, primes, L, log_n, npolys, lazy, s, mul, mul_polys);
}

template <class A, bool INV, bool MUL = false>
int dispatch_block(int logb, u64 *data, const NttPrime *primes, u32 L, u32 log_n, u64 npolys, bool lazy,
                   hipStream_t s, const u64 *mul = nullptr, u64 mul_polys = 0) {
    switch (logb) {
#define PFHE_CASE(B) \
    case B:          \
        return launch_block<A, B, INV, MUL>(data, primes, L, log_n, npolys, lazy, s, mul, mul_polys);
        PFHE_CASE(4) PFHE_CASE(5) PFHE_CASE(6) PFHE_CASE(7) PFHE_CASE(8) PFHE_CASE(9) PFHE_CASE(10)
        PFHE_CASE(11) PFHE_CASE(12) PFHE_CASE(13) PFHE_CASE(14)
#undef PFHE_CASE
    }
    set_last_error("unsupported block size");
    return PFHE_ERR_UNSUPPORTED;
}

template <class A, int K, int VEC, bool INV, bool FINAL>
int launch_strided(u64 *data, const NttPrime *primes, u32 L, u32 log_n, u32 log_s, u64 npolys, bool lazy,
                   hipStream_t s) {
    const u64 total = (npolys << (log_n - K)) >> (VEC == 2 ? 1 : 0);
    const u64 grid = (total + 255) / 256;
    if (grid == 0) return PFHE_OK;
    if (grid > 0x7fffffffull) {
        set_last_error("batch too large for one launch");
        return PFHE_ERR_BAD_LENGTH;
    }
    if (K >= 3 && (npolys << log_n) * sizeof(u64) >= kNtMinBytes)
        hipLaunchKernelGGL((ntt_strided_kernel<A, K, VEC, INV, FINAL, (K >= 3)>), dim3((u32)grid), dim3(256), 0, s, data,
                           primes, L, log_n, log_s, total, lazy ? 1u : 0u);
    else
        hipLaunchKernelGGL((ntt_strided_kernel<A, K, VEC, INV, FINAL>), dim3((u32)grid), dim3(256), 0, s, data, primes,
                           L, log_n, log_s, total, lazy ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class A, bool INV, bool FINAL>
int dispatch_strided(int k, u64 *data, const NttPrime *primes, u32 L, u32 log_n, u32 log_s, u64 npolys, bool lazy,
                     hipStream_t s) {
    // the radices make_ntt_plan produces: the stages above a 2^12 block (2^11 words for the u32 tables at N = 2^16), at
    // most five per pass, balanced — never fewer than three
    switch (k) {
        case 3: return launch_strided<A, 3, 2, INV, FINAL>(data, primes, L, log_n, log_s, npolys, lazy, s);
        case 4: return launch_strided<A, 4, 2, INV, FINAL>(data, primes, L, log_n, log_s, npolys, lazy, s);
        case 5: return launch_strided<A, 5, 1, INV, FINAL>(data, primes, L, log_n, log_s, npolys, lazy, s);
    }
    set_last_error("unsupported strided radix");
    return PFHE_ERR_UNSUPPORTED;
}

int launch_tiny(bool inverse, const NttPrime *primes, u32 L, u32 log_n, u64 *data, u64 npolys, bool lazy,
                hipStream_t s) {
    if (log_n == 0 || npolys == 0) return PFHE_OK;
    const dim3 g((u32)((npolys + 255) / 256)), t(256);
    if (inverse) hipLaunchKernelGGL(ntt_tiny_kernel<true>, g, t, 0, s, data, primes, L, log_n, npolys, lazy ? 1u : 0u);
    else hipLaunchKernelGGL(ntt_tiny_kernel<false>, g, t, 0, s, data, primes, L, log_n, npolys, lazy ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// one pass of the plan, in execution order (forward: strided passes then block; inverse: block
// then strided passes, the last of which carries the fused final stage)
template <class A>
int run_pass(const NttPlan &plan, const NttPrime *primes, u32 L, u32 log_n, u64 *data, u64 npolys, bool inverse,
             int index, bool lazy, hipStream_t s, const u64 *mul, u64 mul_polys) {
    const int block_at = inverse ? 0 : plan.n_strided;
    if (mul != nullptr) {
        if constexpr (A::kPacked) {
            return PFHE_ERR_UNSUPPORTED;
        } else {
            if (!inverse || index != block_at) return PFHE_ERR_BAD_ARGUMENT;
            return dispatch_block<A, true, true>(plan.block_log, data, primes, L, log_n, npolys, lazy, s, mul, mul_polys);
        }
    }
    if (index == block_at) {
        return inverse ? dispatch_block<A, true>(plan.block_log, data, primes, L, log_n, npolys, lazy, s)
                       : dispatch_block<A, false>(plan.block_log, data, primes, L, log_n, npolys, lazy, s);
    }
    const int i = inverse ? plan.n_strided - index : index;  // index into plan.strided (forward order)
    u32 log_s = log_n;
    for (int j = 0; j <= i; ++j) log_s -= plan.strided[j];
    const int k = plan.strided[i];
    if (!inverse) {
        // Montgomery tables: the first pass of a forward transform skips the fold of its first stage (strided_forward_regs)
        if constexpr (A::kMont) {
            if (i == 0) return dispatch_strided<A, false, true>(k, data, primes, L, log_n, log_s, npolys, false, s);
        }
        return dispatch_strided<A, false, false>(k, data, primes, L, log_n, log_s, npolys, false, s);
    }
    if (i == 0) return dispatch_strided<A, true, true>(k, data, primes, L, log_n, log_s, npolys, lazy, s);
    return dispatch_strided<A, true, false>(k, data, primes, L, log_n, log_s, npolys, false, s);
}

}  // namespace

static int env_int(const char *name, int lo, int hi) {
    const char *e = std::getenv(name);
    if (!e) return 0;
    const int v = std::atoi(e);
    return v >= lo && v <= hi ? v : 0;
}

NttTuning NttTuning::from_env() {
    NttTuning t;
    t.pipe_tiles = env_int("PFHE_PIPE_TILES", 2, 4096);
    t.pipelined = std::getenv("PFHE_DISABLE_PIPELINED") == nullptr;
    t.pipelined_min_mb = env_int("PFHE_PIPELINED_MIN_MB", 1, 1 << 20);
    return t;
}


NttPlan make_ntt_plan(u32 log_n, int arith, const NttTuning &tune) {
    NttPlan p;
    if (log_n <= 3) {
        p.tiny = true;
        return p;
    }
    (void)tune;  // the plan depends on the ring only (the tuning switches select between FORMS of running it)
    if (log_n <= kMaxSinglePassLog) {
        p.block_log = (int)log_n;
        return p;
    }
    p.block_log = kTwoPassBlockLog;
    // u32 tables at N = 2^16 (2^15 words): 4 strided stages + blocks of 2^11 words in 128-thread workgroups
    // (8 resident per CU) measured 3.05 ms against 3.24 ms for 3 + 2^12
    if (arith == 2 /* kArithB32 */ && log_n == 15) p.block_log = 11;
    int rest = (int)log_n - p.block_log;
    // fewest strided passes with at most 5 stages each, balanced
    int passes = (rest + 4) / 5;
    for (int i = 0; i < passes; ++i) {
        int k = (rest + (passes - i) - 1) / (passes - i);
        p.strided[p.n_strided++] = k;
        rest -= k;
    }
    return p;
}

int ntt_num_passes(u32 log_n, int arith, const NttTuning &tune) {
    const NttPlan plan = make_ntt_plan(log_n, arith, tune);
    return plan.tiny ? 1 : plan.n_strided + 1;
}

void ntt_pass_name(u32 log_n, bool inverse, int index, char *buf, size_t cap, int arith, const NttTuning &tune) {
    const NttPlan plan = make_ntt_plan(log_n, arith, tune);
    if (plan.tiny) {
        std::snprintf(buf, cap, "ntt_tiny_kernel");
        return;
    }
    const int block_at = inverse ? 0 : plan.n_strided;
    if (index == block_at) {
        std::snprintf(buf, cap, "ntt_block_kernel<%d,%s>", plan.block_log, inverse ? "inv" : "fwd");
    } else {
        const int i = inverse ? plan.n_strided - index : index;
        std::snprintf(buf, cap, "ntt_strided_kernel<K=%d,%s>", plan.strided[i], inverse ? "inv" : "fwd");
    }
}

int ntt_pass_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, bool inverse, int index,
                 bool lazy, hipStream_t s, const u64 *mul, u64 mul_polys, const NttTuning &tune) {
    const NttPlan plan = make_ntt_plan(log_n, arith, tune);
    if (index < 0 || index >= (plan.tiny ? 1 : plan.n_strided + 1)) return PFHE_ERR_BAD_ARGUMENT;
    if (arith == kArithB32) {
        if (plan.tiny) return PFHE_ERR_UNSUPPORTED;  // N <= 16 is served by ntt32_tiny_kernel
        return run_pass<B32Arith>(plan, primes, L, log_n, data, npolys, inverse, index, lazy, s, mul, mul_polys);
    }
    if (plan.tiny) return mul ? PFHE_ERR_UNSUPPORTED : launch_tiny(inverse, primes, L, log_n, data, npolys, lazy, s);
    if (arith == kArithMont)
        return run_pass<MontArith>(plan, primes, L, log_n, data, npolys, inverse, index, lazy, s, mul, mul_polys);
    return arith == kArithPm
               ? run_pass<PmArith>(plan, primes, L, log_n, data, npolys, inverse, index, lazy, s, mul, mul_polys)
               : run_pass<ShoupArith>(plan, primes, L, log_n, data, npolys, inverse, index, lazy, s, mul, mul_polys);
}

// pipelined form: from 256 MiB of data (2^16-point transforms: 512 limb-polynomials), tiles of 256 MiB = the Infinity
// Cache, whose share of the intermediate (plain stores, kPipeIntermediateNt) the next launch then reads on-die.  Measured,
// forward / inverse ms per 6 GiB: 24 tiles 4.60-4.63 / 4.61-4.63, 20 tiles 4.83 / 4.80, 32 tiles 4.68 / 4.68, against
// 4.72 / 4.64 for 8 tiles with a non-temporal intermediate; 3 GiB: 12 tiles 2.34 ms (8: 2.46); 1.5 GiB: 6 tiles 1.22 ms
// (12: 1.26); 384 MiB 0.395 -> 0.37 ms against the two plain launches; 96 MiB 0.087 -> 0.096 ms, so smaller batches
// keep those.
// (Until round 3 other two-pass rings — N = 2^15, 2^17 — ran their passes tiled on two internal streams; with today's
// kernels that form measures slower than two full-size launches, 5.08 vs 4.93 ms at 2^15 and 5.13 vs 5.11 ms at 2^17 per
// 6 GiB, and it is gone: every transform runs on the caller's stream only.)
namespace {
constexpr u64 kPipelinedMinBytes = 256ull << 20;
constexpr u64 kPipelinedTileBytes = 256ull << 20;
constexpr int kPipelinedMaxTiles = 64;
}  // namespace

// the pipelined form of the two-pass transform (ntt_pipe_{fwd,inv}_kernel): tiles + 1 launches on the caller's stream
template <class A, int LOGB>
static int transform_pipelined(const NttPrime *primes, u32 L, u64 *data, u64 npolys, bool inverse, bool lazy,
                               hipStream_t s, int tiles, const u64 *mul, u64 mul_polys) {
    if (tiles > 64) tiles = 64;
    constexpr u32 log_n = LOGB + 4;
    constexpr size_t lds_bytes = (size_t)BlockCfg<LOGB>::LDS_WORDS * sizeof(u64);
    constexpr u32 threads = BlockCfg<LOGB>::THREADS;
    const u64 units = npolys / L;
    // equal tiles of whole RNS polynomials (tile weights that ramp up and down again, so that the first and the last
    // launch — one pass's worth of work each — are small, measured no gain and are gone)
    u64 cum[66];
    for (int k = 0; k <= tiles; ++k) cum[k] = units * (u64)k / (u64)tiles;
    for (int k = 0; k <= tiles; ++k) {
        // forward: strided pass of tile k, block pass of tile k-1; inverse: block pass of tile k, strided pass of tile k-1
        const int kb = inverse ? k : k - 1, ks = inverse ? k - 1 : k;
        u64 *bptr = nullptr, *sptr = nullptr;
        const u64 *mptr = nullptr;
        u64 bt = 0, st = 0, mp = 0;
        if (kb >= 0 && kb < tiles) {
            const u64 u0 = cum[kb], u1 = cum[kb + 1];
            bptr = data + ((u0 * L) << log_n);
            bt = ((u1 - u0) * L) << 4;
            // a per-element multiplicand is tiled like the data; a shared one (one unit of L) is not
            if (mul) {
                mptr = mul_polys == npolys ? mul + ((u0 * L) << log_n) : mul;
                mp = mul_polys == npolys ? (u1 - u0) * L : mul_polys;
            }
        }
        if (ks >= 0 && ks < tiles) {
            const u64 u0 = cum[ks], u1 = cum[ks + 1];
            sptr = data + ((u0 * L) << log_n);
            st = ((u1 - u0) * L) << 4;
        }
        const u64 grid = bt > st ? bt : st;
        if (grid == 0) continue;
        if (grid > 0x7fffffffull) {
            set_last_error("batch too large for one launch");
            return PFHE_ERR_BAD_LENGTH;
        }
        if constexpr (!A::kPacked) {
            if (inverse && mul) {
                hipLaunchKernelGGL((ntt_pipe_inv_kernel<A, LOGB, true>), dim3((u32)grid), dim3(threads), lds_bytes, s, bptr, bt,
                                   sptr, st, primes, L, lazy ? 1u : 0u, mptr, mp);
                PFHE_HIP(hipGetLastError());
                continue;
            }
        }
        if (inverse)
            hipLaunchKernelGGL((ntt_pipe_inv_kernel<A, LOGB, false>), dim3((u32)grid), dim3(threads), lds_bytes, s, bptr, bt,
                               sptr, st, primes, L, lazy ? 1u : 0u, mptr, mp);
        else
            hipLaunchKernelGGL((ntt_pipe_fwd_kernel<A, LOGB>), dim3((u32)grid), dim3(threads), lds_bytes, s, bptr, bt, sptr, st,
                               primes, L, lazy ? 1u : 0u);
        PFHE_HIP(hipGetLastError());
    }
    return PFHE_OK;
}

// tiles of the pipelined form for this batch, 0 when the batch does not take it
static int pipelined_tiles(u32 L, u32 log_n, int pm, u64 npolys, bool inverse, bool has_mul, const NttTuning &tune) {
    const u64 bytes = (npolys << log_n) * sizeof(u64);
    // 64-bit tables, N = 2^16 = 2^4 x 2^12.  (The u32 tables' 2^15 words = 2^4 x 2^11 fit the same kernel template; measured
    // 2.858 ms against 2.866-2.874 ms for their two plain launches per 12 288 transforms: not instantiated.)
    // u32 tables: 2^15 words = 2^4 x 2^11, the same kernel template with 128-thread workgroups.  Round 5, 12 288 transforms,
    // same box: INVERSE 2.594 -> 2.462-2.499 ms (6 / 12 tiles); forward 2.512-2.527 -> 2.472-2.480 ms once the wave-local
    // I/O vectors had brought its kernel from 128 registers + 28 bytes of scratch to 120 and none (before that: 2.468 ->
    // 2.473, not taken).  Both directions take it.
    const bool shape = (pm != kArithB32 && log_n == 16 && make_ntt_plan(log_n, pm, tune).block_log == 12) ||
                       (pm == kArithB32 && log_n == 15 &&
                        make_ntt_plan(log_n, pm, tune).block_log == 11);
    if (!(tune.pipelined && shape && ntt_num_passes(log_n, pm, tune) == 2 &&
          (!has_mul || inverse) && npolys % L == 0 &&
          bytes >= (tune.pipelined_min_mb ? (u64)tune.pipelined_min_mb << 20
                                           : pm == kArithB32 ? 4 * kPipelinedMinBytes : kPipelinedMinBytes)))
        return 0;
    // (u32 tables: tiles of 512 MiB — 6 tiles 2.462 ms, 12 tiles 2.494 ms, 24 tiles 2.586 ms per 3 GiB)
    const u64 tile_bytes = pm == kArithB32 ? 2 * kPipelinedTileBytes : kPipelinedTileBytes;
    int pt = tune.pipe_tiles ? tune.pipe_tiles
                                : (int)std::min<u64>((bytes + tile_bytes / 2) / tile_bytes, (u64)kPipelinedMaxTiles);
    if (pt < 2) pt = 2;
    if (pt > kPipelinedMaxTiles) pt = kPipelinedMaxTiles;  // what transform_pipelined runs (its launch count is reported)
    if ((u64)pt > npolys / L) pt = (int)(npolys / L);
    return pt;
}

// how transform() will run a batch: kernel (or form) name and the number of kernel launches
int ntt_transform_form(u32 L, u32 log_n, int arith, u64 npolys, bool inverse, const NttTuning &tune, char *buf, size_t cap) {
    const int pt = pipelined_tiles(L, log_n, arith, npolys, inverse, false, tune);
    if (pt >= 1) {
        std::snprintf(buf, cap, inverse ? "ntt_pipe_inv_kernel" : "ntt_pipe_fwd_kernel");
        return pt + 1;
    }
    // one launch per pass: name them in execution order
    const NttPlan plan = make_ntt_plan(log_n, arith, tune);
    const int passes = plan.tiny ? 1 : plan.n_strided + 1;
    size_t at = 0;
    buf[0] = 0;
    for (int i = 0; i < passes && at + 1 < cap; ++i) {
        char one[96];
        ntt_pass_name(log_n, inverse, i, one, sizeof one, arith, tune);
        if (!plan.tiny && plan.n_strided == 0 && plan.block_log == 14 && arith != kArithB32 &&
            npolys >= 2 * (u64)device_cu_count())
            std::snprintf(one, sizeof one, "ntt_persist_kernel<14,%s>", inverse ? "inv" : "fwd");  // launch_block's choice
        at += (size_t)std::snprintf(buf + at, cap - at, "%s%s", i ? " + " : "", one);
    }
    return passes;
}

static int transform(const NttPrime *primes, u32 L, u32 log_n, int pm, u64 *data, u64 npolys, bool inverse,
                     bool lazy, hipStream_t s, const NttTuning &tune, const u64 *mul = nullptr, u64 mul_polys = 0) {
    const int passes = ntt_num_passes(log_n, pm, tune);
    {
        const int pt = pipelined_tiles(L, log_n, pm, npolys, inverse, mul != nullptr, tune);
        if (pt >= 1 && pm == kArithB32)
            return transform_pipelined<B32Arith, 11>(primes, L, data, npolys, inverse, lazy, s, pt, mul, mul_polys);
        if (pt >= 1 && pm == kArithMont)
            return transform_pipelined<MontArith, 12>(primes, L, data, npolys, inverse, lazy, s, pt, mul, mul_polys);
        if (pt >= 1)
            return pm == kArithPm
                       ? transform_pipelined<PmArith, 12>(primes, L, data, npolys, inverse, lazy, s, pt, mul, mul_polys)
                       : transform_pipelined<ShoupArith, 12>(primes, L, data, npolys, inverse, lazy, s, pt, mul, mul_polys);
    }
    // one launch per pass on the caller's stream
    for (int i = 0; i < passes; ++i)
        PFHE_TRY(ntt_pass_dev(primes, L, log_n, pm, data, npolys, inverse, i, lazy, s, i == 0 ? mul : nullptr, mul_polys, tune));
    return PFHE_OK;
}

int ntt_forward_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, bool lazy,
                    hipStream_t s, const NttTuning &tune) {
    return transform(primes, L, log_n, arith, data, npolys, false, lazy, s, tune);
}

int ntt_inverse_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, bool lazy,
                    hipStream_t s, const NttTuning &tune) {
    return transform(primes, L, log_n, arith, data, npolys, true, lazy, s, tune);
}

// Transform of a batch that lives in device-visible HOST memory (`io`: pinned / registered), every word crossing the link
// once in each direction and no copy engine involved: single-pass rings run their in-place kernel on `io` itself; two-pass
// rings (one strided pass + blocks of 2^12) read `io` in their first pass, keep the intermediate in `scratch` (device,
// same size) and write `io` in their second.  PFHE_ERR_UNSUPPORTED: shape not covered (the caller copies instead).
template <class A>
static int through_dev_two_pass(const NttPlan &plan, const NttPrime *primes, u32 L, u32 log_n, u64 *io, u64 *scratch, u64 npolys,
                                bool inverse, bool lazy, hipStream_t s) {
    const int k = plan.strided[0];
    const u32 log_s = log_n - k;
    const u64 sthreads = (npolys << (log_n - k)) >> (k == 5 ? 0 : 1), sgrid = (sthreads + 255) / 256;  // K = 5: one column per thread
    const u64 blocks = npolys << (log_n - 12);
    if (sgrid > 0x7fffffffull || blocks > 0x7fffffffull) return PFHE_ERR_UNSUPPORTED;
    constexpr size_t lds_bytes = (size_t)BlockCfg<12>::LDS_WORDS * sizeof(u64);
    const auto strided = [&](auto kc, u64 *dst, const u64 *src) {
        constexpr int K = decltype(kc)::value, VEC = K == 5 ? 1 : 2;
        if (!inverse)
            hipLaunchKernelGGL((ntt_strided_oop_kernel<A, K, VEC, false, false>), dim3((u32)sgrid), dim3(256), 0, s, dst, src, primes,
                               L, log_n, log_s, sthreads, 0u);
        else
            hipLaunchKernelGGL((ntt_strided_oop_kernel<A, K, VEC, true, true>), dim3((u32)sgrid), dim3(256), 0, s, dst, src, primes,
                               L, log_n, log_s, sthreads, lazy ? 1u : 0u);
    };
    const auto strided_k = [&](u64 *dst, const u64 *src) {
        if (k == 3) strided(std::integral_constant<int, 3>{}, dst, src);
        else if (k == 4) strided(std::integral_constant<int, 4>{}, dst, src);
        else strided(std::integral_constant<int, 5>{}, dst, src);
    };
    if (!inverse) {
        strided_k(scratch, io);
        hipLaunchKernelGGL((ntt_block_oop_kernel<A, false>), dim3((u32)blocks), dim3(BlockCfg<12>::THREADS), lds_bytes, s, io,
                           scratch, primes, L, log_n, blocks, lazy ? 1u : 0u);
    } else {
        hipLaunchKernelGGL((ntt_block_oop_kernel<A, true>), dim3((u32)blocks), dim3(BlockCfg<12>::THREADS), lds_bytes, s, scratch,
                           io, primes, L, log_n, blocks, 0u);
        strided_k(io, scratch);
    }
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int ntt_transform_through_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *io, u64 *scratch, u64 npolys,
                              bool inverse, bool lazy, hipStream_t s, const NttTuning &tune) {
    if (arith == kArithB32) return PFHE_ERR_UNSUPPORTED;
    const NttPlan plan = make_ntt_plan(log_n, arith, tune);
    if (plan.tiny || plan.n_strided == 0)
        return inverse ? ntt_inverse_dev(primes, L, log_n, arith, io, npolys, lazy, s, tune)
                       : ntt_forward_dev(primes, L, log_n, arith, io, npolys, lazy, s, tune);
    if (plan.n_strided != 1 || plan.block_log != 12 || plan.strided[0] < 3 || plan.strided[0] > 5 || scratch == nullptr)
        return PFHE_ERR_UNSUPPORTED;
    if (arith == kArithMont) return through_dev_two_pass<MontArith>(plan, primes, L, log_n, io, scratch, npolys, inverse, lazy, s);
    return arith == kArithPm ? through_dev_two_pass<PmArith>(plan, primes, L, log_n, io, scratch, npolys, inverse, lazy, s)
                             : through_dev_two_pass<ShoupArith>(plan, primes, L, log_n, io, scratch, npolys, inverse, lazy, s);
}

int ntt_inverse_mul_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, const u64 *mul,
                        u64 mul_polys, hipStream_t s, const NttTuning &tune) {
    if (mul == nullptr || mul_polys == 0 || mul_polys % L != 0 || (mul_polys != npolys && mul_polys != L))
        return PFHE_ERR_BAD_ARGUMENT;
    return transform(primes, L, log_n, arith, data, npolys, true, false, s, tune, mul, mul_polys);
}


// NTT -> product -> INTT with the middle kernel (see block_mid_body).  PFHE_ERR_UNSUPPORTED when the shape does not
// take it (the caller falls back to transform + fused inverse-mul).
template <class A>
static int polymul_impl(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, const u64 *mul,
                        u64 mul_polys, hipStream_t s, const NttTuning &tune) {
    const NttPlan plan = make_ntt_plan(log_n, arith, tune);
    if (plan.tiny || plan.n_strided > 1 || plan.block_log < 10) return PFHE_ERR_UNSUPPORTED;
    const bool large = (npolys << log_n) * sizeof(u64) >= kNtMinBytes;
    const auto launch_mid = [&](auto logb_c, u64 *ptr, u64 np, const u64 *mptr, u64 mp) -> int {
        constexpr int LOGB = decltype(logb_c)::value;
        using Cfg = BlockCfg<LOGB>;
        const u64 total_blocks = np << (log_n - LOGB);
        if (total_blocks == 0) return PFHE_OK;
        if (total_blocks > 0x7fffffffull) {
            set_last_error("batch too large for one launch");
            return PFHE_ERR_BAD_LENGTH;
        }
        constexpr size_t lds_bytes = (size_t)Cfg::LDS_WORDS * sizeof(u64);
        // a per-element multiplicand of a large batch is read once (non-temporal); a shared one stays cacheable
        const int form = !large ? 0 : (mp == np ? 2 : 1);
        void (*kern)(u64 *, const NttPrime *, u32, u32, u64, const u64 *, u64) =
            form == 0 ? ntt_block_mid_kernel<A, LOGB, false> : form == 1 ? ntt_block_mid_kernel<A, LOGB, true, false>
                                                                          : ntt_block_mid_kernel<A, LOGB, true, true>;
        if (lds_bytes > 64 * 1024) {  // once per device and instantiation
            static thread_local bool configured[64][3] = {};
            int dev = 0;
            PFHE_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !configured[dev][form]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute", __FILE__, __LINE__);
                if (dev >= 0 && dev < 64) configured[dev][form] = true;
            }
        }
        hipLaunchKernelGGL(kern, dim3((u32)total_blocks), dim3(Cfg::THREADS), lds_bytes, s, ptr, primes, L, log_n,
                           total_blocks, mptr, mp);
        PFHE_HIP(hipGetLastError());
        return PFHE_OK;
    };
    if (plan.n_strided == 0 && plan.block_log == 14) {
        // N = 2^14: resident workgroups that prefetch their next polynomial (ntt_persist_mid_kernel)
        const u64 resident = (u64)device_cu_count();
        if (npolys >= 2 * resident) {
            using Cfg = BlockCfg<14>;
            constexpr size_t lds_bytes = (size_t)Cfg::LDS_WORDS * sizeof(u64);
            void (*kern)(u64 *, const NttPrime *, u32, u64, const u64 *, u64) = ntt_persist_mid_kernel<A, 14>;
            static thread_local bool configured[64] = {};
            int dev = 0;
            PFHE_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !configured[dev]) {
                PFHE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds_bytes));
                if (dev >= 0 && dev < 64) configured[dev] = true;
            }
            const u64 rounds = (npolys + resident - 1) / resident;
            const u64 grid = (npolys + rounds - 1) / rounds;
            hipLaunchKernelGGL(kern, dim3((u32)grid), dim3(Cfg::THREADS), lds_bytes, s, data, primes, L, npolys, mul, mul_polys);
            PFHE_HIP(hipGetLastError());
            return PFHE_OK;
        }
    }
    if (plan.n_strided == 0) {  // single-pass rings: the middle kernel is the whole product
        switch (plan.block_log) {
#define PFHE_CASE(B) \
    case B: return launch_mid(std::integral_constant<int, B>{}, data, npolys, mul, mul_polys);
            PFHE_CASE(10) PFHE_CASE(11) PFHE_CASE(12) PFHE_CASE(13) PFHE_CASE(14)
#undef PFHE_CASE
        }
        return PFHE_ERR_UNSUPPORTED;
    }
    if (plan.block_log != 12) return PFHE_ERR_UNSUPPORTED;
    const int pt = pipelined_tiles(L, log_n, arith, npolys, true, true, tune);
    if (pt >= 1) {
        constexpr size_t lds_bytes = (size_t)BlockCfg<12>::LDS_WORDS * sizeof(u64);
        const int tiles = pt > kPipelinedMaxTiles ? kPipelinedMaxTiles : pt;
        const u64 units = npolys / L;
        for (int k = 0; k <= tiles + 1; ++k) {
            // launch k: forward strided pass of tile k, middle kernel on tile k-1, inverse strided pass of tile k-2
            u64 *ptr[3] = {nullptr, nullptr, nullptr};
            u64 tot[3] = {0, 0, 0};
            const u64 *mptr = nullptr;
            u64 mp = 0;
            for (int role = 0; role < 3; ++role) {
                const int kt = k - role;
                if (kt < 0 || kt >= tiles) continue;
                const u64 u0 = units * (u64)kt / (u64)tiles, u1 = units * (u64)(kt + 1) / (u64)tiles;
                ptr[role] = data + ((u0 * L) << log_n);
                tot[role] = ((u1 - u0) * L) << 4;
                if (role == 1) {  // a per-element multiplicand is tiled like the data; a shared one (one unit of L) is not
                    mptr = mul_polys == npolys ? mul + ((u0 * L) << log_n) : mul;
                    mp = mul_polys == npolys ? (u1 - u0) * L : mul_polys;
                }
            }
            const u64 grid = std::max(tot[0], std::max(tot[1], tot[2]));
            if (grid == 0) continue;
            if (grid > 0x7fffffffull) {
                set_last_error("batch too large for one launch");
                return PFHE_ERR_BAD_LENGTH;
            }
            hipLaunchKernelGGL((ntt_pipe_mid_kernel<A, 12>), dim3((u32)grid), dim3(BlockCfg<12>::THREADS), lds_bytes, s, ptr[1],
                               tot[1], ptr[0], tot[0], ptr[2], tot[2], primes, L, mptr, mp);
            PFHE_HIP(hipGetLastError());
        }
        return PFHE_OK;
    }
    PFHE_TRY(ntt_pass_dev(primes, L, log_n, arith, data, npolys, false, 0, false, s, nullptr, 0, tune));
    PFHE_TRY(launch_mid(std::integral_constant<int, 12>{}, data, npolys, mul, mul_polys));
    return ntt_pass_dev(primes, L, log_n, arith, data, npolys, true, 1, false, s, nullptr, 0, tune);
}

int ntt_polymul_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u64 *data, u64 npolys, const u64 *mul,
                    u64 mul_polys, hipStream_t s, const NttTuning &tune) {
    if (mul == nullptr || mul_polys == 0 || mul_polys % L != 0 || (mul_polys != npolys && mul_polys != L))
        return PFHE_ERR_BAD_ARGUMENT;
    if (arith == kArithB32) return PFHE_ERR_UNSUPPORTED;
    if (npolys == 0) return PFHE_OK;
    if (arith == kArithMont) return polymul_impl<MontArith>(primes, L, log_n, arith, data, npolys, mul, mul_polys, s, tune);
    return arith == kArithPm ? polymul_impl<PmArith>(primes, L, log_n, arith, data, npolys, mul, mul_polys, s, tune)
                             : polymul_impl<ShoupArith>(primes, L, log_n, arith, data, npolys, mul, mul_polys, s, tune);
}

// ------------------------------------------------------------------------------------------
// u32 tables (U32NttTable, prime32/table.rs): N <= 16 runs the reference's loop nest with one
// thread per polynomial (scalar/transform.rs:13-273); larger N view the data as N/2 64-bit words
// (two adjacent coefficients each) and run the word kernels above with B32Arith.
// ------------------------------------------------------------------------------------------
template <bool INV>
__global__ void ntt32_tiny_kernel(u32 *__restrict__ data, const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                  u64 npolys, u32 lazy) {
    u64 pid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= npolys) return;
    const B32Arith ar(primes + pid % L);
    const u32 n = 1u << log_n, two_q = ar.two_q32;
    u32 *x = data + pid * n;
    if (!INV) {
        for (u32 p = log_n; p-- > 0;) {
            for (u32 e = 0; e < n; ++e) {
                if (e & (1u << p)) continue;
                const u32 f = e | (1u << p);
                const B32Arith::Tw wn = ar.fwd_tw((n + e) >> (p + 1));  // negated twiddle (NttPrime::fwd_wn)
                const u32 tx = B32Arith::once(x[e], two_q), tn = ar.mul1_neg(x[f], wn);
                x[e] = tx - tn;
                x[f] = tx + two_q + tn;
            }
        }
        if (!lazy)
            for (u32 e = 0; e < n; ++e) x[e] = B32Arith::once(B32Arith::once(x[e], two_q), ar.q);
    } else {
        const GCWordPtr inv = ar.inv - (n >> 1);  // NttPrime::inv_w is biased by N/2 entries
        for (u32 p = 0; p + 1 < log_n; ++p) {
            for (u32 e = 0; e < n; ++e) {
                if (e & (1u << p)) continue;
                const u32 f = e | (1u << p);
                const B32Arith::Tw w = B32Arith::unpack(inv[1 + n - (n >> p) + (e >> (p + 1))]);
                const u32 a = x[e], b = x[f];
                x[e] = B32Arith::once(a + b, two_q);
                x[f] = ar.mul1(a + two_q - b, w);
            }
        }
        const u32 h = n >> 1;
        for (u32 e = 0; e < h; ++e) {  // scalar/transform.rs:253-271
            const u32 a = x[e], b = x[e + h];
            u32 rx = ar.mul1(B32Arith::once(a + b, two_q), ar.inv_n);
            u32 ry = ar.mul1(a + two_q - b, ar.inv_n_w);
            if (!lazy) {
                rx = B32Arith::once(rx, ar.q);
                ry = B32Arith::once(ry, ar.q);
            }
            x[e] = rx;
            x[e + h] = ry;
        }
    }
}

int ntt32_transform_dev(const NttPrime *primes, u32 L, u32 log_n, u32 *data, u64 npolys, bool inverse, bool lazy,
                        hipStream_t s, const NttTuning &tune) {
    if (log_n == 0 || npolys == 0) return PFHE_OK;  // N = 1: the reference's loops do not execute
    if (log_n <= 4) {
        const dim3 g((u32)((npolys + 255) / 256)), t(256);
        if (inverse) hipLaunchKernelGGL(ntt32_tiny_kernel<true>, g, t, 0, s, data, primes, L, log_n, npolys, lazy ? 1u : 0u);
        else hipLaunchKernelGGL(ntt32_tiny_kernel<false>, g, t, 0, s, data, primes, L, log_n, npolys, lazy ? 1u : 0u);
        PFHE_HIP(hipGetLastError());
        return PFHE_OK;
    }
    return transform(primes, L, log_n - 1, kArithB32, reinterpret_cast<u64 *>(data), npolys, inverse, lazy, s, tune);
}

}  // namespace pfhe
