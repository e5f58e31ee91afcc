// pfhe_extprod.hip — fused kernel of the RNS gadget external product.
//
// Reference dataflow per digit polynomial (primus_lattice/src/glwe/dcrt.rs:228-254):
//   table.transform_slice(digit)  ->  for every component c: acc_c += key_c * digit_hat
// Here the LAST pass of the forward transform (block pass, pfhe_ntt.hip) and the multiply-accumulate
// are one kernel: a workgroup owns one 2^12-coefficient block of one limb of one ciphertext, loops
// over the rows*ell digit polynomials, transforms each block on chip and accumulates key * digit_hat
// into registers; the transformed digits are never written back (the unfused path writes and
// re-reads 8*(k+1)*ell*L*N bytes per ciphertext).  Accumulation is exact modular arithmetic (canonical
// result), so the value equals the reference's sequence of reduce_mul_add calls.
#include <type_traits>

#include "pfhe_common.hpp"
#include "pfhe_modmath.hpp"
#include "pfhe_ntt_device.hpp"
#include "pfhe_rns.hpp"
#include "pfhe_rns_device.hpp"

namespace pfhe {

namespace {

// modular multiply-accumulate policies matching the NTT arithmetic policies
// acc + d*k without folding: a product term is < 1.5 * 2^K (K <= 61), so an accumulator folded below
// 2^K + 2^(K-9) can take FOUR terms before it must be folded again (1.002 + 4 * 1.5 = 7.002 < 8 = 2^64 / 2^61)
__device__ __forceinline__ u64 mac(const PmArith &ar, u64 acc, u64 d, u64 k) {
    return acc + ar.mul_full(d, k);
}
constexpr u32 kPmMacFoldEvery = 4;

struct BarrettMac {
    u64 q, lo, hi;
};
__device__ __forceinline__ u64 mac(const BarrettMac &m, u64 acc, u64 d, u64 k) {
    return mul_add_mod_barrett(d, k, acc, m.q, m.lo, m.hi);
}

constexpr int kMulaccMinWg = 2;  // resident workgroups per CU the 256-thread form's register allocation is sized for
// LOGE: 4 = 256 threads with 16 coefficients each, 3 = 512 threads with 8 each.  The accumulators (NC 64-bit words per
// coefficient) live in registers for the whole loop over the terms: with 16 coefficients per thread they are 64 registers
// on top of the transform's ~106 and only two workgroups (two waves per SIMD) fit a CU; with 8 they are 32 on top of ~80.
constexpr int kMulacc8MinWaves = 4;
template <class A, int NC, int LOGE = 4>
__global__ __launch_bounds__(LOGE == 3 ? 512 : 256, LOGE == 3 ? kMulacc8MinWaves : kMulaccMinWg) void gadget_block_mulacc_kernel(const u64 *__restrict__ digits,
                                                                  const u64 *__restrict__ ggsw, u64 ggsw_stride,
                                                                  u64 *__restrict__ result,
                                                                  const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                                  u32 terms, u64 total_blocks, u32 accumulate, u32 inv_tail) {
    constexpr int LOGB = 12;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 lt = threadIdx.x;
    const u64 blk = blockIdx.x;
    if (blk >= total_blocks) return;
    const u32 log_nb = log_n - LOGB;
    const u32 n = 1u << log_n;
    // blk -> (ciphertext e, limb r, block bi).  Workgroups are dealt round-robin over the 8 XCDs (blk and blk + 8 share
    // one), and every workgroup reads the 24 key blocks of its (limb, block) pair: 768 KiB that only the workgroups of the
    // same pair share.  With blk = (e, r, bi) in that order an XCD cycles through 3 * 16 / 8 = 6 pairs, 4.6 MiB of key
    // against its 4 MiB L2, and re-fetches them from the fabric about 24 times per launch (FETCH_SIZE 3.15 GiB for 2.25 GiB
    // of digits).  XCD-aware order instead: XCD x keeps the pairs whose block index is x (mod 8) and walks ALL ciphertexts
    // of one pair before it turns to the next, so the 64 workgroups resident on it read the same 768 KiB.
    const u32 nb = 1u << log_nb;
    u32 bi, r;
    u64 e;
    if (nb >= 8) {
        const u64 batch = total_blocks / ((u64)L << log_nb);
        const u32 xcd = (u32)(blk & 7), per = nb >> 3;  // pairs of one limb on this XCD
        const u64 i = blk >> 3;
        const u32 pr = (u32)(i / batch);
        e = i - (u64)pr * batch;
        r = pr / per;
        bi = xcd + 8u * (pr - r * per);
    } else {
        bi = (u32)(blk & (nb - 1));
        const u64 er = blk >> log_nb;
        r = (u32)(er % L);
        e = er / L;
    }
    const NttPrime *__restrict__ P = primes + r;
    const A ar(P);
    const u32 eblk = bi << LOGB;
    const u64 W = (u64)L << log_n;
    const u64 limb_off = ((u64)r << log_n) + eblk;
    const u64 *__restrict__ dg = digits + e * terms * W + limb_off;
    const u64 *__restrict__ key = ggsw + e * ggsw_stride + limb_off;
    u64 *__restrict__ out = result + e * NC * W + limb_off;

    constexpr int NV = 1 << (LOGE - 1);  // 16-byte vectors per thread
    // (accumulators in the layout the transform ends in, the key read as each thread's own run of 16-byte pieces and ONE
    // transposition per output at the end instead of one per term: 20.1 against 19.5 ms per 1024 products — the key
    // reads, 64 pieces 64 bytes apart per wave instruction, cost more than the LDS trips they save)
    // The accumulators start at zero; an accumulating call adds the previous result in the epilogue.  (Loading it in
    // front of the loop made the compiler park all 32 accumulator registers in scratch on the way into the loop — one
    // 148-byte store per lane, 432 MiB per launch of 128 ciphertexts, reloaded only on the never-taken zero-terms path.)
    u64x2 acc[NC][NV];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[c][j] = u64x2{0, 0};

    u32 ij = 0;
    do {  // terms >= 1 (checked on the host): no zero-trip path to keep the initial values alive for
        u64x2 io[NV];
        u64 x[1 << LOGE];
        // Per-term opaque copy of the thread id: every address of the body (digit and key vectors, LDS layouts, the
        // lane-ordered twiddle entries) is recomputed from it each term — a few dozen instructions against the ~1500 of
        // a term — instead of being hoisted out of the loop as an invariant, where the 64-bit addresses alone held 30
        // registers across the loop and pushed 17 into scratch.
        u32 ltl = lt;
        asm volatile("" : "+v"(ltl));
        // the first register pass wants register k = element (k << POS0) + lt: 8-byte loads deliver it without staging
        // (512 contiguous bytes per wave instruction; 19.4-19.6 against 19.9 ms per 1024 products with staged 16-byte
        // loads); the first exchange then needs its leading barrier, because other threads may still be reading the
        // previous term's natural-order image
#pragma unroll
        for (int k = 0; k < (1 << LOGE); ++k)
            x[k] = __builtin_nontemporal_load(dg + (u64)ij * W + ((u32)k << (LOGB - LOGE)) + ltl);
        block_forward_core<A, LOGB, true, LOGE, NoLateHook, true>(ar, x, lds, n, eblk, ltl, /*lazy=*/true);  // digit_hat mod~ q, raw
        lds_put_layout<0, LOGE>(x, lds, ltl);
        sync_vectors_layout0<LOGB, LOGE>();  // wave-local transposition (vec_index): no workgroup barrier
        lds_get_vectors<LOGB, LOGE>(io, lds, ltl);  // natural order again: same positions as the key vectors
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            u64x2 kv[NV];
            load_block_vectors<LOGB, LOGE>(kv, key + ((u64)ij * NC + c) * W, ltl);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (std::is_same<A, PmArith>::value) {
                    acc[c][j].x = mac(ar, acc[c][j].x, io[j].x, kv[j].x);
                    acc[c][j].y = mac(ar, acc[c][j].y, io[j].y, kv[j].y);
                    if ((ij % kPmMacFoldEvery) == kPmMacFoldEvery - 1) {
                        acc[c][j].x = ar.reduce_x(acc[c][j].x);
                        acc[c][j].y = ar.reduce_x(acc[c][j].y);
                    }
                } else {
                    // the lazy transform leaves digit_hat in [0,4q): Barrett takes any product < q*2^64
                    const BarrettMac m{P->q, P->bar_lo, P->bar_hi};
                    acc[c][j].x = mac(m, acc[c][j].x, io[j].x, kv[j].x);
                    acc[c][j].y = mac(m, acc[c][j].y, io[j].y, kv[j].y);
                }
            }
        }
    } while (++ij < terms);
    // The epilogue's addresses, too, are computed from an opaque copy of the thread id (one per component): they cannot
    // be hoisted above the loop or above the previous component's inverse pass.
    // explicit instantiation per component (the body may contain a whole inverse block pass, which the compiler
    // does not unroll a loop around; a runtime-indexed accumulator array would live in scratch)
    auto epilogue = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        u64 *__restrict__ oc = out + (u64)c * W;
        u32 lte = lt;
        asm volatile("" : "+v"(lte));
        if (accumulate) {  // DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign: acc += previous result (canonical)
            u64x2 old[NV];
            load_block_vectors<LOGB, LOGE>(old, oc, lte);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (std::is_same<A, PmArith>::value) {
                    // pending terms < 5.5 * 2^K, old < 2^K: the sum fits 64 bits and is folded below
                    acc[c][j].x += old[j].x;
                    acc[c][j].y += old[j].y;
                } else {
                    acc[c][j].x = csub(acc[c][j].x + old[j].x, P->q);
                    acc[c][j].y = csub(acc[c][j].y + old[j].y, P->q);
                }
            }
        }
        if (inv_tail) {
            // DcrtGlwe::into_coeff_form (macros/mod.rs:901-911), first pass: the inverse transform's block pass runs on the
            // accumulators while they are on chip; what is stored is the intermediate the inverse strided pass expects
            // (exactly what ntt_block_kernel<12, inverse> would have left in place).  Its butterflies take any
            // representative below 3 * 2^K, so the fold alone is enough.
            if constexpr (std::is_same<A, PmArith>::value) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    acc[c][j].x = ar.reduce_x(acc[c][j].x);
                    acc[c][j].y = ar.reduce_x(acc[c][j].y);
                }
            }
            u64 y[1 << LOGE];
            __syncthreads();  // other threads may still be reading the previous image (last term / previous component)
            lds_put_vectors<LOGB, LOGE>(acc[c], lds, lte);
            sync_vectors_layout0<LOGB, LOGE>();  // wave-local (vec_index)
            lds_get_layout<0, LOGE>(y, lds, lte);
            block_inverse_core<A, LOGB, false, LOGE>(ar, y, lds, n, eblk, lte, /*final_block=*/false, /*lazy=*/false);
#pragma unroll
            for (int k = 0; k < (1 << LOGE); ++k) gstore<true>(oc + ((u32)k << (LOGB - LOGE)) + lte, y[k]);
        } else {
            if constexpr (std::is_same<A, PmArith>::value) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {  // fold the pending terms, then [0, 2^K + 2^31) -> canonical
                    acc[c][j].x = csub(ar.reduce_x(acc[c][j].x), ar.q);
                    acc[c][j].y = csub(ar.reduce_x(acc[c][j].y), ar.q);
                }
            }
            store_block_vectors<LOGB, LOGE>(acc[c], oc, lte);
        }
    };
    epilogue(std::integral_constant<int, 0>{});
    if constexpr (NC > 1) epilogue(std::integral_constant<int, 1>{});
    if constexpr (NC > 2) epilogue(std::integral_constant<int, 2>{});
    static_assert(NC <= 3, "add an epilogue call per component");
}

// ------------------------------------------------------------------------------------------
// Steps (1)-(4) of glwe/dcrt.rs:219-244 fused with the FIRST (strided) pass of the forward transform, as two kernels:
//   gadget_signed_digits_kernel: steps (1)-(3), two coefficients per thread: CRT-compose, carry init, all ell BALANCED
//     digits written as DT = int32 (log_basis <= 31) or int64 ([poly][level][N]: 4*ell*N or 8*ell*N bytes per input
//     polynomial — the centred lift (4) of a signed digit d is d or q_i + d, so the limb copies are not materialised);
//   digits_strided_kernel: step (4) + the strided pass of (5): a thread owns one column of one (polynomial, level),
//     reads its 2^K digits once, and for every limb lifts them, runs the K stages in registers and stores where the
//     strided pass would have stored.  The coefficient-domain digit polynomials (8*ell*L*N bytes per input polynomial)
//     are never written.
// (Until round 3 digits wider than 32 bits took a single kernel that kept 2^K composed big integers per thread alive
// across all levels: 234 VGPRs at 3 limbs and 528 bytes of scratch per lane at 4.  The digit width is now a template
// parameter of the two kernels and that kernel is gone.)
// ------------------------------------------------------------------------------------------
// CPT coefficients per thread.  Two adjacent ones for big integers of at most 4 limbs held by value (16-byte loads of
// the residues, one store of the digit pair: every BASELINE config); one for longer integers and for the device-table
// form of a wide base (RT = RnsWide, BT = BasisDev or BasisWide), whose residues are fetched as the lift consumes them.
template <int LEN, class DT, class RT, class BT, int CPT, class WT = u64>
__global__ __launch_bounds__(256) void gadget_signed_digits_kernel(RT R, BT B, u32 log_n,
                                                                  const WT *__restrict__ crt, DT *__restrict__ out,
                                                                  u64 total_threads) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_threads) return;
    const u32 n = 1u << log_n;
    const u64 poly = (gid * CPT) >> log_n;
    const u32 t = (u32)((gid * CPT) & (n - 1));
    u64 v[CPT][LEN];
    if (R.big_input) {  // BigUintPolynomial input (glwe/dcrt.rs:258-338): already composed
        const u32 vw = words_of<LEN, WT>(R);
#pragma unroll
        for (int e = 0; e < CPT; ++e) load_limbs<LEN, WT>(crt + (poly * n + t + e) * vw, vw, v[e]);
    } else if constexpr (kByValue<RT>) {
        u64 r[CPT][kMaxLimbs];
        for (u32 i = 0; i < R.L; ++i) {
            if constexpr (CPT == 2) {  // two adjacent residues in one load (16 bytes of u64 words, 8 bytes of u32 words), read once
                typedef WT WT2 __attribute__((ext_vector_type(2)));
                const WT2 w = __builtin_nontemporal_load(reinterpret_cast<const WT2 *>(crt + (poly * R.L + i) * n + t));
                r[0][i] = w.x;
                r[1][i] = w.y;
            } else {
                r[0][i] = __builtin_nontemporal_load(crt + (poly * R.L + i) * n + t);
            }
        }
#pragma unroll
        for (int e = 0; e < CPT; ++e) compose<LEN>(R, r[e], v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < CPT; ++e)
            compose_general<LEN>(R, [&](u32 i) { return (u64)__builtin_nontemporal_load(crt + (poly * R.L + i) * n + t + e); }, v[e]);
    }
    u32 carry[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) carry[e] = init_value_carry<LEN>(B, v[e]);
    const u64 half = (B.basis + 1) / 2;
    DT *__restrict__ o = out + poly * B.ell * n + t;
    typedef DT DT2 __attribute__((ext_vector_type(2)));
    for (u32 j = 0; j < B.ell; ++j) {
        DT d[CPT];
#pragma unroll
        for (int e = 0; e < CPT; ++e) {
            const u64 temp = window<LEN>(v[e], B.drop_bits + j * B.log_basis, B.basis_minus_one, B.log_basis) + carry[e];
            carry[e] = (temp & B.carry_mask) != 0;
            const u64 u = temp & B.basis_minus_one;
            d[e] = (B.basis != 2 && u >= half) ? (DT)((long long)u - (long long)B.basis) : (DT)u;
        }
        if constexpr (CPT == 2) *reinterpret_cast<DT2 *>(o + (u64)j * n) = DT2{d[0], d[1]};
        else o[(u64)j * n] = d[0];
    }
}

// steps (1)-(3) for `npolys` polynomials of 2^log_n coefficients (log_n >= 1), any base the handles accept
template <class DT, class WT = u64>
struct SignedDigitsLaunch {
    template <int LEN>
    struct At {
        static int run(const RnsParams &r, const BasisParams &b, u32 log_n, const WT *crt, DT *sdigits, u64 npolys, hipStream_t s) {
            const u64 coeffs = npolys << log_n;
            constexpr int CPT = LEN <= 4 ? 2 : 1;
            if (b.wide()) {
                hipLaunchKernelGGL((gadget_signed_digits_kernel<LEN, DT, RnsWide, BasisWide, 1, WT>), dim3((u32)((coeffs + 255) / 256)), dim3(256), 0, s,
                                   r.wide_tab, b.wide_tab, log_n, crt, sdigits, coeffs);
            } else if constexpr (LEN <= kMaxLimbs) {
                if (r.wide()) {
                    hipLaunchKernelGGL((gadget_signed_digits_kernel<LEN, DT, RnsWide, BasisDev, 1, WT>), dim3((u32)((coeffs + 255) / 256)), dim3(256), 0, s,
                                       r.wide_tab, b.dev, log_n, crt, sdigits, coeffs);
                } else {
                    const u64 threads = coeffs / CPT;
                    hipLaunchKernelGGL((gadget_signed_digits_kernel<LEN, DT, RnsDev, BasisDev, CPT, WT>), dim3((u32)((threads + 255) / 256)), dim3(256), 0, s,
                                       r.dev, b.dev, log_n, crt, sdigits, threads);
                }
            }
            return PFHE_OK;
        }
    };
};

template <class A, int K, class DT = int>
__global__ __launch_bounds__(256, 4) void digits_strided_kernel(const NttPrime *__restrict__ primes, u32 L, u32 log_n,
                                                               const DT *__restrict__ dig, u64 *__restrict__ out,
                                                               u64 total_threads) {
    constexpr int RK = 1 << K;
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_threads) return;
    const u32 log_s = log_n - K;
    const u32 n = 1u << log_n;
    const u32 col = (u32)(gid & ((1ull << log_s) - 1));
    const u64 pl = gid >> log_s;  // (input polynomial, level)
    const DT *__restrict__ src = dig + pl * n + col;
    DT d[RK];
#pragma unroll
    for (int k = 0; k < RK; ++k) d[k] = __builtin_nontemporal_load(src + ((u64)k << log_s));  // read once
#pragma unroll 1
    for (u32 i = 0; i < L; ++i) {
        const A ar(primes + i);
        u64 x[RK][1];
        // centred lift (base.rs:279-312): d >= 0 -> d, d < 0 -> q_i - |d|
#pragma unroll
        for (int k = 0; k < RK; ++k) x[k][0] = d[k] < 0 ? ar.q + (u64)(long long)d[k] : (u64)d[k];
        strided_forward_regs<A, K, 1, true>(ar, x, n, 0u, log_s);  // log_s + K = log_n: the transform's first stage is here
        u64 *__restrict__ dst = out + (pl * L + i) * n + col;
#pragma unroll
        for (int k = 0; k < RK; ++k) gstore<true>(dst + ((u64)k << log_s), x[k][0]);
    }
}

template <class A, int K, class DT>
int launch_digits_strided(const RnsParams &r, const BasisParams &b, const NttPrime *primes, u32 log_n, const u64 *crt,
                          DT *sdigits, u64 *digits, u64 npolys, hipStream_t s) {
    PFHE_TRY((dispatch_len<SignedDigitsLaunch<DT>::template At>(r.dev.value_len, r, b, log_n, crt, sdigits, npolys, s)));
    PFHE_HIP(hipGetLastError());
    const u64 total = (npolys * b.dev.ell) << (log_n - K);
    hipLaunchKernelGGL((digits_strided_kernel<A, K, DT>), dim3((u32)((total + 255) / 256)), dim3(256), 0, s, primes, r.dev.L,
                       log_n, (const DT *)sdigits, digits, total);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// ------------------------------------------------------------------------------------------
// The <u32> product (CrtGlwe<u32> x DcrtGgsw over U32DcrtTable) at N = 2^16, k = 1: the same two fused kernels on
// B32Arith, where a 64-bit word carries two adjacent u32 coefficients and a polynomial is 2^15 words = 2^4 strided
// stages x blocks of 2^11 words (+ the intra-word stage inside the block core).
//   digits_strided32_kernel: a thread owns one word column of one (polynomial, level): reads its 2^K pairs of balanced
//     int32 digits once and, for every limb, lifts both halves, runs the K strided stages in registers and stores where
//     the forward transform's strided pass would have stored.
//   gadget_block_mulacc32_kernel: a workgroup (256 threads x 8 words) owns one 2^11-word block of one limb of one
//     ciphertext: block pass of every digit polynomial on chip (canonical output), key x digit products accumulated
//     lazily in one 64-bit register per coefficient — ONE v_mad_u64_u32 per coefficient and term (products below 2^60,
//     folded by a single-word Barrett step every 8 terms) — then the fold, optionally the inverse transform's block pass,
//     and the store.  The transformed digits never reach HBM.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256, 4) void digits_strided32_kernel(const NttPrime *__restrict__ primes, u32 L, u32 log_nw,
                                                                 const int *__restrict__ dig, u64 *__restrict__ out,
                                                                 u64 total_threads) {
    constexpr int RK = 1 << K;
    typedef int int2v __attribute__((ext_vector_type(2)));
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_threads) return;
    const u32 log_s = log_nw - K;
    const u32 nw = 1u << log_nw;
    const u32 col = (u32)(gid & ((1ull << log_s) - 1));
    const u64 pl = gid >> log_s;  // (input polynomial, level)
    const int2v *__restrict__ src = reinterpret_cast<const int2v *>(dig) + pl * nw + col;
    int2v d[RK];
#pragma unroll
    for (int k = 0; k < RK; ++k) d[k] = __builtin_nontemporal_load(src + ((u64)k << log_s));  // read once
#pragma unroll 1
    for (u32 i = 0; i < L; ++i) {
        const B32Arith ar(primes + i);
        u64 x[RK][1];
        // centred lift (base.rs:279-312) of both halves: d >= 0 -> d, d < 0 -> q_i - |d|
#pragma unroll
        for (int k = 0; k < RK; ++k)
            x[k][0] = B32Arith::pack((u32)d[k].x + (d[k].x < 0 ? ar.q : 0u), (u32)d[k].y + (d[k].y < 0 ? ar.q : 0u));
        strided_forward_regs<B32Arith, K, 1, true>(ar, x, nw, 0u, log_s);
        u64 *__restrict__ dst = out + (pl * L + i) * nw + col;
#pragma unroll
        for (int k = 0; k < RK; ++k) gstore<true>(dst + ((u64)k << log_s), x[k][0]);
    }
}

// x mod q for any 64-bit x, q < 2^30, bar = floor(2^64 / q): the quotient estimate is exact or one short
__device__ __forceinline__ u32 fold32(u64 x, u32 q, u64 bar) {
    const u64 r = x - mulhi64(x, bar) * q;
    return (u32)(r >= q ? r - q : r);
}

template <int NC>
__global__ __launch_bounds__(256, 4) void gadget_block_mulacc32_kernel(const u64 *__restrict__ digits,
                                                                      const u64 *__restrict__ ggsw, u64 ggsw_stride,
                                                                      u64 *__restrict__ result,
                                                                      const NttPrime *__restrict__ primes, u32 L, u32 log_nw,
                                                                      u32 terms, u64 total_blocks, u32 accumulate,
                                                                      u32 inv_tail) {
    using A = B32Arith;
    constexpr int LOGB = 11, LOGE = 3, E = 1 << LOGE, NV = E / 2;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 lt = threadIdx.x;
    const u64 blk = blockIdx.x;
    if (blk >= total_blocks) return;
    const u32 log_nb = log_nw - LOGB;
    const u32 nw = 1u << log_nw;
    // blk -> (ciphertext e, limb r, block bi), XCD-aware as in gadget_block_mulacc_kernel: XCD x keeps the (limb, block) pairs
    // whose block index is x (mod 8) and walks all ciphertexts of one pair before it turns to the next
    const u32 nb = 1u << log_nb;
    u32 bi, r;
    u64 e;
    if (nb >= 8) {
        const u64 batch = total_blocks / ((u64)L << log_nb);
        const u32 xcd = (u32)(blk & 7), per = nb >> 3;
        const u64 i = blk >> 3;
        const u32 pr = (u32)(i / batch);
        e = i - (u64)pr * batch;
        r = pr / per;
        bi = xcd + 8u * (pr - r * per);
    } else {
        bi = (u32)(blk & (nb - 1));
        const u64 er = blk >> log_nb;
        r = (u32)(er % L);
        e = er / L;
    }
    const NttPrime *__restrict__ P = primes + r;
    const A ar(P);
    const u32 q = ar.q;
    const u64 bar = P->bar_lo;
    const u32 eblk = bi << LOGB;
    const u64 W = (u64)L << log_nw;  // words per RNS polynomial
    const u64 limb_off = ((u64)r << log_nw) + eblk;
    const u64 *__restrict__ dg = digits + e * terms * W + limb_off;
    const u64 *__restrict__ key = ggsw + e * ggsw_stride + limb_off;
    u64 *__restrict__ out = result + e * NC * W + limb_off;

    u64 acc[NC][2 * E];  // one lazy accumulator per coefficient: coefficient 4j + 2h + half of vector j, word h
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int m = 0; m < 2 * E; ++m) acc[c][m] = 0;

    u32 ij = 0, pending = 0;
    do {  // terms >= 1 (checked on the host)
        u64 x[E];
        u64x2 io[NV];
        u32 ltl = lt;
        asm volatile("" : "+v"(ltl));  // per-term addresses are recomputed, not carried around the loop
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = __builtin_nontemporal_load(dg + (u64)ij * W + ((u32)k << (LOGB - LOGE)) + ltl);
        block_forward_core<A, LOGB, true, LOGE>(ar, x, lds, nw, eblk, ltl, /*lazy=*/false);  // canonical digit_hat
        lds_put_layout<0, LOGE>(x, lds, ltl);
        sync_vectors_layout0<LOGB, LOGE>();  // wave-local transposition
        lds_get_vectors<LOGB, LOGE>(io, lds, ltl);  // natural order: same positions as the key vectors
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            u64x2 kv[NV];
            load_block_vectors<LOGB, LOGE>(kv, key + ((u64)ij * NC + c) * W, ltl);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                acc[c][4 * j + 0] += (u64)(u32)io[j].x * (u32)kv[j].x;
                acc[c][4 * j + 1] += (io[j].x >> 32) * (kv[j].x >> 32);
                acc[c][4 * j + 2] += (u64)(u32)io[j].y * (u32)kv[j].y;
                acc[c][4 * j + 3] += (io[j].y >> 32) * (kv[j].y >> 32);
            }
        }
        if (++pending == kFold32Every) {  // fifteen products below 2^60 on top of a folded value: below 2^64 (pfhe_rns.hpp)
            pending = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int m = 0; m < 2 * E; ++m) acc[c][m] = fold32(acc[c][m], q, bar);
        }
    } while (++ij < terms);

    auto epilogue = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        u64 *__restrict__ oc = out + (u64)c * W;
        u32 lte = lt;
        asm volatile("" : "+v"(lte));
        u64x2 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j].x = A::pack(fold32(acc[c][4 * j + 0], q, bar), fold32(acc[c][4 * j + 1], q, bar));
            v[j].y = A::pack(fold32(acc[c][4 * j + 2], q, bar), fold32(acc[c][4 * j + 3], q, bar));
        }
        if (accumulate) {  // DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign: acc += previous result (canonical)
            u64x2 old[NV];
            load_block_vectors<LOGB, LOGE>(old, oc, lte);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v[j].x = ar.reduce_2q(v[j].x + old[j].x);  // both halves below 2q < 2^31: no carry between them
                v[j].y = ar.reduce_2q(v[j].y + old[j].y);
            }
        }
        if (inv_tail) {
            // DcrtGlwe::into_coeff_form, first pass: the inverse transform's block pass on the accumulators while they
            // are on chip; what is stored is what the inverse block pass would have left for the strided pass
            u64 y[E];
            __syncthreads();  // other threads may still be reading the previous image (last term / previous component)
            lds_put_vectors<LOGB, LOGE>(v, lds, lte);
            sync_vectors_layout0<LOGB, LOGE>();
            lds_get_layout<0, LOGE>(y, lds, lte);
            block_inverse_core<A, LOGB, false, LOGE>(ar, y, lds, nw, eblk, lte, /*final_block=*/false, /*lazy=*/false);
#pragma unroll
            for (int k = 0; k < E; ++k) gstore<true>(oc + ((u32)k << (LOGB - LOGE)) + lte, y[k]);
        } else {
            store_block_vectors<LOGB, LOGE>(v, oc, lte);
        }
    };
    epilogue(std::integral_constant<int, 0>{});
    if constexpr (NC > 1) epilogue(std::integral_constant<int, 1>{});
    static_assert(NC <= 2, "add an epilogue call per component");
}

// ------------------------------------------------------------------------------------------
// Small rings (N = 2^10, 2^11: single-block-pass sizes with at least one wave per polynomial): the
// whole product after the digit extraction is ONE kernel.  A workgroup owns one limb of one ciphertext:
// for every row and level it reads the 16 int32 digits of each thread (coalesced, in the register layout
// of the first register pass), lifts them into its limb, runs the complete forward transform on chip,
// and multiply-accumulates the result with the key polynomials of every output component; at the end it
// runs the inverse transform of each accumulator (coefficient-form output) and stores.  Digit polynomials
// and their transforms never exist in HBM: traffic is 4*ell*N bytes of digits per input polynomial,
// 8*N per output polynomial, and the key out of L2.
// ------------------------------------------------------------------------------------------
template <class A, int LOGB, int NC>
__device__ __forceinline__ void extprod_small_body(
    const int *__restrict__ sdigits, const u64 *__restrict__ ggsw, u64 ggsw_stride, u64 *__restrict__ result,
    const NttPrime *__restrict__ primes, u32 L, u32 rows, u32 ell, u64 total, u32 accumulate, u32 into_coeff) {
    using Cfg = BlockCfg<LOGB>;
    static_assert(Cfg::BPW == 1, "one polynomial per workgroup");
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr u32 n = 1u << LOGB;
    const u32 lt = threadIdx.x;
    const u64 er = blockIdx.x;
    if (er >= total) return;
    const u32 r = (u32)(er % L);
    const u64 e = er / L;
    const NttPrime *__restrict__ P = primes + r;
    const A ar(P);
    const u64 W = (u64)L * n;
    u64 *__restrict__ out = result + e * NC * W + (u64)r * n;

    u64 acc[NC][16];  // NTT-domain positions lt*16 .. lt*16+15 (register layout <0>)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (accumulate) {
            const GCVec2Ptr ap = (GCVec2Ptr)(const void *)(out + (u64)c * W + lt * 16);
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const u64x2 t = ap[v];
                acc[c][2 * v] = t.x;
                acc[c][2 * v + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[c][k] = 0;
        }
    }
    const u32 terms = rows * ell;
    for (u32 ij = 0; ij < terms; ++ij) {
        u64 x[16];
        const u32 ltl = opaque_tid();  // per-term addresses are recomputed, not carried around the loop (see gadget_block_mulacc_kernel)
#pragma unroll
        for (int k = 0; k < 16; ++k) {  // centred lift (base.rs:279-312) of the balanced digit
            const int d = (sdigits + e * rows * ell * n + ltl)[(u64)ij * n + (u32)k * Cfg::TPB];
            x[k] = d < 0 ? ar.q + (u64)(long long)d : (u64)d;
        }
        block_forward_core<A, LOGB, true, 4, NoLateHook, true>(ar, x, lds, n, 0u, ltl, /*lazy=*/true);
        const bool fold_now = (ij % kPmMacFoldEvery) == kPmMacFoldEvery - 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const GCVec2Ptr kp = (GCVec2Ptr)(const void *)(ggsw + e * ggsw_stride + (u64)r * n + ltl * 16 + ((u64)ij * NC + c) * W);
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const u64x2 kv = kp[v];
                if constexpr (std::is_same<A, PmArith>::value) {
                    acc[c][2 * v] = mac(ar, acc[c][2 * v], x[2 * v], kv.x);
                    acc[c][2 * v + 1] = mac(ar, acc[c][2 * v + 1], x[2 * v + 1], kv.y);
                    if (fold_now) {
                        acc[c][2 * v] = ar.reduce_x(acc[c][2 * v]);
                        acc[c][2 * v + 1] = ar.reduce_x(acc[c][2 * v + 1]);
                    }
                } else {
                    const BarrettMac m{P->q, P->bar_lo, P->bar_hi};
                    acc[c][2 * v] = mac(m, acc[c][2 * v], x[2 * v], kv.x);
                    acc[c][2 * v + 1] = mac(m, acc[c][2 * v + 1], x[2 * v + 1], kv.y);
                }
            }
        }
    }
    // explicit instantiation per component (a runtime-indexed accumulator array would live in scratch: the
    // compiler does not unroll a loop whose body contains a whole inverse transform)
    auto epilogue = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        u64 y[16];
        const u32 lte = opaque_tid();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (std::is_same<A, PmArith>::value) y[k] = csub(ar.reduce_x(acc[c][k]), ar.q);
            else y[k] = acc[c][k];
        }
        if (into_coeff) {
            block_inverse_core<A, LOGB>(ar, y, lds, n, 0u, lte, /*final_block=*/true, /*lazy=*/false);
            u64 *__restrict__ o = out + (u64)c * W + lte;  // register layout <LOGB-4>: element lt + k * TPB
#pragma unroll
            for (int k = 0; k < 16; ++k) o[(u32)k * Cfg::TPB] = y[k];
        } else {
            const GVec2Ptr o = (GVec2Ptr)(void *)(out + (u64)c * W + lte * 16);
#pragma unroll
            for (int v = 0; v < 8; ++v) o[v] = u64x2{y[2 * v], y[2 * v + 1]};
        }
        __syncthreads();  // the next component reuses the LDS buffer
    };
    epilogue(std::integral_constant<int, 0>{});
    if constexpr (NC > 1) epilogue(std::integral_constant<int, 1>{});
    if constexpr (NC > 2) epilogue(std::integral_constant<int, 2>{});
    static_assert(NC <= 3, "add an epilogue call per component");
}

// (the body runs with the arithmetic's fold-inside form: at 256 registers this kernel has no room for the values the
// 16-instruction butterfly keeps alive in front of its asm block — pfhe_ntt_device.hpp, PmArith::kFoldOutside)
template <class A, int LOGB, int NC>
__global__ __launch_bounds__(BlockCfg<LOGB>::THREADS) __attribute__((amdgpu_waves_per_eu(2, 3))) void extprod_small_kernel(
    const int *__restrict__ sdigits, const u64 *__restrict__ ggsw, u64 ggsw_stride, u64 *__restrict__ result,
    const NttPrime *__restrict__ primes, u32 L, u32 rows, u32 ell, u64 total, u32 accumulate, u32 into_coeff) {
    extprod_small_body<typename FoldInsideOf<A>::type, LOGB, NC>(sdigits, ggsw, ggsw_stride, result, primes, L, rows, ell, total,
                                                                  accumulate, into_coeff);
}

template <class A, int LOGB, int NC>
int launch_extprod_small(const int *sdigits, const u64 *ggsw, u64 stride, u64 *result, const NttPrime *primes, u32 L,
                         u32 rows, u32 ell, u64 batch, bool accumulate, bool into_coeff, hipStream_t s) {
    const u64 total = batch * L;
    if (total == 0) return PFHE_OK;
    if (total > 0x7fffffffull) return PFHE_ERR_BAD_LENGTH;
    constexpr size_t lds_bytes = (size_t)BlockCfg<LOGB>::LDS_WORDS * sizeof(u64);
    hipLaunchKernelGGL((extprod_small_kernel<A, LOGB, NC>), dim3((u32)total), dim3(BlockCfg<LOGB>::THREADS), lds_bytes, s,
                       sdigits, ggsw, stride, result, primes, L, rows, ell, total, accumulate ? 1u : 0u,
                       into_coeff ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

template <class A, int NC>
int dispatch_extprod_small(u32 log_n, const int *sdigits, const u64 *ggsw, u64 stride, u64 *result,
                           const NttPrime *primes, u32 L, u32 rows, u32 ell, u64 batch, bool accumulate, bool into_coeff,
                           hipStream_t s) {
    switch (log_n) {
        case 10: return launch_extprod_small<A, 10, NC>(sdigits, ggsw, stride, result, primes, L, rows, ell, batch, accumulate, into_coeff, s);
        case 11: return launch_extprod_small<A, 11, NC>(sdigits, ggsw, stride, result, primes, L, rows, ell, batch, accumulate, into_coeff, s);
    }
    return PFHE_ERR_UNSUPPORTED;
}

}  // namespace

bool gadget_fused_supported(u32 log_n, u32 k) { return k == 1 && make_ntt_plan(log_n).block_log == 12; }

int gadget_block_mulacc_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u32 k, u32 terms, const u64 *digits,
                            const u64 *ggsw, bool ggsw_shared, u64 *result, u64 batch, bool accumulate,
                            hipStream_t s, bool inv_tail) {
    if (!gadget_fused_supported(log_n, k) || terms == 0 || (inv_tail && accumulate)) return PFHE_ERR_UNSUPPORTED;
    using Cfg = BlockCfg<12>;
    const u64 total_blocks = (batch * L) << (log_n - 12);
    if (total_blocks == 0) return PFHE_OK;
    if (total_blocks > 0x7fffffffull) return PFHE_ERR_BAD_LENGTH;
    const u64 ggsw_words = ((u64)terms * (k + 1) * L) << log_n;
    constexpr size_t lds_bytes = (size_t)Cfg::LDS_WORDS * sizeof(u64);
    const u64 stride = ggsw_shared ? 0ull : ggsw_words;
    if (arith == kArithPm) {
        // 512 threads x 8 coefficients: 32 accumulator registers, four waves per SIMD (the 256 x 16 form holds 64 and
        // fits two: 48.0 -> 49.0 k products/s when it was replaced)
        hipLaunchKernelGGL((gadget_block_mulacc_kernel<PmArith, 2, 3>), dim3((u32)total_blocks), dim3(512), lds_bytes, s,
                           digits, ggsw, stride, result, primes, L, log_n, terms, total_blocks, accumulate ? 1u : 0u,
                           inv_tail ? 1u : 0u);
    } else if (arith == kArithMont) {
        // generic primes below 2^61: the Montgomery-form butterflies (7 multiplies) instead of the Shoup ones (10)
        // (the 512-thread form here too: 38.7 -> 40.0 k products/s for three generic 61-bit primes, same box)
        hipLaunchKernelGGL((gadget_block_mulacc_kernel<MontArith, 2, 3>), dim3((u32)total_blocks), dim3(512), lds_bytes, s,
                           digits, ggsw, stride, result, primes, L, log_n, terms, total_blocks, accumulate ? 1u : 0u,
                           inv_tail ? 1u : 0u);
    } else {
        hipLaunchKernelGGL((gadget_block_mulacc_kernel<ShoupArith, 2>), dim3((u32)total_blocks), dim3(256), lds_bytes, s,
                           digits, ggsw, stride, result, primes, L, log_n, terms, total_blocks, accumulate ? 1u : 0u,
                           inv_tail ? 1u : 0u);
    }
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

}  // namespace pfhe

namespace pfhe {

// decomposition fused with the first strided pass: two-pass transforms whose strided pass has at
// most 4 stages (N = 2^13 * 2^K... i.e. block 2^12 and K <= 4) and big integers of at most 4 limbs
bool gadget_decompose_strided_supported(u32 log_n, u32 value_len) {
    const NttPlan plan = make_ntt_plan(log_n);
    return !plan.tiny && plan.n_strided == 1 && plan.strided[0] >= 3 && plan.strided[0] <= 4 && value_len <= (u32)kMaxWideLimbs;
}

// Small-ring path (extprod_small_kernel): digits that fit int32; enabled where it measured faster than the
// separate kernels at large batches — N = 2^10 (0.84 vs 1.19 ms per 8192 products) and 2^11 (1.88 vs 2.42 ms)
// with k = 1; at N = 2^12 or k = 2 the accumulators leave too few waves per SIMD (4.5 vs 3.7 ms, 2.2 vs 1.8 ms).
bool extprod_small_supported(u32 log_n, u32 k, u32 value_len, u32 log_basis) {
    return log_n >= 10 && log_n <= 11 && k == 1 && value_len <= 4 && log_basis <= 31;
}

// steps (1)-(3) alone: balanced int32 digits of `npolys` CRT polynomials ([poly][level][N])
template <class WT>
int gadget_signed_digits_dev(const RnsParams &r, const BasisParams &b, u32 log_n, const WT *crt_polys, int *sdigits,
                             u64 npolys, hipStream_t s) {
    if ((npolys << log_n) == 0) return PFHE_OK;
    PFHE_TRY((dispatch_len<SignedDigitsLaunch<int, WT>::template At>(r.dev.value_len, r, b, log_n, crt_polys, sdigits, npolys, s)));
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}
template int gadget_signed_digits_dev<u64>(const RnsParams &, const BasisParams &, u32, const u64 *, int *, u64, hipStream_t);
template int gadget_signed_digits_dev<u32>(const RnsParams &, const BasisParams &, u32, const u32 *, int *, u64, hipStream_t);

int extprod_small_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u32 k, u32 rows, u32 ell, const int *sdigits,
                      const u64 *ggsw, bool ggsw_shared, u64 *result, u64 batch, bool accumulate, bool into_coeff,
                      hipStream_t s) {
    const u64 stride = ggsw_shared ? 0ull : (((u64)rows * ell * (k + 1) * L) << log_n);
    if (k == 1) {
        if (arith == kArithMont)
            return dispatch_extprod_small<MontArith, 2>(log_n, sdigits, ggsw, stride, result, primes, L, rows, ell, batch, accumulate, into_coeff, s);
        return arith == kArithPm
                   ? dispatch_extprod_small<PmArith, 2>(log_n, sdigits, ggsw, stride, result, primes, L, rows, ell, batch, accumulate, into_coeff, s)
                   : dispatch_extprod_small<ShoupArith, 2>(log_n, sdigits, ggsw, stride, result, primes, L, rows, ell, batch, accumulate, into_coeff, s);
    }
    return PFHE_ERR_UNSUPPORTED;
}

// ---- the <u32> product's fused kernels: N = 2^16 (2^15 words = 4 strided stages x blocks of 2^11 words), k = 1 ----
bool extprod32_fused_supported(u32 log_n, u32 k) {
    if (k != 1 || log_n != 16) return false;
    const NttPlan plan = make_ntt_plan(log_n - 1, kArithB32);
    return !plan.tiny && plan.n_strided == 1 && plan.strided[0] == 4 && plan.block_log == 11;
}

int digits_strided32_dev(const NttPrime *primes, u32 L, u32 log_n, u32 ell, const int *sdigits, u32 *digits, u64 npolys,
                         hipStream_t s) {
    if (!extprod32_fused_supported(log_n, 1)) return PFHE_ERR_UNSUPPORTED;
    const u32 log_nw = log_n - 1;
    const u64 total = (npolys * ell) << (log_nw - 4);
    if (total == 0) return PFHE_OK;
    hipLaunchKernelGGL(digits_strided32_kernel<4>, dim3((u32)((total + 255) / 256)), dim3(256), 0, s, primes, L, log_nw, sdigits,
                       reinterpret_cast<u64 *>(digits), total);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

int gadget_block_mulacc32_dev(const NttPrime *primes, u32 L, u32 log_n, u32 terms, const u32 *digits, const u32 *ggsw,
                              bool ggsw_shared, u32 *result, u64 batch, bool accumulate, bool inv_tail, hipStream_t s) {
    if (!extprod32_fused_supported(log_n, 1) || terms == 0 || (inv_tail && accumulate)) return PFHE_ERR_UNSUPPORTED;
    const u32 log_nw = log_n - 1;
    const u64 total_blocks = (batch * L) << (log_nw - 11);
    if (total_blocks == 0) return PFHE_OK;
    if (total_blocks > 0x7fffffffull) return PFHE_ERR_BAD_LENGTH;
    const u64 ggsw_words = ((u64)terms * 2 * L) << log_nw;  // 64-bit words of one GGSW's rows
    constexpr size_t lds_bytes = (size_t)BlockCfg<11, 3>::LDS_WORDS * sizeof(u64);
    hipLaunchKernelGGL((gadget_block_mulacc32_kernel<2>), dim3((u32)total_blocks), dim3(256), lds_bytes, s,
                       reinterpret_cast<const u64 *>(digits), reinterpret_cast<const u64 *>(ggsw), ggsw_shared ? 0ull : ggsw_words,
                       reinterpret_cast<u64 *>(result), primes, L, log_nw, terms, total_blocks, accumulate ? 1u : 0u,
                       inv_tail ? 1u : 0u);
    PFHE_HIP(hipGetLastError());
    return PFHE_OK;
}

// balanced digits are stored as int32 when log_basis <= 31 (|digit| <= 2^(log_basis - 1)), else as int64
size_t gadget_digit_bytes(u32 log_basis) { return log_basis <= 31 ? sizeof(int) : sizeof(long long); }

template <class A, int K>
static int digits_strided_by_width(const RnsParams &r, const BasisParams &b, const NttPrime *primes, u32 log_n, const u64 *crt_polys,
                                   void *sdigits, u64 *digits, u64 npolys, hipStream_t s) {
    return b.dev.log_basis <= 31
               ? launch_digits_strided<A, K, int>(r, b, primes, log_n, crt_polys, (int *)sdigits, digits, npolys, s)
               : launch_digits_strided<A, K, long long>(r, b, primes, log_n, crt_polys, (long long *)sdigits, digits, npolys, s);
}

int gadget_decompose_strided_dev(const RnsParams &r, const BasisParams &b, const NttPrime *primes, u32 log_n, int arith,
                                 const u64 *crt_polys, u64 *digits, u64 npolys, hipStream_t s, void *sdigits) {
    if (!gadget_decompose_strided_supported(log_n, r.dev.value_len) || sdigits == nullptr) return PFHE_ERR_UNSUPPORTED;
    if (npolys == 0) return PFHE_OK;
    const int k = make_ntt_plan(log_n).strided[0];
    if (arith == kArithPm) {
        return k == 4 ? digits_strided_by_width<PmArith, 4>(r, b, primes, log_n, crt_polys, sdigits, digits, npolys, s)
                      : digits_strided_by_width<PmArith, 3>(r, b, primes, log_n, crt_polys, sdigits, digits, npolys, s);
    }
    if (arith == kArithMont) {
        return k == 4 ? digits_strided_by_width<MontArith, 4>(r, b, primes, log_n, crt_polys, sdigits, digits, npolys, s)
                      : digits_strided_by_width<MontArith, 3>(r, b, primes, log_n, crt_polys, sdigits, digits, npolys, s);
    }
    return k == 4 ? digits_strided_by_width<ShoupArith, 4>(r, b, primes, log_n, crt_polys, sdigits, digits, npolys, s)
                  : digits_strided_by_width<ShoupArith, 3>(r, b, primes, log_n, crt_polys, sdigits, digits, npolys, s);
}

}  // namespace pfhe
