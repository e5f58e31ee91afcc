// pfhe_rns.hpp — RNS base, gadget basis and external-product plan (host structs + launchers).
#pragma once
#include <memory>
#include <vector>

#include "pfhe_common.hpp"
#include "pfhe_handles.hpp"

namespace pfhe {

// Two forms of the same constants.  Bases of at most kMaxLimbs moduli travel BY VALUE as kernel arguments (RnsDev /
// BasisDev: SGPR-resident, the form every BASELINE config takes).  Wider bases — RNSBase::new (base.rs:79-117) takes any
// number of moduli — keep their constants in a device table built once per handle (RnsWide / BasisWide, up to
// kMaxWideLimbs moduli: a by-value RnsDev of that width would be 9 KiB of kernel arguments).  Both forms offer the same
// accessors, and the device functions (pfhe_rns_device.hpp) and kernels (pfhe_rns.hip) are templates over them.
constexpr int kMaxLimbs = 8;
constexpr int kMaxWideLimbs = 32;
// The <u32> product's lazy 64-bit accumulators (one v_mad_u64_u32 per coefficient and term): digit_hat and key words are
// canonical (< q < 2^30), a product is at most (2^30 - 1)^2 < 2^60, so fifteen of them on top of a folded value (< 2^30)
// stay below 2^64; the accumulators are folded once per kFold32Every terms and at the end.
constexpr unsigned kFold32Every = 15;

// RNSBase<u64, BarrettModulus<u64>> constants (primus_rns/src/base.rs:26-117), passed by value.
// L and value_len are valid for every base; the arrays only when L <= kMaxLimbs.
struct RnsDev {
    u32 L, value_len;
    // 1: the decomposition kernels read their input as big integers (value_len limbs per coefficient,
    // coefficient-major: BigUintPolynomial) instead of composing it from L residues (CrtPolynomial)
    u32 big_input;
    // words of a big integer in the CALLER's word type: value_len for RNSBase<u64>; for RNSBase<u32> (pfhe_rns32: the same
    // constants, 32-bit residues and limbs in memory) the number of u32 limbs of Q, 2 * value_len or one less
    u32 value_words;
    u64 q[kMaxLimbs];
    u64 inv_punct[kMaxLimbs], inv_punct_p[kMaxLimbs];  // (Q/q_i)^-1 mod q_i and its Shoup quotient
    u64 punct[kMaxLimbs][kMaxLimbs];                    // Q/q_i, little-endian limbs
    u64 Q[kMaxLimbs];                                   // product of all moduli
    // Mixed-radix (Garner) form of the same CRT lift for bases of 2 or 3 moduli with max q < 2 min q (every prime of
    // the BASELINE configs): x = v0 + q0*v1 + q0*q1*v2 with v_i < q_i needs 1 / 3 modular products and 1 / 3 widening
    // ones instead of L Shoup products, L*value_len widening ones and L compare-and-subtract passes over value_len
    // limbs; the result is the same integer in [0, Q).  garner = 0: the general form below it.
    u32 garner, pad2_;
    u64 g_inv[3][2], g_inv_p[3][2];  // g_inv[i][j] = q_j^-1 mod q_i (j < i) and its Shoup quotient
    u64 g_prod[2];                   // q0*q1, little-endian
    __host__ __device__ u64 modulus(u32 i) const { return q[i]; }
    __host__ __device__ u64 inv(u32 i) const { return inv_punct[i]; }
    __host__ __device__ u64 inv_p(u32 i) const { return inv_punct_p[i]; }
    __host__ __device__ u64 punctured(u32 i, u32 j) const { return punct[i][j]; }
    __host__ __device__ u64 product(u32 j) const { return Q[j]; }
};

// The device-table form: `tab` holds q[W] | inv_punct[W] | inv_punct_p[W] | Q[W] | floor(2^128/q) low[W] | high[W] |
// punct[W][W] with W = kMaxWideLimbs, every row zero-padded to W words (a kernel instantiated for a limb count above
// value_len computes with zero top limbs).
struct RnsWide {
    u32 L, value_len;
    u32 big_input, value_words;
    const u64 *tab;
    static constexpr u32 W = kMaxWideLimbs;
    static constexpr u32 garner = 0;
    __device__ u64 modulus(u32 i) const { return tab[i]; }
    __device__ u64 inv(u32 i) const { return tab[W + i]; }
    __device__ u64 inv_p(u32 i) const { return tab[2 * W + i]; }
    __device__ u64 product(u32 j) const { return tab[3 * W + j]; }
    __device__ u64 ratio_lo(u32 i) const { return tab[4 * W + i]; }  // BarrettModulus ratio (barrett/mod.rs:52-59)
    __device__ u64 ratio_hi(u32 i) const { return tab[5 * W + i]; }
    __device__ u64 punctured(u32 i, u32 j) const { return tab[(6 + i) * W + j]; }
    static constexpr size_t table_words() { return (size_t)(6 + W) * W; }
};

// BigUintApproxSignedBasis<u64> constants (primus_decompose/src/big_integer/basis.rs:17-211).
struct BasisCore {
    u32 value_len, ell, log_basis, drop_bits;
    u32 mode;  // bit0: extract initial carry, bit1: adjust values >= threshold
    u32 carry_index;
    u32 value_words, pad_;  // as RnsDev::value_words
    u64 carry_bit_mask;
    u64 basis, basis_minus_one, carry_mask;
};
struct BasisDev : BasisCore {  // by value; arrays valid when value_len <= kMaxLimbs
    u64 threshold[kMaxLimbs], add[kMaxLimbs];
    __host__ __device__ u64 split(u32 j) const { return threshold[j]; }
    __host__ __device__ u64 addend(u32 j) const { return add[j]; }
};
struct BasisWide : BasisCore {  // tab: threshold[W] | add[W], zero-padded
    const u64 *tab;
    __device__ u64 split(u32 j) const { return tab[j]; }
    __device__ u64 addend(u32 j) const { return tab[kMaxWideLimbs + j]; }
};

// a device allocation shared by the handles that copy a base's constants (an RNSBase, the bases derived from it, the
// external-product plans): freed with its last holder
struct DeviceBlob {
    int device = 0;
    void *ptr = nullptr;
    ~DeviceBlob();
};

// what the launchers take: the by-value form, or (wide()) the device-table form and its owner
struct RnsParams {
    RnsDev dev{};
    RnsWide wide_tab{};
    std::shared_ptr<DeviceBlob> blob;
    bool wide() const { return dev.L > (u32)kMaxLimbs; }
};
struct BasisParams {
    BasisDev dev{};
    BasisWide wide_tab{};
    std::shared_ptr<DeviceBlob> blob;
    bool wide() const { return dev.value_len > (u32)kMaxLimbs; }
};

struct RnsHost {
    int device = 0;
    u32 word_bits = 64;  // width of the caller's words (see RnsDev::value_words)
    RnsParams par;
    // host copies, any width: moduli; Q and every Q/q_i as value_len little-endian limbs; (Q/q_i)^-1 mod q_i + quotient
    std::vector<u64> moduli, Q, punct, inv_punct, inv_punct_p;
    std::vector<u64> ratio_lo, ratio_hi;  // floor(2^128 / q_i)
};

struct BasisHost {
    int device = 0;
    u32 word_bits = 64;
    BasisParams par;
    RnsParams rns;
    std::vector<u64> Q;                // of the base it was built from (basis.rs:52)
    std::vector<u64> threshold, add;   // value_len limbs each (zero when the mode bit is clear)
    std::vector<u64> scalars;          // ell * value_len : 2^(drop + j*log_basis)
    std::vector<u64> scalars_residue;  // ell * L
};

// word_bits: 64 (RNSBase<u64>: moduli below 2^62) or 32 (RNSBase<u32>: below 2^30, BarrettModulus<u32>)
int build_rns(const u64 *moduli, size_t count, RnsHost &out, u32 word_bits = 64);
int build_basis(const RnsHost &rns, u32 log_basis, size_t reverse_length, BasisHost &out);
// device tables of the wide forms (no-ops for bases that fit the by-value form); `device` must be current
int upload_rns_wide(RnsHost &r);
int upload_basis_wide(BasisHost &b);

// --- device launchers (pfhe_rns.hip); all pointers are device pointers ---
// WT = u64 (RNSBase<u64>, BigUintApproxSignedBasis<u64>) or u32 (the <u32> instantiations of the same generics,
// base.rs:26-37, big_integer/basis.rs:33): residues, digits and big-integer limbs are WT words in memory; the arithmetic
// inside is the same (every result is a canonical integer, so the word width of the reference's arithmetic is not visible).
template <class WT>
int rns_compose_dev(const RnsParams &r, const WT *multi_residues, WT *big_uint_values, u64 value_count, hipStream_t s);
template <class WT>
int rns_wrapping_decompose_dev(const RnsParams &r, const WT *small_values, WT *multi_residues, u64 value_count,
                               u64 small_value_modulus, hipStream_t s);
// acc += factor * lift(small) per modulus (centred lift unless !centred); factor_pairs: L host (value, quotient) pairs of
// 64-bit Shoup factors (quotient = floor(value * 2^64 / q))
template <class WT>
int rns_add_decompose_scaled_dev(const RnsParams &r, const WT *small_values, WT *acc, u64 value_count,
                                 u64 small_value_modulus, bool centred, const u64 *factor_pairs, hipStream_t s);
template <class WT>
int rns_decompose_big_dev(const RnsParams &r, const WT *big_uint_values, WT *multi_residues, u64 value_count, hipStream_t s);
template <class WT>
int basis_init_value_carry_dev(const BasisParams &b, WT *values, unsigned char *carries, u64 count, hipStream_t s);
template <class WT>
int basis_unsigned_decompose_dev(const BasisParams &b, u32 level, const WT *values, WT *digits,
                                 unsigned char *carries, u64 count, hipStream_t s);
// signed digits as residues modulo Q, value_words limbs each (common.rs:255-272, 289-306)
template <class WT>
int basis_signed_decompose_dev(const RnsParams &r, const BasisParams &b, u32 level, const WT *values, WT *decomposed,
                               unsigned char *carries, u64 count, hipStream_t s);
// Fused steps (1)-(4) of add_dcrt_glev_mul_crt_poly_assign (glwe/dcrt.rs:219-244) for `npolys` CRT
// polynomials of L*N words: writes, per input polynomial, ell digit polynomials in CRT form
// (centred lift), laid out [poly][level][limb][N].
template <class WT>
int gadget_decompose_dev(const RnsParams &r, const BasisParams &b, u32 log_n, const WT *crt_polys, WT *digits,
                         u64 npolys, hipStream_t s);
// result[e][c][r][t] = sum_{i,j} ggsw[(e)][i][j][c][r][t] * digits[e][i][j][r][t]  (mod q_r)
// (steps (6) of glwe/dcrt.rs:248 summed over rows and levels, glwe/crt.rs:219-226).
// accumulate != 0 adds into the existing result instead of overwriting it.
int gadget_mulacc_dev(const NttPrime *primes, u32 L, u32 log_n, u32 k, u32 rows, u32 ell, const u64 *digits,
                      const u64 *ggsw, bool ggsw_shared, u64 *result, u64 batch, bool accumulate, hipStream_t s);
// the same on u32 residues (primes: a U32DcrtTable's, q < 2^30, bar_lo = floor(2^64 / q))
int gadget_mulacc32_dev(const NttPrime *primes, u32 L, u32 log_n, u32 k, u32 rows, u32 ell, const u32 *digits,
                        const u32 *ggsw, bool ggsw_shared, u32 *result, u64 batch, bool accumulate, hipStream_t s);

// Fused "last forward pass + multiply-accumulate" (pfhe_extprod.hip): `digits` must already have gone
// through the strided passes of the forward transform.  Available for k = 1 and a 2^12 block pass.
bool gadget_fused_supported(u32 log_n, u32 k);
// Steps (1)-(4) fused with the first (strided) pass of the forward transform (pfhe_extprod.hip).
bool gadget_decompose_strided_supported(u32 log_n, u32 value_len);
// `sdigits`: scratch of npolys * ell * N balanced digits of gadget_digit_bytes(log_basis) bytes each (int32 when
// log_basis <= 31, else int64): a compact signed-digit kernel + a lifting strided pass.
size_t gadget_digit_bytes(u32 log_basis);
// `arith`: the table's transform arithmetic (kArithPm, kArithMont or kArithShoup, pfhe_ntt_device.hpp)
int gadget_decompose_strided_dev(const RnsParams &r, const BasisParams &b, const NttPrime *primes, u32 log_n, int arith,
                                 const u64 *crt_polys, u64 *digits, u64 npolys, hipStream_t s, void *sdigits);
// inv_tail (not with accumulate): the kernel also runs the block pass of the INVERSE transform on its result blocks
// before storing them; the caller finishes with the inverse transform's strided pass (ntt_pass_dev, inverse, index 1).
int gadget_block_mulacc_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u32 k, u32 terms, const u64 *digits,
                            const u64 *ggsw, bool ggsw_shared, u64 *result, u64 batch, bool accumulate,
                            hipStream_t s, bool inv_tail = false);

// Small rings (N = 2^10..2^12): digit extraction to int32 + ONE kernel for transforms, multiply-accumulate and
// (optionally) the inverse transforms (pfhe_extprod.hip, extprod_small_kernel).
bool extprod_small_supported(u32 log_n, u32 k, u32 value_len, u32 log_basis);
template <class WT>
int gadget_signed_digits_dev(const RnsParams &r, const BasisParams &b, u32 log_n, const WT *crt_polys, int *sdigits,
                             u64 npolys, hipStream_t s);
// The <u32> product's fused kernels (pfhe_extprod.hip, B32Arith: two u32 coefficients per 64-bit word) for N = 2^16, k = 1:
// lift of the balanced digits + the forward transform's strided pass (4 stages) into `digits` ([poly][level][limb][N] u32),
// then block pass (2^11 words) + multiply-accumulate (+ the inverse transform's block pass when inv_tail) per
// (ciphertext, limb, block).  The caller finishes a coefficient-form result with the inverse strided pass.
bool extprod32_fused_supported(u32 log_n, u32 k);
int digits_strided32_dev(const NttPrime *primes, u32 L, u32 log_n, u32 ell, const int *sdigits, u32 *digits, u64 npolys,
                         hipStream_t s);
int gadget_block_mulacc32_dev(const NttPrime *primes, u32 L, u32 log_n, u32 terms, const u32 *digits, const u32 *ggsw,
                              bool ggsw_shared, u32 *result, u64 batch, bool accumulate, bool inv_tail, hipStream_t s);
int extprod_small_dev(const NttPrime *primes, u32 L, u32 log_n, int arith, u32 k, u32 rows, u32 ell, const int *sdigits,
                      const u64 *ggsw, bool ggsw_shared, u64 *result, u64 batch, bool accumulate, bool into_coeff,
                      hipStream_t s);

}  // namespace pfhe

// handles behind the C ABI (shared by pfhe_capi_rns.hip and pfhe_convert.hip); the *32 ones are the <u32> instantiations
// of the same reference generics: the same host structs with word_bits = 32
struct pfhe_rns {
    pfhe::RnsHost h;
};
struct pfhe_rns32 {
    pfhe::RnsHost h;
};
struct pfhe_basis {
    pfhe::BasisHost h;
};
struct pfhe_basis32 {
    pfhe::BasisHost h;
};
