/*
 * pfhe.h — C ABI of the MI355X-native polynomial-ring engine (libpfhe_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of primus-labs/primus-fhe: negacyclic NTT/INTT
 * over 64-bit primes, RNS ("DCRT") pointwise arithmetic, and the RNS gadget external product.
 * The reference has no FFI of its own; the seam is the pair of Rust traits `NttTable` /
 * `DcrtTable` plus the slice-level functions of primus_poly / primus_rns / primus_decompose /
 * primus_lattice that are generic over them.  Every entry point below names the reference item
 * (file:line under /root/reference/crates/) it replaces; INTEGRATION.md shows the Rust binding.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; every function returns a pfhe_status (0 = OK)
 *     and never aborts or throws across the boundary (the reference panics / debug_asserts).
 *   - words are uint64_t; layouts are the reference's: a polynomial is N contiguous words, an
 *     RNS polynomial is L x N words modulus-major (primus_rns/src/lib.rs:12-16), batches are
 *     plain concatenation.  `len` is always the TOTAL number of words and must be a multiple of
 *     the unit size (the reference only debug_asserts lengths; we return PFHE_ERR_BAD_LENGTH).
 *   - `*_slice` functions take HOST pointers and behave exactly like the reference call
 *     (in place, synchronous).  `*_dev` functions take DEVICE pointers that live on the handle's
 *     GPU plus a hipStream_t (as void*, NULL = default stream); they are asynchronous and are
 *     the measured hot path.  Device buffers must be 16-byte aligned (PFHE_ERR_BAD_ARGUMENT
 *     otherwise; hipMalloc / pfhe_device_malloc memory always is).
 *   - handles are immutable after creation and may be shared between host threads
 *     (NttTable: Send + Sync, primus_ntt/src/ntt/mod.rs:16); an external-product plan owns
 *     scratch and has one holder at a time (it mirrors `&mut DcrtGlevContext`,
 *     primus_lattice/src/context/glev.rs:4-10): a call from a second thread while one is inside
 *     is refused with PFHE_ERR_BUSY, not raced.
 *   - there is NO CPU fallback: without a HIP device create() fails with PFHE_ERR_NO_DEVICE.
 */
#ifndef PFHE_H
#define PFHE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: 1..5 mirror NttError (primus_ntt/src/error.rs:7-49), 16..18 RNSError
 *      (primus_rns/src/lib.rs), 32.. are boundary errors the reference cannot have. ---- */
typedef enum pfhe_status {
    PFHE_OK = 0,
    PFHE_ERR_NO_PRIMITIVE_ROOT = 1,
    PFHE_ERR_DEGREE_CONVERSION = 2,
    PFHE_ERR_DEGREE_TOO_LARGE = 3,
    PFHE_ERR_NTT_TABLE = 4,
    PFHE_ERR_MODULUS_TOO_LARGE = 5,
    PFHE_ERR_EMPTY_BASE = 16,
    PFHE_ERR_COPRIME = 17,
    PFHE_ERR_UNREPRESENTABLE_MODULUS = 18,
    PFHE_ERR_BAD_LENGTH = 32,
    PFHE_ERR_BAD_ARGUMENT = 33,
    PFHE_ERR_NO_DEVICE = 34,
    PFHE_ERR_HIP = 35,
    PFHE_ERR_UNSUPPORTED = 36,
    PFHE_ERR_NO_INVERSE = 37, /* ReduceError::NoInverse: an element of an inversion is not a unit */
    PFHE_ERR_BUSY = 38        /* an external-product plan is held by another thread (one holder at a time) */
} pfhe_status;

const char *pfhe_status_string(int status);
/* Thread-local detail of the last failure on this thread (e.g. the HIP error string). */
const char *pfhe_last_error(void);
/* "libpfhe_hip <version> gfx950" */
const char *pfhe_version(void);

/* ---- device plumbing (so a Rust/C caller needs no HIP headers) ---- */
int pfhe_device_count(int *count);
int pfhe_device_malloc(int device, size_t bytes, void **out);
int pfhe_device_free(int device, void *ptr);
int pfhe_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes, void *stream);
int pfhe_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes, void *stream);
int pfhe_memcpy_d2d(int device, void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int pfhe_memset_dev(int device, void *dst_dev, int byte, size_t bytes, void *stream);
/* Measurement aid (bench.py `device_copy`): a copy by a kernel of the element-wise family's launch shape — one 16-byte vector
 * per thread, non-temporal loads and stores; `bytes` a multiple of 16, both pointers 16-byte aligned, ranges disjoint. */
int pfhe_stream_copy_dev(int device, void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int pfhe_stream_synchronize(int device, void *stream);
/* Host-pointer entry points (`*_slice`, `*_to` without `_dev`) behave like the reference's `&self, &mut [T]` methods
 * (table.rs:541-563: in place, no allocation per call): each call borrows a staging context — private streams, cached
 * device arena, pinned host buffer — from a per-device pool and returns it, so calls of a size seen before allocate
 * nothing and any number of threads may call through one handle concurrently (NttTable: Send + Sync).
 * pfhe_debug_alloc_count: device / pinned allocation and free calls the library has made since it was loaded (a
 * steady-state loop must not move it).  pfhe_staging_release: frees the idle contexts of `device` (-1: all devices),
 * returns their number (at most four are kept per device anyway).
 * pfhe_debug_stage_path_count(which): host-pointer calls that took path `which` since the library was loaded —
 * 0 kernels on pinned memory the caller ALLOCATED (hipHostMalloc, a torch pinned tensor), 1 kernels on the pool's own pinned
 * buffer (pageable slices, and since round 5 slices the caller merely REGISTERED with hipHostRegister: kernels running on a
 * per-call registration return rare wrong words on this platform with plain HIP alone — tools/microbench12_register_hazard.hip),
 * 2 copy engines on caller-pinned memory (allocated or registered), 3 the runtime's pageable copies, 4 long pageable slices
 * with the copy back on the context's helper thread, 5 small uploads of the non-transform entry points: a CPU copy into the
 * pool's pinned buffer + a copy-engine transfer.
 * Every entry point clears the calling thread's pending HIP error (hipGetLastError is sticky per thread) on the way in — the
 * outermost one of a call only — so that an earlier failed HIP call of the caller's own is not reported as a failure of
 * this library's launches: check the return values of your own HIP calls, not hipGetLastError after a pfhe_* call.
 * Streams: since round 4 the host-pointer entry points run on PRIVATE non-blocking streams of the borrowed context, not on
 * the legacy null stream: they are ordered with respect to nothing the caller has queued elsewhere (they block until
 * their own work is done, which is all `&mut [T]` semantics need). */
uint64_t pfhe_debug_alloc_count(void);
uint64_t pfhe_debug_stage_path_count(int which);
int pfhe_staging_release(int device);
/* Synthetic data: word i of the buffer = floor(splitmix64(seed, i) * q / 2^64), i.e. uniform in
 * [0,q) (per-modulus uniform sampling as primus_distr/src/common.rs:244-263).  Modulus-major RNS
 * layout when `moduli_count` > 1: word i uses moduli[(i / poly_len) % moduli_count]. */
int pfhe_fill_uniform_dev(int device, uint64_t *dst_dev, size_t len, const uint64_t *moduli,
                          size_t moduli_count, size_t poly_len, uint64_t seed, void *stream);

/* =====================================================================================
 * U64NttTable — primus_ntt/src/ntt/prime64/table.rs:41 implementing NttTable
 * (primus_ntt/src/ntt/mod.rs:16-113)
 * ===================================================================================== */
typedef struct pfhe_ntt pfhe_ntt;

/* NttTable::new(log_n, modulus) — table.rs:308-516.  Errors: NO_PRIMITIVE_ROOT when 2N does not
 * divide q-1 (root.rs:76-81), MODULUS_TOO_LARGE when q >= 2^62 (table.rs:318-323). */
int pfhe_ntt_create(uint32_t log_n, uint64_t modulus, int device, pfhe_ntt **out);
void pfhe_ntt_destroy(pfhe_ntt *table);
/* getters — table.rs:127-161, ntt/mod.rs:26 (poly_length) */
size_t pfhe_ntt_poly_length(const pfhe_ntt *table);
uint32_t pfhe_ntt_log_n(const pfhe_ntt *table);
uint64_t pfhe_ntt_modulus(const pfhe_ntt *table);
uint64_t pfhe_ntt_root(const pfhe_ntt *table);
uint64_t pfhe_ntt_inv_root(const pfhe_ntt *table);
uint64_t pfhe_ntt_inv_n(const pfhe_ntt *table);
int pfhe_ntt_device(const pfhe_ntt *table);

/* transform_slice / inverse_transform_slice / lazy_* — table.rs:541-563.  `len` = batch * N;
 * each N-word chunk is transformed independently in place.
 *   forward: normal order in, bit-reversed out; canonical [0,q) (lazy: [0,4q) in and out)
 *   inverse: bit-reversed in, normal order out; canonical [0,q) (lazy: [0,2q) in and out) */
int pfhe_ntt_transform_slice(const pfhe_ntt *table, uint64_t *poly, size_t len);
int pfhe_ntt_inverse_transform_slice(const pfhe_ntt *table, uint64_t *values, size_t len);
int pfhe_ntt_lazy_transform_slice(const pfhe_ntt *table, uint64_t *poly, size_t len);
int pfhe_ntt_lazy_inverse_transform_slice(const pfhe_ntt *table, uint64_t *values, size_t len);
/* transform_monomial / transform_coeff_one_monomial / transform_coeff_minus_one_monomial —
 * table.rs:565-651.  `values` receives N words. */
int pfhe_ntt_transform_monomial(const pfhe_ntt *table, uint64_t coeff, size_t degree,
                                uint64_t *values, size_t len);
int pfhe_ntt_transform_coeff_one_monomial(const pfhe_ntt *table, size_t degree, uint64_t *values,
                                          size_t len);
int pfhe_ntt_transform_coeff_minus_one_monomial(const pfhe_ntt *table, size_t degree,
                                                uint64_t *values, size_t len);
/* device-pointer variants (hot path).  lazy != 0 selects the lazy_* contract. */
int pfhe_ntt_transform_dev(const pfhe_ntt *table, uint64_t *poly_dev, size_t len, int lazy,
                           void *stream);
int pfhe_ntt_inverse_transform_dev(const pfhe_ntt *table, uint64_t *values_dev, size_t len,
                                   int lazy, void *stream);
int pfhe_ntt_transform_monomial_dev(const pfhe_ntt *table, uint64_t coeff, size_t degree,
                                    uint64_t *values_dev, size_t len, void *stream);
/* NttPolynomial::mul_assign / add_mul_assign — primus_poly/src/ntt/mul.rs:84-90,
 * ntt/mod.rs:101-112 (-> reduce_mul_slice_assign / reduce_add_mul_slice_assign,
 * primus_modulus/src/common/compact/slice.rs:106-115,210-221).  len_b is len_a (elementwise)
 * or N (one multiplicand shared by the whole batch). */
int pfhe_ntt_mul_assign_dev(const pfhe_ntt *table, uint64_t *a_dev, size_t len_a,
                            const uint64_t *b_dev, size_t len_b, void *stream);
int pfhe_ntt_add_mul_assign_dev(const pfhe_ntt *table, uint64_t *acc_dev, const uint64_t *a_dev,
                                size_t len_a, const uint64_t *b_dev, size_t len_b, void *stream);
/* NttPolynomial::mul_to / mul_add_to — primus_poly/src/ntt/mul.rs:100-107, ntt/mod.rs:169-187:
 * out = a*b and out = a*b + c (out may alias an input). */
int pfhe_ntt_mul_to_dev(const pfhe_ntt *table, const uint64_t *a_dev, size_t len_a,
                        const uint64_t *b_dev, size_t len_b, uint64_t *out_dev, void *stream);
int pfhe_ntt_mul_add_to_dev(const pfhe_ntt *table, const uint64_t *a_dev, size_t len_a,
                            const uint64_t *b_dev, size_t len_b, const uint64_t *c_dev,
                            uint64_t *out_dev, void *stream);

/* =====================================================================================
 * U64DcrtTable — primus_ntt/src/dcrt/prime64.rs:11 implementing DcrtTable
 * (primus_ntt/src/dcrt/mod.rs:19-135).  Unit = one RNS polynomial = L*N words, modulus-major.
 * ===================================================================================== */
typedef struct pfhe_dcrt pfhe_dcrt;

/* DcrtTable::new(log_n, moduli) — dcrt/prime64.rs:24-43 (per-limb NttTable::new errors). */
int pfhe_dcrt_create(uint32_t log_n, const uint64_t *moduli, size_t moduli_count, int device,
                     pfhe_dcrt **out);
void pfhe_dcrt_destroy(pfhe_dcrt *table);
size_t pfhe_dcrt_poly_length(const pfhe_dcrt *table);     /* dcrt/prime64.rs:56 */
size_t pfhe_dcrt_moduli_count(const pfhe_dcrt *table);    /* :61 */
size_t pfhe_dcrt_crt_poly_length(const pfhe_dcrt *table); /* :66 */
int pfhe_dcrt_device(const pfhe_dcrt *table);
/* ntt_tables()[i] getters (dcrt/prime64.rs:46) */
uint64_t pfhe_dcrt_modulus(const pfhe_dcrt *table, size_t i);
uint64_t pfhe_dcrt_root(const pfhe_dcrt *table, size_t i);
uint64_t pfhe_dcrt_inv_n(const pfhe_dcrt *table, size_t i);

/* transform_slice / inverse_transform_slice / lazy_* — dcrt/prime64.rs:98-127.
 * `len` = batch * L * N. */
int pfhe_dcrt_transform_slice(const pfhe_dcrt *table, uint64_t *poly, size_t len);
int pfhe_dcrt_inverse_transform_slice(const pfhe_dcrt *table, uint64_t *poly, size_t len);
int pfhe_dcrt_lazy_transform_slice(const pfhe_dcrt *table, uint64_t *poly, size_t len);
int pfhe_dcrt_lazy_inverse_transform_slice(const pfhe_dcrt *table, uint64_t *poly, size_t len);
/* DcrtTable::transform_monomial & co — dcrt/mod.rs:105-134.  `values` receives L*N words. */
int pfhe_dcrt_transform_monomial(const pfhe_dcrt *table, uint64_t coeff, size_t degree,
                                 uint64_t *values, size_t len);
/* transform_coeff_one_monomial / transform_coeff_minus_one_monomial — dcrt/mod.rs:113-134
 * (-X^degree uses q_i - 1 in limb i). */
int pfhe_dcrt_transform_coeff_one_monomial(const pfhe_dcrt *table, size_t degree, uint64_t *values,
                                           size_t len);
int pfhe_dcrt_transform_coeff_minus_one_monomial(const pfhe_dcrt *table, size_t degree,
                                                 uint64_t *values, size_t len);
/* device-pointer variants of the three monomial transforms: launches on `stream` only (the per-limb coefficients
 * travel as kernel arguments), no allocation or synchronisation — capturable into a HIP graph (a CMUX / blind-rotate
 * loop builds X^d in NTT form every step).  minus_one != 0 selects -X^degree (coeff is then ignored). */
int pfhe_dcrt_transform_monomial_dev(const pfhe_dcrt *table, uint64_t coeff, size_t degree,
                                     uint64_t *values_dev, size_t len, int minus_one, void *stream);
int pfhe_dcrt_transform_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len, int lazy,
                            void *stream);
int pfhe_dcrt_inverse_transform_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len,
                                    int lazy, void *stream);
/* DcrtPolynomial::mul_assign (primus_poly/src/dcrt/mul.rs:176-187) and add_mul_assign
 * (primus_poly/src/dcrt/mod.rs:105-123): per limb Barrett a = a*b, acc = a*b + acc, canonical
 * inputs and outputs.  len_b is len_a or L*N (shared multiplicand). */
int pfhe_dcrt_mul_assign_dev(const pfhe_dcrt *table, uint64_t *a_dev, size_t len_a,
                             const uint64_t *b_dev, size_t len_b, void *stream);
int pfhe_dcrt_add_mul_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev,
                                 const uint64_t *a_dev, size_t len_a, const uint64_t *b_dev,
                                 size_t len_b, void *stream);
/* DcrtGlwe::add_dcrt_glwe_mul_dcrt_polynomial_assign — primus_lattice/src/glwe/dcrt.rs:107-126,
 * batched: acc and dcrt_glwe hold batch ciphertexts of `glwe_polys` (= k+1) RNS polynomials,
 * dcrt_poly one RNS polynomial per ciphertext (len_poly = len / glwe_polys):
 * acc[e][c] += dcrt_glwe[e][c] * dcrt_poly[e]. */
int pfhe_dcrt_add_dcrt_glwe_mul_dcrt_polynomial_assign_dev(const pfhe_dcrt *table,
                                                           uint64_t *acc_dev,
                                                           const uint64_t *dcrt_glwe_dev, size_t len,
                                                           const uint64_t *dcrt_poly_dev,
                                                           size_t len_poly, size_t glwe_polys,
                                                           void *stream);
/* DcrtGlwe::mul_dcrt_polynomial_to — glwe/dcrt.rs:377-395, batched the same way:
 * result[e][c] = dcrt_glwe[e][c] * dcrt_poly[e]; result may alias dcrt_glwe. */
int pfhe_dcrt_glwe_mul_dcrt_polynomial_to_dev(const pfhe_dcrt *table, const uint64_t *dcrt_glwe_dev, size_t len,
                                              const uint64_t *dcrt_poly_dev, size_t len_poly, size_t glwe_polys,
                                              uint64_t *result_dev, void *stream);
/* DcrtPolynomial::mul_to (primus_poly/src/dcrt/mul.rs:232-250) and the out-of-place
 * multiply-add: out = a*b, out = a*b + c (out may alias an input). */
int pfhe_dcrt_mul_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, size_t len_a,
                         const uint64_t *b_dev, size_t len_b, uint64_t *out_dev, void *stream);
int pfhe_dcrt_mul_add_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, size_t len_a,
                             const uint64_t *b_dev, size_t len_b, const uint64_t *c_dev,
                             uint64_t *out_dev, void *stream);
/* GLWE butterfly (self, result) = (self + rhs, (self_orig - rhs) * w), canonical in and out, over
 * every DCRT polynomial of a (batch of) DcrtGlwe:
 *   DcrtGlwe::butterfly_mul_dcrt_polynomial_to — primus_lattice/src/glwe/dcrt.rs:128-155
 *     (-> DcrtPolynomial::butterfly_mul_to, primus_poly/src/dcrt/mod.rs:125-160); w = len_w plain
 *     residues: one DCRT polynomial (L*N, shared by all components) or len.
 *   DcrtGlwe::butterfly_mul_factor_to — glwe/dcrt.rs:157-175 (-> DcrtPolynomial::butterfly_mul_factor_to,
 *     primus_poly/src/dcrt/mul.rs:15-30,196-222); w = ShoupFactor<u64> pairs (value, quotient),
 *     len_w = 2*L*N (shared) or 2*len words. */
int pfhe_dcrt_butterfly_mul_dcrt_polynomial_to_dev(const pfhe_dcrt *table, uint64_t *a_dev,
                                                   const uint64_t *rhs_dev, size_t len,
                                                   const uint64_t *dcrt_poly_dev, size_t len_w,
                                                   uint64_t *result_dev, void *stream);
int pfhe_dcrt_butterfly_mul_factor_to_dev(const pfhe_dcrt *table, uint64_t *a_dev,
                                          const uint64_t *rhs_dev, size_t len,
                                          const uint64_t *factor_poly_dev, size_t len_w,
                                          uint64_t *result_dev, void *stream);
/* Fused "NTT -> pointwise mul by dcrt_poly -> INTT" of CRT polynomials in place:
 * CrtRlwe::mul_dcrt_polynomial_to (primus_lattice/src/rlwe/crt.rs:42-65) followed by
 * DcrtRlwe::into_coeff_form (primus_lattice/src/macros/mod.rs:901-911) per CRT polynomial.
 * Contract (the reference's: reduce_mul_slice_assign takes canonical operands, primus_reduce/src/slice_ops.rs:137-229):
 * crt_poly_dev AND dcrt_poly_dev hold CANONICAL residues in [0, q_i).  A lazily transformed multiplicand ([0, 4q), the
 * output of lazy_transform_slice) is outside the contract: the fused kernels multiply the forward half's unreduced words
 * (up to 2^63 + 3q) by it with one 128 -> 64-bit reduction whose precondition is product < q * 2^64.  Reduce it first
 * (pfhe_dcrt_transform_dev with lazy = 0 gives canonical values). */
int pfhe_dcrt_mul_dcrt_polynomial_dev(const pfhe_dcrt *table, uint64_t *crt_poly_dev,
                                      size_t len, const uint64_t *dcrt_poly_dev, size_t len_b,
                                      void *stream);

/* ---- element-wise family on canonical residues (coefficient or NTT form alike) ----
 * CrtPolynomial / DcrtPolynomial: add, sub, neg (primus_poly/src/{crt,dcrt}/{add,sub,neg}.rs), mul_scalar,
 * add_mul_scalar, mul_factor, add_mul_factor, mul_monomial (crt/mul.rs:16-180, dcrt/mul.rs:78-300), inv
 * (dcrt/inv.rs:19-68); CrtGlwe::{add,sub}_element_wise{,_assign,_to} (primus_lattice/src/macros/mod.rs:367-531),
 * CrtGlwe::mul_scalar_{assign,to}, mul_factor_to, mul_monic_monomial_assign (glwe/crt.rs:59-175) — a GLWE is k+1
 * consecutive RNS polynomials, so the same entry points serve both.
 * `len` = total words, a multiple of L*N; every buffer holds `len` words; `out` may alias `a` (the *_assign
 * forms) and, for sub, `b` (sub_rev_assign, crt/sub.rs:69).  Inputs must be canonical ([0, q_r)), as in the
 * reference.  `scalars`: L residues on the host; `factors`: L ShoupFactor (value, quotient) pairs on the host.
 * The single-modulus NttPolynomial / Polynomial forms (primus_poly/src/ntt/{add,sub,neg,inv}.rs) are the L = 1 case:
 * bind them to a pfhe_dcrt created with one modulus.  add / sub / neg / mul_monomial / inv take tables of any number
 * of limbs; the forms with per-limb scalars or factors up to 32 (UNSUPPORTED beyond). */
int pfhe_dcrt_add_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev,
                         size_t len, void *stream);
int pfhe_dcrt_sub_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev,
                         size_t len, void *stream);
int pfhe_dcrt_neg_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, uint64_t *out_dev, size_t len, void *stream);
int pfhe_dcrt_mul_scalar_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *scalars,
                                uint64_t *out_dev, size_t len, void *stream);
int pfhe_dcrt_add_mul_scalar_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev, const uint64_t *rhs_dev,
                                        const uint64_t *scalars, size_t len, void *stream);
int pfhe_dcrt_mul_factor_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, const uint64_t *factors,
                                uint64_t *out_dev, size_t len, void *stream);
int pfhe_dcrt_add_mul_factor_assign_dev(const pfhe_dcrt *table, uint64_t *acc_dev, const uint64_t *rhs_dev,
                                        const uint64_t *factors, size_t len, void *stream);
/* self * X^r, 0 <= r < 2N, per N-word polynomial (rotate_right + negation of the wrapped part).  The _to form
 * needs out != a and moves each word once; the in-place form keeps a polynomial in one workgroup's registers for
 * 2^9 <= N <= 2^14 and goes through a stream-ordered scratch tile (twice the traffic, not capturable) otherwise. */
int pfhe_dcrt_mul_monomial_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, size_t r, uint64_t *out_dev,
                                  size_t len, void *stream);
int pfhe_dcrt_mul_monomial_assign_dev(const pfhe_dcrt *table, uint64_t *data_dev, size_t r, size_t len, void *stream);
/* Point-wise inverse.  The reference panics on a non-invertible element; this call synchronises the stream and
 * returns PFHE_ERR_NO_INVERSE (outputs unspecified).  Not capturable into a HIP graph. */
int pfhe_dcrt_inv_to_dev(const pfhe_dcrt *table, const uint64_t *a_dev, uint64_t *out_dev, size_t len, void *stream);

/* =====================================================================================
 * RNSBase<u64, BarrettModulus<u64>> — primus_rns/src/base.rs:26
 * ===================================================================================== */
typedef struct pfhe_rns pfhe_rns;

/* RNSBase::new(moduli) — base.rs:79-117.  Errors: EMPTY_BASE (:47-49), COPRIME (:83-89),
 * UNREPRESENTABLE_MODULUS when a modulus is not in (1, 2^62) (BarrettModulus::new,
 * primus_modulus/src/barrett/mod.rs:39-44), UNSUPPORTED for more than 32 moduli (bases of up to 8 moduli carry their
 * constants as kernel arguments, wider ones — up to 32 — in a device table owned by the handle and shared with the
 * bases, converters and plans derived from it; the same limit holds for both bases of a pfhe_conv). */
int pfhe_rns_create(const uint64_t *moduli, size_t count, int device, pfhe_rns **out);
void pfhe_rns_destroy(pfhe_rns *base);
size_t pfhe_rns_moduli_count(const pfhe_rns *base);        /* base.rs:124 */
size_t pfhe_rns_big_uint_value_len(const pfhe_rns *base);  /* base.rs:139 */
int pfhe_rns_moduli_product(const pfhe_rns *base, uint64_t *out, size_t len); /* base.rs:133 */
/* compose_multiple_values_to — base.rs:648-675 (-> compose_to :609-633): residue i of value c is
 * multi_residues[i*value_count + c] (modulus-major); value c is written as big_uint_value_len
 * little-endian limbs at big_uint_values[c*big_uint_value_len], canonical in [0, Q). */
int pfhe_rns_compose_multiple_values_to(const pfhe_rns *base, const uint64_t *multi_residues,
                                        size_t len_in, uint64_t *big_uint_values, size_t len_out,
                                        size_t value_count);
int pfhe_rns_compose_multiple_values_to_dev(const pfhe_rns *base, const uint64_t *multi_residues_dev,
                                            size_t len_in, uint64_t *big_uint_values_dev,
                                            size_t len_out, size_t value_count, void *stream);
/* wrapping_decompose_small_values_to — base.rs:279-312 (+ :721-730): centred lift of u in
 * [0, small_value_modulus): u < ceil(m/2) ? u : q_i - m + u ; m == 2 copies. */
int pfhe_rns_wrapping_decompose_small_values_to(const pfhe_rns *base, const uint64_t *small_values,
                                                size_t value_count, uint64_t *multi_residues,
                                                size_t len_out, uint64_t small_value_modulus);
int pfhe_rns_wrapping_decompose_small_values_to_dev(const pfhe_rns *base,
                                                    const uint64_t *small_values_dev,
                                                    size_t value_count, uint64_t *multi_residues_dev,
                                                    size_t len_out, uint64_t small_value_modulus,
                                                    void *stream);
/* add_wrapping_decompose_small_values_scaled — base.rs:326-384 (+ slice::wrapping_decompose_chunk_scaled_to
 * :739-757): acc[i][c] = reduce_add(acc[i][c], factor_i * lift_i(small[c])) with the centred lift above (m == 2
 * takes the unsigned branch, base.rs:371-378); add_decompose_small_values_scaled — base.rs:398-416: the same
 * without the lift (= add_decompose_small_polynomial_scaled, :429-443).  `acc`: L*value_count words,
 * modulus-major, accumulated in place; `factors`: L ShoupFactor (value, quotient) pairs on the host. */
int pfhe_rns_add_wrapping_decompose_small_values_scaled(const pfhe_rns *base, const uint64_t *small_values,
                                                        size_t value_count, uint64_t *acc, size_t len_acc,
                                                        uint64_t small_value_modulus, const uint64_t *factors);
int pfhe_rns_add_wrapping_decompose_small_values_scaled_dev(const pfhe_rns *base, const uint64_t *small_values_dev,
                                                            size_t value_count, uint64_t *acc_dev, size_t len_acc,
                                                            uint64_t small_value_modulus, const uint64_t *factors,
                                                            void *stream);
int pfhe_rns_add_decompose_small_values_scaled(const pfhe_rns *base, const uint64_t *small_values, size_t value_count,
                                               uint64_t *acc, size_t len_acc, const uint64_t *factors);
int pfhe_rns_add_decompose_small_values_scaled_dev(const pfhe_rns *base, const uint64_t *small_values_dev,
                                                   size_t value_count, uint64_t *acc_dev, size_t len_acc,
                                                   const uint64_t *factors, void *stream);

/* =====================================================================================
 * BigUintApproxSignedBasis<u64> — primus_decompose/src/big_integer/basis.rs:17
 * ===================================================================================== */
typedef struct pfhe_basis pfhe_basis;

/* BigUintApproxSignedBasis::new(Q, log_basis, reverse_length, rns_base) — basis.rs:40-211.
 * reverse_length == 0 means None (full chain).  The reference asserts; we return BAD_ARGUMENT. */
int pfhe_basis_create(const pfhe_rns *base, uint32_t log_basis, size_t reverse_length,
                      pfhe_basis **out);
void pfhe_basis_destroy(pfhe_basis *basis);
size_t pfhe_basis_decompose_length(const pfhe_basis *basis); /* basis.rs:236 */
uint32_t pfhe_basis_log_basis(const pfhe_basis *basis);      /* :242 */
uint32_t pfhe_basis_drop_bits(const pfhe_basis *basis);      /* :248 */
uint64_t pfhe_basis_basis_value(const pfhe_basis *basis);    /* :224 */
/* scalar_iter (:283) = 2^(drop + j*log_basis) as big_uint_value_len limbs per level;
 * iter_scalar_residues (:266) = the same reduced modulo every RNS modulus. */
int pfhe_basis_scalars(const pfhe_basis *basis, uint64_t *out, size_t len);
int pfhe_basis_scalars_residue(const pfhe_basis *basis, uint64_t *out, size_t len);
/* init_value_carry_slice_inplace — basis.rs:326-367.  carries are one byte (0/1) per value. */
int pfhe_basis_init_value_carry_slice_inplace(const pfhe_basis *basis, uint64_t *values, size_t len,
                                              uint8_t *carries, size_t count);
int pfhe_basis_init_value_carry_slice_inplace_dev(const pfhe_basis *basis, uint64_t *values_dev,
                                                  size_t len, uint8_t *carries_dev, size_t count,
                                                  void *stream);
/* decomposer_iter().nth(level).unsigned_decompose_slice_to — big_integer/common.rs:309-325
 * (-> unsigned_decompose_to :275-285); level 0 is the least significant digit. */
int pfhe_basis_unsigned_decompose_slice_to(const pfhe_basis *basis, size_t level,
                                           const uint64_t *values, size_t len, uint64_t *digits,
                                           uint8_t *carries, size_t count);
int pfhe_basis_unsigned_decompose_slice_to_dev(const pfhe_basis *basis, size_t level,
                                               const uint64_t *values_dev, size_t len,
                                               uint64_t *digits_dev, uint8_t *carries_dev,
                                               size_t count, void *stream);
/* init_value_carry_slice_to — basis.rs:371-420: the out-of-place form (input untouched). */
int pfhe_basis_init_value_carry_slice_to(const pfhe_basis *basis, const uint64_t *values, size_t len,
                                         uint64_t *adjusted_values, uint8_t *carries, size_t count);
int pfhe_basis_init_value_carry_slice_to_dev(const pfhe_basis *basis, const uint64_t *values_dev, size_t len,
                                             uint64_t *adjusted_values_dev, uint8_t *carries_dev, size_t count,
                                             void *stream);
/* decomposer_iter().nth(level).decompose_slice_to — big_integer/common.rs:289-306 (-> decompose_to :255-272): the
 * SIGNED digit as a residue modulo Q, big_uint_value_len limbs per value (a negative digit d is stored as Q + d);
 * len_out must equal len; input and output must be distinct buffers. */
int pfhe_basis_decompose_slice_to(const pfhe_basis *basis, size_t level, const uint64_t *values, size_t len,
                                  uint64_t *decomposed_values, size_t len_out, uint8_t *carries, size_t count);
int pfhe_basis_decompose_slice_to_dev(const pfhe_basis *basis, size_t level, const uint64_t *values_dev, size_t len,
                                      uint64_t *decomposed_values_dev, size_t len_out, uint8_t *carries_dev,
                                      size_t count, void *stream);

/* =====================================================================================
 * RNS gadget external product — primus_lattice
 * ===================================================================================== */
typedef struct pfhe_extprod_plan pfhe_extprod_plan;

/* Bundles what the reference passes as (&BigUintApproxSignedBasis, &Table, &RNSBase,
 * &mut DcrtGlevContext) — glwe/crt.rs:200-212, context/glev.rs:4-68.  The plan borrows `table`
 * (which must outlive it) and owns device scratch for `chunk` ciphertexts (0 = default: about 1 GiB of digit polynomials
 * per buffer, at least 64 and at most 65536 ciphertexts):
 * one buffer of chunk*(k+1)*ell*L*N words (+ chunk*(k+1)*ell*N balanced digits: int32 for log_basis <= 31, else int64).
 * Like `&mut DcrtGlevContext` (context/glev.rs:4-10) the plan has ONE holder at a time, and that is enforced: every
 * pfhe_extprod_* call takes the plan for its duration, and a call from a second thread meanwhile returns
 * PFHE_ERR_BUSY ("plan in use") instead of racing on the digit buffers (pfhe_extprod_plan_in_use reports the
 * flag; one plan per thread).  Streams: device-pointer calls return when their kernels are queued; the plan remembers an event
 * behind its last call and a call on a DIFFERENT stream first makes that stream wait for it, so using one plan from one
 * stream after another needs no event handling by the caller (round 5; until then this was the caller's job).  Work
 * captured into a HIP graph is outside that bookkeeping: do not replay a graph that uses a plan beside other users of it. */
int pfhe_extprod_plan_create(const pfhe_dcrt *table, const pfhe_rns *base, const pfhe_basis *basis,
                             size_t glwe_dimension, size_t chunk, pfhe_extprod_plan **out);
void pfhe_extprod_plan_destroy(pfhe_extprod_plan *plan);
int pfhe_extprod_plan_in_use(const pfhe_extprod_plan *plan);  /* 1 while some thread is inside a pfhe_extprod_* call on it */
/* Test aid, not a reference interface: hold != 0 takes the plan for the calling thread exactly as an entry point does
 * (PFHE_ERR_BUSY when another thread holds it) and keeps it until the same thread calls with hold == 0. */
int pfhe_extprod_plan_debug_hold(pfhe_extprod_plan *plan, int hold);
size_t pfhe_extprod_plan_scratch_bytes(const pfhe_extprod_plan *plan);
/* CrtGlwe::mul_dcrt_ggsw_to — glwe/crt.rs:200-227.  crt_glwe: batch x (k+1) CRT polynomials
 * |a1|..|ak|b|; dcrt_ggsw: ONE GGSW shared by the batch or batch GGSWs, each
 * (k+1) rows x ell levels x (k+1) components x L x N words; result: batch DcrtGlwe (NTT form), or
 * coefficient form when into_coeff_form != 0 (DcrtGlwe::into_coeff_form, macros/mod.rs:901-911). */
int pfhe_extprod_mul_dcrt_ggsw_to(pfhe_extprod_plan *plan, const uint64_t *crt_glwe, size_t len_glwe,
                                  const uint64_t *dcrt_ggsw, size_t len_ggsw, uint64_t *result,
                                  size_t len_result, int into_coeff_form);
int pfhe_extprod_mul_dcrt_ggsw_to_dev(pfhe_extprod_plan *plan, const uint64_t *crt_glwe_dev,
                                      size_t len_glwe, const uint64_t *dcrt_ggsw_dev, size_t len_ggsw,
                                      uint64_t *result_dev, size_t len_result, int into_coeff_form,
                                      void *stream);
/* Measurement aid (bench.py, tools/): the same product (coefficient-form output) with HIP events between its kernel
 * groups on `stream`; waits for the stream.  ms_out[0] = digit extraction + lifting strided pass, ms_out[1] = block pass of
 * the digits' transform + multiply-accumulate (+ the inverse transform's block pass, fused into the same kernel), both
 * summed over the *launches_out chunks; the inverse transform's strided pass runs after the last event.  Not a reference
 * interface. */
int pfhe_extprod_profile_dev(pfhe_extprod_plan *plan, const uint64_t *crt_glwe_dev, size_t len_glwe,
                             const uint64_t *dcrt_ggsw_dev, size_t len_ggsw, uint64_t *result_dev, size_t len_result,
                             double *ms_out, size_t *launches_out, void *stream);
/* DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign — glwe/dcrt.rs:178-255: acc += glev (x) crt_poly. */
int pfhe_extprod_add_dcrt_glev_mul_crt_poly_assign_dev(pfhe_extprod_plan *plan, uint64_t *acc_dev,
                                                       size_t len_acc, const uint64_t *dcrt_glev_dev,
                                                       size_t len_glev, const uint64_t *crt_poly_dev,
                                                       size_t len_poly, void *stream);

/* DcrtGlev::mul_crt_poly_to — primus_lattice/src/glev/dcrt.rs:45-110: result = glev (x) crt_poly
 * (one GGSW row; overwrites instead of accumulating). */
int pfhe_extprod_glev_mul_crt_poly_to_dev(pfhe_extprod_plan *plan, const uint64_t *dcrt_glev_dev,
                                          size_t len_glev, const uint64_t *crt_poly_dev, size_t len_poly,
                                          uint64_t *result_dev, size_t len_result, void *stream);
/* The same two products with the polynomial given as a BigUintPolynomial (big_uint_value_len limbs per
 * coefficient, canonical modulo Q): DcrtGlwe::add_dcrt_glev_mul_big_uint_poly_assign — glwe/dcrt.rs:258-338 — and
 * DcrtGlev::mul_big_uint_poly_to — glev/dcrt.rs:113-175.  len_poly = batch * big_uint_value_len * N. */
int pfhe_extprod_add_dcrt_glev_mul_big_uint_poly_assign_dev(pfhe_extprod_plan *plan, uint64_t *acc_dev, size_t len_acc,
                                                            const uint64_t *dcrt_glev_dev, size_t len_glev,
                                                            const uint64_t *big_uint_poly_dev, size_t len_poly,
                                                            void *stream);
int pfhe_extprod_glev_mul_big_uint_poly_to_dev(pfhe_extprod_plan *plan, const uint64_t *dcrt_glev_dev, size_t len_glev,
                                               const uint64_t *big_uint_poly_dev, size_t len_poly, uint64_t *result_dev,
                                               size_t len_result, void *stream);

/* Profiling hooks (bench.py / rocprofv3): a transform is executed as a short sequence of kernel
 * passes (DESIGN.md "Kernels"); these run or name ONE pass so that each kernel can be timed with
 * HIP events in isolation.  The data is only meaningful after all passes have run in order. */
int pfhe_dcrt_transform_num_passes(const pfhe_dcrt *table);
const char *pfhe_dcrt_transform_pass_name(const pfhe_dcrt *table, int inverse, int index);
int pfhe_dcrt_transform_pass_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len, int inverse,
                                 int index, int lazy, void *stream);
/* How pfhe_dcrt_transform_dev / pfhe_dcrt_inverse_transform_dev will run `len` words: the kernel (or form) name and
 * the number of kernel launches.  Large batches of N = 2^16 run as tiles + 1 launches of ntt_pipe_{fwd,inv}_kernel (block pass of
 * one tile and strided pass of the next in each workgroup), not as the per-pass kernels above. */
int pfhe_dcrt_transform_form(const pfhe_dcrt *table, size_t len, int inverse, char *name, size_t cap,
                             int *launches);

/* =====================================================================================
 * BaseConverter — primus_rns/src/converter.rs:21 — and RNSBase::decompose_big_uint_values_to
 * (primus_rns/src/base.rs:457-481).  Residue arrays are modulus-major, as in the reference; the
 * reference's coefficient-major `scratch` argument has no counterpart (the scaled residues stay
 * in registers).
 * ===================================================================================== */
typedef struct pfhe_conv pfhe_conv;

/* BaseConverter::new(input_base, output_base) — converter.rs:43-69 (the bases are copied) */
int pfhe_conv_create(const pfhe_rns *input_base, const pfhe_rns *output_base, pfhe_conv **out);
void pfhe_conv_destroy(pfhe_conv *conv);
size_t pfhe_conv_input_moduli_count(const pfhe_conv *conv);  /* converter.rs:82 */
size_t pfhe_conv_output_moduli_count(const pfhe_conv *conv); /* :87 */
/* row-major output-by-input matrix (Q/q_i) mod p_j — converter.rs:28-32 */
int pfhe_conv_base_change_matrix(const pfhe_conv *conv, uint64_t *out, size_t len);
/* fast_convert_array — converter.rs:192-218.  len_in = L_in * poly_length, len_out = L_out *
 * poly_length. */
int pfhe_conv_fast_convert_array(const pfhe_conv *conv, const uint64_t *crt_poly_in, size_t len_in,
                                 uint64_t *crt_poly_out, size_t len_out, size_t poly_length);
int pfhe_conv_fast_convert_array_dev(const pfhe_conv *conv, const uint64_t *crt_poly_in_dev,
                                     size_t len_in, uint64_t *crt_poly_out_dev, size_t len_out,
                                     size_t poly_length, void *stream);
/* fast_convert_array_to_pair_iter — converter.rs:233-272: two output moduli; `pairs_out_dev`
 * receives poly_length interleaved (mod p_0, mod p_1) pairs (len_out = 2 * poly_length) */
int pfhe_conv_fast_convert_array_to_pairs_dev(const pfhe_conv *conv, const uint64_t *crt_poly_in_dev,
                                              size_t len_in, uint64_t *pairs_out_dev,
                                              size_t len_out, size_t poly_length, void *stream);
/* exact_convert_array — converter.rs:274-364 (single output modulus; f64 correction term) */
int pfhe_conv_exact_convert_array(const pfhe_conv *conv, const uint64_t *crt_poly_in, size_t len_in,
                                  uint64_t *crt_poly_out, size_t len_out, size_t poly_length);
int pfhe_conv_exact_convert_array_dev(const pfhe_conv *conv, const uint64_t *crt_poly_in_dev,
                                      size_t len_in, uint64_t *crt_poly_out_dev, size_t len_out,
                                      size_t poly_length, void *stream);
/* RNSBase::decompose_big_uint_values_to — base.rs:457-481 */
int pfhe_rns_decompose_big_uint_values_to(const pfhe_rns *base, const uint64_t *big_uint_values,
                                          size_t len_in, uint64_t *multi_residues, size_t len_out,
                                          size_t value_count);
int pfhe_rns_decompose_big_uint_values_to_dev(const pfhe_rns *base,
                                              const uint64_t *big_uint_values_dev, size_t len_in,
                                              uint64_t *multi_residues_dev, size_t len_out,
                                              size_t value_count, void *stream);

/* =====================================================================================
 * u32 tables — U32NttTable (primus_ntt/src/ntt/prime32/table.rs:37-91, NttTable impl :184-470)
 * and U32DcrtTable (primus_ntt/src/dcrt/prime32.rs:11-128).  Same contracts as the 64-bit
 * tables with uint32_t words; q < 2^30 (table.rs:195-200 -> PFHE_ERR_MODULUS_TOO_LARGE).
 * Butterflies are the reference's Barrett-32 ones (prime32/scalar/arithmetic.rs:16-51).
 * ===================================================================================== */
typedef struct pfhe_ntt32 pfhe_ntt32;

/* NttTable::new — prime32/table.rs:184-333 */
int pfhe_ntt32_create(uint32_t log_n, uint32_t modulus, int device, pfhe_ntt32 **out);
void pfhe_ntt32_destroy(pfhe_ntt32 *table);
/* getters — table.rs:103-137 */
size_t pfhe_ntt32_poly_length(const pfhe_ntt32 *table);
uint32_t pfhe_ntt32_log_n(const pfhe_ntt32 *table);
uint32_t pfhe_ntt32_modulus(const pfhe_ntt32 *table);
uint32_t pfhe_ntt32_root(const pfhe_ntt32 *table);
uint32_t pfhe_ntt32_inv_root(const pfhe_ntt32 *table);
uint32_t pfhe_ntt32_inv_n(const pfhe_ntt32 *table);
int pfhe_ntt32_device(const pfhe_ntt32 *table);
/* transform_slice / inverse_transform_slice / lazy_* — table.rs:356-374 (host pointers) */
int pfhe_ntt32_transform_slice(const pfhe_ntt32 *table, uint32_t *poly, size_t len);
int pfhe_ntt32_inverse_transform_slice(const pfhe_ntt32 *table, uint32_t *values, size_t len);
int pfhe_ntt32_lazy_transform_slice(const pfhe_ntt32 *table, uint32_t *poly, size_t len);
int pfhe_ntt32_lazy_inverse_transform_slice(const pfhe_ntt32 *table, uint32_t *values, size_t len);
/* transform_monomial / transform_coeff_one_monomial / transform_coeff_minus_one_monomial —
 * table.rs:376-470 */
int pfhe_ntt32_transform_monomial(const pfhe_ntt32 *table, uint32_t coeff, size_t degree,
                                  uint32_t *values, size_t len);
int pfhe_ntt32_transform_coeff_one_monomial(const pfhe_ntt32 *table, size_t degree,
                                            uint32_t *values, size_t len);
int pfhe_ntt32_transform_coeff_minus_one_monomial(const pfhe_ntt32 *table, size_t degree,
                                                  uint32_t *values, size_t len);
/* device-pointer variants */
int pfhe_ntt32_transform_dev(const pfhe_ntt32 *table, uint32_t *poly_dev, size_t len, int lazy,
                             void *stream);
int pfhe_ntt32_inverse_transform_dev(const pfhe_ntt32 *table, uint32_t *values_dev, size_t len,
                                     int lazy, void *stream);
int pfhe_ntt32_transform_monomial_dev(const pfhe_ntt32 *table, uint32_t coeff, size_t degree,
                                      uint32_t *values_dev, size_t len, void *stream);
/* NttPolynomial<u32>::mul_assign / add_mul_assign (primus_poly/src/ntt/mul.rs:84-90,
 * ntt/mod.rs:101-112 with BarrettModulus<u32>) */
int pfhe_ntt32_mul_assign_dev(const pfhe_ntt32 *table, uint32_t *a_dev, size_t len_a,
                              const uint32_t *b_dev, size_t len_b, void *stream);
int pfhe_ntt32_add_mul_assign_dev(const pfhe_ntt32 *table, uint32_t *acc_dev,
                                  const uint32_t *a_dev, size_t len_a, const uint32_t *b_dev,
                                  size_t len_b, void *stream);

typedef struct pfhe_dcrt32 pfhe_dcrt32;

/* DcrtTable::new — dcrt/prime32.rs:24-43 */
int pfhe_dcrt32_create(uint32_t log_n, const uint32_t *moduli, size_t moduli_count, int device,
                       pfhe_dcrt32 **out);
void pfhe_dcrt32_destroy(pfhe_dcrt32 *table);
size_t pfhe_dcrt32_poly_length(const pfhe_dcrt32 *table);     /* dcrt/prime32.rs:56 */
size_t pfhe_dcrt32_moduli_count(const pfhe_dcrt32 *table);    /* :61 */
size_t pfhe_dcrt32_crt_poly_length(const pfhe_dcrt32 *table); /* :66 */
int pfhe_dcrt32_device(const pfhe_dcrt32 *table);
uint32_t pfhe_dcrt32_modulus(const pfhe_dcrt32 *table, size_t i);
uint32_t pfhe_dcrt32_root(const pfhe_dcrt32 *table, size_t i);
/* transform_slice / inverse_transform_slice / lazy_* — dcrt/prime32.rs:96-127 */
int pfhe_dcrt32_transform_slice(const pfhe_dcrt32 *table, uint32_t *poly, size_t len);
int pfhe_dcrt32_inverse_transform_slice(const pfhe_dcrt32 *table, uint32_t *poly, size_t len);
int pfhe_dcrt32_lazy_transform_slice(const pfhe_dcrt32 *table, uint32_t *poly, size_t len);
int pfhe_dcrt32_lazy_inverse_transform_slice(const pfhe_dcrt32 *table, uint32_t *poly, size_t len);
/* DcrtTable::transform_monomial & co — dcrt/mod.rs:107-134 */
int pfhe_dcrt32_transform_monomial(const pfhe_dcrt32 *table, uint32_t coeff, size_t degree,
                                   uint32_t *values, size_t len);
int pfhe_dcrt32_transform_coeff_one_monomial(const pfhe_dcrt32 *table, size_t degree,
                                             uint32_t *values, size_t len);
int pfhe_dcrt32_transform_coeff_minus_one_monomial(const pfhe_dcrt32 *table, size_t degree,
                                                   uint32_t *values, size_t len);
int pfhe_dcrt32_transform_dev(const pfhe_dcrt32 *table, uint32_t *poly_dev, size_t len, int lazy,
                              void *stream);
int pfhe_dcrt32_inverse_transform_dev(const pfhe_dcrt32 *table, uint32_t *poly_dev, size_t len,
                                      int lazy, void *stream);
/* DcrtPolynomial<u32>::mul_assign / add_mul_assign — primus_poly/src/dcrt/mul.rs:176-187,
 * dcrt/mod.rs:105-123 */
int pfhe_dcrt32_mul_assign_dev(const pfhe_dcrt32 *table, uint32_t *a_dev, size_t len_a,
                               const uint32_t *b_dev, size_t len_b, void *stream);
int pfhe_dcrt32_add_mul_assign_dev(const pfhe_dcrt32 *table, uint32_t *acc_dev,
                                   const uint32_t *a_dev, size_t len_a, const uint32_t *b_dev,
                                   size_t len_b, void *stream);
/* synthetic residues: word i = floor(splitmix64(seed, i) * q_limb(i) / 2^64) */
int pfhe_dcrt32_fill_uniform_dev(const pfhe_dcrt32 *table, uint32_t *dst_dev, size_t len,
                                 uint64_t seed, void *stream);
/* profiling hooks (one kernel launch per pass), as pfhe_dcrt_transform_pass_dev */
int pfhe_dcrt32_transform_num_passes(const pfhe_dcrt32 *table);
const char *pfhe_dcrt32_transform_pass_name(const pfhe_dcrt32 *table, int inverse, int index);
/* as pfhe_dcrt_transform_form: the kernel (or form) a transform of `len` words runs as, and its number of launches */
int pfhe_dcrt32_transform_form(const pfhe_dcrt32 *table, size_t len, int inverse, char *name, size_t cap, int *launches);
int pfhe_dcrt32_transform_pass_dev(const pfhe_dcrt32 *table, uint32_t *poly_dev, size_t len,
                                   int inverse, int index, int lazy, void *stream);

/* =====================================================================================
 * The <u32> instantiations of the same generics: RNSBase<u32, BarrettModulus<u32>> (primus_rns/src/base.rs:26-37),
 * BigUintApproxSignedBasis<u32> (primus_decompose/src/big_integer/basis.rs:33 — the type the reference's own
 * tests/big_uint.rs:13 runs) and CrtGlwe<u32>::mul_dcrt_ggsw_to over a U32DcrtTable (primus_lattice/src/glwe/crt.rs:200-227,
 * primus_ntt/src/dcrt/prime32.rs:11).  Same contracts as the 64-bit entry points above with uint32_t words: residues,
 * digits and the limbs of big integers are u32 in memory (big_uint_value_len counts u32 limbs: as many as Q needs,
 * big_integer.rs:675-686); moduli below 2^30; log_basis below 32.  Every output is the canonical integer the reference's
 * u32 arithmetic produces.
 * ===================================================================================== */
typedef struct pfhe_rns32 pfhe_rns32;
typedef struct pfhe_basis32 pfhe_basis32;
typedef struct pfhe_extprod32_plan pfhe_extprod32_plan;

/* RNSBase::new(moduli) — base.rs:79-117.  Errors: EMPTY_BASE (:47-49), COPRIME (:83-89),
 * UNREPRESENTABLE_MODULUS when a modulus is not in (1, 2^30) (BarrettModulus::<u32>::new,
 * primus_modulus/src/barrett/mod.rs:39-44), UNSUPPORTED for more than 32 moduli (bases of up to 8 moduli carry their
 * constants as kernel arguments, wider ones — up to 32 — in a device table owned by the handle and shared with the
 * bases, converters and plans derived from it; the same limit holds for both bases of a pfhe_conv). */
int pfhe_rns32_create(const uint32_t *moduli, size_t count, int device, pfhe_rns32 **out);
void pfhe_rns32_destroy(pfhe_rns32 *base);
size_t pfhe_rns32_moduli_count(const pfhe_rns32 *base);        /* base.rs:124 */
size_t pfhe_rns32_big_uint_value_len(const pfhe_rns32 *base);  /* base.rs:139 */
int pfhe_rns32_moduli_product(const pfhe_rns32 *base, uint32_t *out, size_t len); /* base.rs:133 */
/* compose_multiple_values_to — base.rs:648-675 (-> compose_to :609-633): residue i of value c is
 * multi_residues[i*value_count + c] (modulus-major); value c is written as big_uint_value_len
 * little-endian limbs at big_uint_values[c*big_uint_value_len], canonical in [0, Q). */
int pfhe_rns32_compose_multiple_values_to(const pfhe_rns32 *base, const uint32_t *multi_residues,
                                        size_t len_in, uint32_t *big_uint_values, size_t len_out,
                                        size_t value_count);
int pfhe_rns32_compose_multiple_values_to_dev(const pfhe_rns32 *base, const uint32_t *multi_residues_dev,
                                            size_t len_in, uint32_t *big_uint_values_dev,
                                            size_t len_out, size_t value_count, void *stream);
/* wrapping_decompose_small_values_to — base.rs:279-312 (+ :721-730): centred lift of u in
 * [0, small_value_modulus): u < ceil(m/2) ? u : q_i - m + u ; m == 2 copies. */
int pfhe_rns32_wrapping_decompose_small_values_to(const pfhe_rns32 *base, const uint32_t *small_values,
                                                size_t value_count, uint32_t *multi_residues,
                                                size_t len_out, uint32_t small_value_modulus);
int pfhe_rns32_wrapping_decompose_small_values_to_dev(const pfhe_rns32 *base,
                                                    const uint32_t *small_values_dev,
                                                    size_t value_count, uint32_t *multi_residues_dev,
                                                    size_t len_out, uint32_t small_value_modulus,
                                                    void *stream);
/* add_wrapping_decompose_small_values_scaled — base.rs:326-384 (+ slice::wrapping_decompose_chunk_scaled_to
 * :739-757): acc[i][c] = reduce_add(acc[i][c], factor_i * lift_i(small[c])) with the centred lift above (m == 2
 * takes the unsigned branch, base.rs:371-378); add_decompose_small_values_scaled — base.rs:398-416: the same
 * without the lift (= add_decompose_small_polynomial_scaled, :429-443).  `acc`: L*value_count words,
 * modulus-major, accumulated in place; `factors`: L ShoupFactor<u32> (value, quotient) pairs on the host (quotient = floor(value * 2^32 / q_i)). */
int pfhe_rns32_add_wrapping_decompose_small_values_scaled(const pfhe_rns32 *base, const uint32_t *small_values,
                                                        size_t value_count, uint32_t *acc, size_t len_acc,
                                                        uint32_t small_value_modulus, const uint32_t *factors);
int pfhe_rns32_add_wrapping_decompose_small_values_scaled_dev(const pfhe_rns32 *base, const uint32_t *small_values_dev,
                                                            size_t value_count, uint32_t *acc_dev, size_t len_acc,
                                                            uint32_t small_value_modulus, const uint32_t *factors,
                                                            void *stream);
int pfhe_rns32_add_decompose_small_values_scaled(const pfhe_rns32 *base, const uint32_t *small_values, size_t value_count,
                                               uint32_t *acc, size_t len_acc, const uint32_t *factors);
int pfhe_rns32_add_decompose_small_values_scaled_dev(const pfhe_rns32 *base, const uint32_t *small_values_dev,
                                                   size_t value_count, uint32_t *acc_dev, size_t len_acc,
                                                   const uint32_t *factors, void *stream);

/* BigUintApproxSignedBasis::new(Q, log_basis, reverse_length, rns_base) — basis.rs:40-211.
 * reverse_length == 0 means None (full chain); 0 < log_basis < 32 (:51).  The reference asserts; we return BAD_ARGUMENT. */
int pfhe_basis32_create(const pfhe_rns32 *base, uint32_t log_basis, size_t reverse_length,
                      pfhe_basis32 **out);
void pfhe_basis32_destroy(pfhe_basis32 *basis);
size_t pfhe_basis32_decompose_length(const pfhe_basis32 *basis); /* basis.rs:236 */
uint32_t pfhe_basis32_log_basis(const pfhe_basis32 *basis);      /* :242 */
uint32_t pfhe_basis32_drop_bits(const pfhe_basis32 *basis);      /* :248 */
uint32_t pfhe_basis32_basis_value(const pfhe_basis32 *basis);    /* :224 */
/* scalar_iter (:283) = 2^(drop + j*log_basis) as big_uint_value_len limbs per level;
 * iter_scalar_residues (:266) = the same reduced modulo every RNS modulus. */
int pfhe_basis32_scalars(const pfhe_basis32 *basis, uint32_t *out, size_t len);
int pfhe_basis32_scalars_residue(const pfhe_basis32 *basis, uint32_t *out, size_t len);
/* init_value_carry_slice_inplace — basis.rs:326-367.  carries are one byte (0/1) per value. */
int pfhe_basis32_init_value_carry_slice_inplace(const pfhe_basis32 *basis, uint32_t *values, size_t len,
                                              uint8_t *carries, size_t count);
int pfhe_basis32_init_value_carry_slice_inplace_dev(const pfhe_basis32 *basis, uint32_t *values_dev,
                                                  size_t len, uint8_t *carries_dev, size_t count,
                                                  void *stream);
/* decomposer_iter().nth(level).unsigned_decompose_slice_to — big_integer/common.rs:309-325
 * (-> unsigned_decompose_to :275-285); level 0 is the least significant digit. */
int pfhe_basis32_unsigned_decompose_slice_to(const pfhe_basis32 *basis, size_t level,
                                           const uint32_t *values, size_t len, uint32_t *digits,
                                           uint8_t *carries, size_t count);
int pfhe_basis32_unsigned_decompose_slice_to_dev(const pfhe_basis32 *basis, size_t level,
                                               const uint32_t *values_dev, size_t len,
                                               uint32_t *digits_dev, uint8_t *carries_dev,
                                               size_t count, void *stream);
/* init_value_carry_slice_to — basis.rs:371-420: the out-of-place form (input untouched). */
int pfhe_basis32_init_value_carry_slice_to(const pfhe_basis32 *basis, const uint32_t *values, size_t len,
                                         uint32_t *adjusted_values, uint8_t *carries, size_t count);
int pfhe_basis32_init_value_carry_slice_to_dev(const pfhe_basis32 *basis, const uint32_t *values_dev, size_t len,
                                             uint32_t *adjusted_values_dev, uint8_t *carries_dev, size_t count,
                                             void *stream);
/* decomposer_iter().nth(level).decompose_slice_to — big_integer/common.rs:289-306 (-> decompose_to :255-272): the
 * SIGNED digit as a residue modulo Q, big_uint_value_len limbs per value (a negative digit d is stored as Q + d);
 * len_out must equal len; input and output must be distinct buffers. */
int pfhe_basis32_decompose_slice_to(const pfhe_basis32 *basis, size_t level, const uint32_t *values, size_t len,
                                  uint32_t *decomposed_values, size_t len_out, uint8_t *carries, size_t count);
int pfhe_basis32_decompose_slice_to_dev(const pfhe_basis32 *basis, size_t level, const uint32_t *values_dev, size_t len,
                                      uint32_t *decomposed_values_dev, size_t len_out, uint8_t *carries_dev,
                                      size_t count, void *stream);
/* RNSBase::decompose_big_uint_values_to — base.rs:457-481 */
int pfhe_rns32_decompose_big_uint_values_to(const pfhe_rns32 *base, const uint32_t *big_uint_values,
                                          size_t len_in, uint32_t *multi_residues, size_t len_out,
                                          size_t value_count);
int pfhe_rns32_decompose_big_uint_values_to_dev(const pfhe_rns32 *base,
                                              const uint32_t *big_uint_values_dev, size_t len_in,
                                              uint32_t *multi_residues_dev, size_t len_out,
                                              size_t value_count, void *stream);

/* BaseConverter<u32> — primus_rns/src/converter.rs:21 (generic over T: FheUint): the contracts of pfhe_conv_* with
 * uint32_t residues.  reduce_dot_product folds a 64-bit accumulator every 16 terms there (common/compact/slice.rs:380-405);
 * the value reduced is the same integer, so every output is the canonical residue the reference produces; the exact form
 * rounds with `(sum + 0.5) as u32`. */
typedef struct pfhe_conv32 pfhe_conv32;
int pfhe_conv32_create(const pfhe_rns32 *input_base, const pfhe_rns32 *output_base, pfhe_conv32 **out);
void pfhe_conv32_destroy(pfhe_conv32 *conv);
size_t pfhe_conv32_input_moduli_count(const pfhe_conv32 *conv);
size_t pfhe_conv32_output_moduli_count(const pfhe_conv32 *conv);
int pfhe_conv32_base_change_matrix(const pfhe_conv32 *conv, uint32_t *out, size_t len);
int pfhe_conv32_fast_convert_array(const pfhe_conv32 *conv, const uint32_t *crt_poly_in, size_t len_in,
                                   uint32_t *crt_poly_out, size_t len_out, size_t poly_length);
int pfhe_conv32_fast_convert_array_dev(const pfhe_conv32 *conv, const uint32_t *crt_poly_in_dev, size_t len_in,
                                       uint32_t *crt_poly_out_dev, size_t len_out, size_t poly_length, void *stream);
int pfhe_conv32_fast_convert_array_to_pairs_dev(const pfhe_conv32 *conv, const uint32_t *crt_poly_in_dev, size_t len_in,
                                                uint32_t *pairs_out_dev, size_t len_out, size_t poly_length, void *stream);
int pfhe_conv32_exact_convert_array(const pfhe_conv32 *conv, const uint32_t *crt_poly_in, size_t len_in,
                                    uint32_t *crt_poly_out, size_t len_out, size_t poly_length);
int pfhe_conv32_exact_convert_array_dev(const pfhe_conv32 *conv, const uint32_t *crt_poly_in_dev, size_t len_in,
                                        uint32_t *crt_poly_out_dev, size_t len_out, size_t poly_length, void *stream);

/* The plan of the u32 product: as pfhe_extprod_plan_create (one holder at a time, `table` borrowed, device scratch for
 * `chunk` ciphertexts: chunk*(k+1)*ell*L*N u32 words; 0 = about 1 GiB, at least 128 ciphertexts). */
int pfhe_extprod32_plan_create(const pfhe_dcrt32 *table, const pfhe_rns32 *base, const pfhe_basis32 *basis,
                               size_t glwe_dimension, size_t chunk, pfhe_extprod32_plan **out);
void pfhe_extprod32_plan_destroy(pfhe_extprod32_plan *plan);
int pfhe_extprod32_plan_in_use(const pfhe_extprod32_plan *plan);
size_t pfhe_extprod32_plan_scratch_bytes(const pfhe_extprod32_plan *plan);
/* CrtGlwe::mul_dcrt_ggsw_to — glwe/crt.rs:200-227 (layouts as pfhe_extprod_mul_dcrt_ggsw_to) */
int pfhe_extprod32_mul_dcrt_ggsw_to(pfhe_extprod32_plan *plan, const uint32_t *crt_glwe, size_t len_glwe,
                                    const uint32_t *dcrt_ggsw, size_t len_ggsw, uint32_t *result, size_t len_result,
                                    int into_coeff_form);
int pfhe_extprod32_mul_dcrt_ggsw_to_dev(pfhe_extprod32_plan *plan, const uint32_t *crt_glwe_dev, size_t len_glwe,
                                        const uint32_t *dcrt_ggsw_dev, size_t len_ggsw, uint32_t *result_dev,
                                        size_t len_result, int into_coeff_form, void *stream);
/* DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign — glwe/dcrt.rs:178-255; DcrtGlev::mul_crt_poly_to — glev/dcrt.rs:45-110 */
int pfhe_extprod32_add_dcrt_glev_mul_crt_poly_assign_dev(pfhe_extprod32_plan *plan, uint32_t *acc_dev, size_t len_acc,
                                                         const uint32_t *dcrt_glev_dev, size_t len_glev,
                                                         const uint32_t *crt_poly_dev, size_t len_poly, void *stream);
int pfhe_extprod32_glev_mul_crt_poly_to_dev(pfhe_extprod32_plan *plan, const uint32_t *dcrt_glev_dev, size_t len_glev,
                                            const uint32_t *crt_poly_dev, size_t len_poly, uint32_t *result_dev,
                                            size_t len_result, void *stream);
/* DcrtGlwe::add_dcrt_glev_mul_big_uint_poly_assign — glwe/dcrt.rs:258-338; DcrtGlev::mul_big_uint_poly_to —
 * glev/dcrt.rs:113-175 (the polynomial as big_uint_value_len u32 limbs per coefficient) */
int pfhe_extprod32_add_dcrt_glev_mul_big_uint_poly_assign_dev(pfhe_extprod32_plan *plan, uint32_t *acc_dev, size_t len_acc,
                                                              const uint32_t *dcrt_glev_dev, size_t len_glev,
                                                              const uint32_t *big_uint_poly_dev, size_t len_poly,
                                                              void *stream);
int pfhe_extprod32_glev_mul_big_uint_poly_to_dev(pfhe_extprod32_plan *plan, const uint32_t *dcrt_glev_dev, size_t len_glev,
                                                 const uint32_t *big_uint_poly_dev, size_t len_poly, uint32_t *result_dev,
                                                 size_t len_result, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFHE_H */
