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
 *     the measured hot path.
 *   - handles are immutable after creation and may be shared between host threads
 *     (NttTable: Send + Sync, primus_ntt/src/ntt/mod.rs:16); an external-product plan owns
 *     scratch and is NOT concurrently usable (it mirrors `&mut DcrtGlevContext`,
 *     primus_lattice/src/context/glev.rs:4-10).
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
    PFHE_ERR_UNSUPPORTED = 36
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
int pfhe_stream_synchronize(int device, void *stream);
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
/* Fused "NTT -> pointwise mul by dcrt_poly -> INTT" of CRT polynomials in place:
 * CrtRlwe::mul_dcrt_polynomial_to (primus_lattice/src/rlwe/crt.rs:42-65) followed by
 * DcrtRlwe::into_coeff_form (primus_lattice/src/macros/mod.rs:901-911) per CRT polynomial. */
int pfhe_dcrt_mul_dcrt_polynomial_dev(const pfhe_dcrt *table, uint64_t *crt_poly_dev,
                                      size_t len, const uint64_t *dcrt_poly_dev, size_t len_b,
                                      void *stream);

/* Profiling hooks (bench.py / rocprofv3): a transform is executed as a short sequence of kernel
 * passes (DESIGN.md "Kernels"); these run or name ONE pass so that each kernel can be timed with
 * HIP events in isolation.  The data is only meaningful after all passes have run in order. */
int pfhe_dcrt_transform_num_passes(const pfhe_dcrt *table);
const char *pfhe_dcrt_transform_pass_name(const pfhe_dcrt *table, int inverse, int index);
int pfhe_dcrt_transform_pass_dev(const pfhe_dcrt *table, uint64_t *poly_dev, size_t len, int inverse,
                                 int index, int lazy, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFHE_H */
