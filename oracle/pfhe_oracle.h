/*
 * pfhe_oracle.h — CPU restatement of the primus-fhe NTT / RNS / gadget / external-product path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as
 * the checker / the reported CPU baseline.  The product (primus-fhe_amd/csrc) never links,
 * includes or calls anything from this directory.
 *
 * PARITY STATUS: "parity unpinned" against reference *binary* output — the reference is Rust,
 * no cargo/rustc exists in the build container, and the reference's own tests hold no golden
 * vectors or known-answer tests for this path (every test draws unseeded random inputs and
 * cross-checks two implementations or a closed-form property).  What pins this
 * restatement instead (tests/test_oracle_*.py):
 *   - the closed-form cases the reference tests do spell out (CRT of (3,5,7) residues (2,3,2),
 *     the centred-lift rule on (97,101,103), Barrett/Shoup == u128 %),
 *   - the reference's own cross-implementation checks re-created here: U64NttTable == UintNttTable
 *     on canonical fwd/inv/monomial outputs, round trips, lazy ranges, Barrett-32 == Barrett-64,
 *   - an independent Python big-integer evaluation (schoolbook negacyclic product, direct
 *     evaluation at psi^(2*brv(i)+1)) — tests/pyref.py,
 *   - the survey's independently computed minimal roots.
 * Every canonical output on this path is a mathematically determined integer, so agreement of
 * these independent routes is the strongest pin available without a Rust toolchain.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/crates/).
 */
#ifndef PFHE_ORACLE_H
#define PFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes mirror primus_ntt/src/error.rs:7-49 (NttError) and primus_rns RNSError. */
enum orc_status {
    ORC_OK = 0,
    ORC_ERR_NO_PRIMITIVE_ROOT = 1,
    ORC_ERR_DEGREE_CONVERSION = 2,
    ORC_ERR_DEGREE_TOO_LARGE = 3,
    ORC_ERR_NTT_TABLE = 4,
    ORC_ERR_MODULUS_TOO_LARGE = 5,
    ORC_ERR_EMPTY_BASE = 16,
    ORC_ERR_COPRIME = 17,
    ORC_ERR_BAD_ARG = 32
};

/* ---------------- scalar modular arithmetic ---------------- */
uint64_t orc_reduce_once(uint64_t x, uint64_t q);
uint64_t orc_reduce_twice(uint64_t x, uint64_t q, uint64_t two_q);
uint64_t orc_shoup_quotient(uint64_t w, uint64_t q);
/* MultiplyFactor::new(w, shift, q).quotient() = floor(w * 2^shift / q) (primus_factor/src/mul_factor/mod.rs:18) */
uint64_t orc_multiply_factor_quotient(uint64_t w, uint32_t shift, uint64_t q);
uint64_t orc_mul_mod_lazy(uint64_t y, uint64_t w, uint64_t w_precon, uint64_t q);
uint64_t orc_mul_mod_lazy32(uint64_t y, uint64_t w, uint64_t w_precon32, uint64_t q);
uint64_t orc_shoup_mul(uint64_t w, uint64_t w_precon, uint64_t b, uint64_t q);

typedef struct {
    uint64_t value;
    uint64_t ratio[2]; /* floor(2^128 / value), little-endian */
} orc_barrett;

int orc_barrett_new(uint64_t value, orc_barrett *out);
uint64_t orc_barrett_lazy_reduce_wide(const orc_barrett *m, uint64_t lo, uint64_t hi);
uint64_t orc_barrett_reduce_wide(const orc_barrett *m, uint64_t lo, uint64_t hi);
uint64_t orc_barrett_reduce(const orc_barrett *m, uint64_t v);
uint64_t orc_barrett_mul(const orc_barrett *m, uint64_t a, uint64_t b);
uint64_t orc_barrett_mul_add(const orc_barrett *m, uint64_t a, uint64_t b, uint64_t c);
uint64_t orc_reduce_add(uint64_t q, uint64_t a, uint64_t b);
uint64_t orc_reduce_sub(uint64_t q, uint64_t a, uint64_t b);
uint64_t orc_pow_mod(uint64_t base, uint64_t exp, uint64_t q);
uint64_t orc_inv_mod(uint64_t a, uint64_t q);

/* slice kernels: primus_modulus/src/common/compact/slice.rs:106-115,210-221 */
void orc_reduce_mul_slice_assign(uint64_t q, uint64_t *a, const uint64_t *b, size_t n);
void orc_reduce_add_mul_slice_assign(uint64_t q, uint64_t *acc, const uint64_t *a,
                                     const uint64_t *b, size_t n);

/* ---------------- primitive root ---------------- */
int orc_minimal_primitive_root(uint32_t log_degree, uint64_t q, uint64_t *root);

/* ---------------- U64NttTable (prime64/table.rs + scalar/) ---------------- */
typedef struct orc_u64_ntt orc_u64_ntt;

int orc_u64_ntt_new(uint32_t log_n, uint64_t q, orc_u64_ntt **out);
void orc_u64_ntt_free(orc_u64_ntt *t);
size_t orc_u64_ntt_n(const orc_u64_ntt *t);
uint64_t orc_u64_ntt_modulus(const orc_u64_ntt *t);
uint64_t orc_u64_ntt_root(const orc_u64_ntt *t);
uint64_t orc_u64_ntt_inv_root(const orc_u64_ntt *t);
uint64_t orc_u64_ntt_inv_n(const orc_u64_ntt *t);
uint64_t orc_u64_ntt_inv_n_w(const orc_u64_ntt *t);
const uint64_t *orc_u64_ntt_roots(const orc_u64_ntt *t);
const uint64_t *orc_u64_ntt_roots_precon64(const orc_u64_ntt *t);
const uint64_t *orc_u64_ntt_roots_precon32(const orc_u64_ntt *t);     /* null unless q < 2^30 */
const uint64_t *orc_u64_ntt_inv_roots_precon32(const orc_u64_ntt *t);
const uint64_t *orc_u64_ntt_roots_precon52(const orc_u64_ntt *t);     /* null unless q < 2^50 */
const uint64_t *orc_u64_ntt_inv_roots_precon52(const orc_u64_ntt *t); /* null unless q < 2^50 */
const uint64_t *orc_u64_ntt_inv_roots(const orc_u64_ntt *t);
const uint64_t *orc_u64_ntt_inv_roots_precon64(const orc_u64_ntt *t);
const uint64_t *orc_u64_ntt_ordinal_roots(const orc_u64_ntt *t);

/* bit_shift: 0 = table's own dispatch (32 when q < 2^30 else 64), or force 32 / 64. */
void orc_u64_ntt_scalar_forward(const orc_u64_ntt *t, uint64_t *values, uint32_t bit_shift,
                                uint32_t output_mod_factor);
void orc_u64_ntt_scalar_inverse(const orc_u64_ntt *t, uint64_t *values, uint32_t bit_shift,
                                uint32_t output_mod_factor);
void orc_u64_ntt_transform_slice(const orc_u64_ntt *t, uint64_t *poly);
void orc_u64_ntt_inverse_transform_slice(const orc_u64_ntt *t, uint64_t *values);
void orc_u64_ntt_lazy_transform_slice(const orc_u64_ntt *t, uint64_t *poly);
void orc_u64_ntt_lazy_inverse_transform_slice(const orc_u64_ntt *t, uint64_t *values);
void orc_u64_ntt_transform_monomial(const orc_u64_ntt *t, uint64_t coeff, size_t degree,
                                    uint64_t *values);
void orc_u64_ntt_transform_coeff_one_monomial(const orc_u64_ntt *t, size_t degree,
                                              uint64_t *values);
void orc_u64_ntt_transform_coeff_minus_one_monomial(const orc_u64_ntt *t, size_t degree,
                                                    uint64_t *values);

/* ---------------- UintNttTable<u64> (ntt/primitive.rs) — second implementation -------- */
typedef struct orc_uint_ntt orc_uint_ntt;
int orc_uint_ntt_new(uint32_t log_n, uint64_t q, orc_uint_ntt **out);
void orc_uint_ntt_free(orc_uint_ntt *t);
void orc_uint_ntt_transform_slice(const orc_uint_ntt *t, uint64_t *poly);
void orc_uint_ntt_inverse_transform_slice(const orc_uint_ntt *t, uint64_t *values);
void orc_uint_ntt_lazy_transform_slice(const orc_uint_ntt *t, uint64_t *poly);
void orc_uint_ntt_lazy_inverse_transform_slice(const orc_uint_ntt *t, uint64_t *values);
void orc_uint_ntt_transform_monomial(const orc_uint_ntt *t, uint64_t coeff, size_t degree,
                                     uint64_t *values);

/* ---------------- U64DcrtTable (dcrt/prime64.rs) ---------------- */
/* AVX-512 (DQ) backend of the forward transform (pfhe_oracle_avx512.c; prime64/avx512/), n >= 16.
 * Returns ORC_ERR_BAD_ARG when the host lacks AVX-512 DQ. */
int orc_avx512_available(void);
/* process-wide switch (default off): canonical forward transforms of every U64NttTable go through the
 * AVX-512 backend when the host has one (bench.py's cpu_baseline only) */
void orc_set_vector_backend(int on);
int orc_get_vector_backend(void);
/* dispatch as the reference does (table.rs:166-302): IFMA (BIT_SHIFT = 52) when the CPU has it and q < 2^50, else DQ */
int orc_u64_ntt_forward_avx512(const orc_u64_ntt *t, uint64_t *values, int lazy);
int orc_u64_ntt_inverse_avx512(const orc_u64_ntt *t, uint64_t *values, int lazy);
/* shift = 64 / 32 force the DQ rungs (32: q < 2^30), 52 the IFMA rung (q < 2^50; ORC_ERR_BAD_ARG when the CPU lacks it or
 * the modulus is too wide), 0 dispatches */
int orc_avx512_ifma_available(void);
int orc_u64_ntt_forward_avx512_shift(const orc_u64_ntt *t, uint64_t *values, int lazy, int shift);
int orc_u64_ntt_inverse_avx512_shift(const orc_u64_ntt *t, uint64_t *values, int lazy, int shift);
int orc_u64_ntt_forward_avx512_batch(const orc_u64_ntt *t, uint64_t *values, size_t count, int lazy, int shift);

/* ---------------- U32NttTable (prime32/table.rs, prime32/scalar/) ---------------- */
typedef struct orc_u32_ntt orc_u32_ntt;
uint32_t orc_u32_mul_mod_lazy(uint32_t y, uint32_t w, uint32_t w_precon, uint32_t q);
int orc_u32_ntt_new(uint32_t log_n, uint32_t q, orc_u32_ntt **out);
void orc_u32_ntt_free(orc_u32_ntt *t);
size_t orc_u32_ntt_n(const orc_u32_ntt *t);
uint32_t orc_u32_ntt_modulus(const orc_u32_ntt *t);
uint32_t orc_u32_ntt_root(const orc_u32_ntt *t);
uint32_t orc_u32_ntt_inv_root(const orc_u32_ntt *t);
uint32_t orc_u32_ntt_inv_n(const orc_u32_ntt *t);
uint32_t orc_u32_ntt_inv_n_w(const orc_u32_ntt *t);
const uint32_t *orc_u32_ntt_roots(const orc_u32_ntt *t);
const uint32_t *orc_u32_ntt_inv_roots(const orc_u32_ntt *t);
void orc_u32_ntt_scalar_forward(const orc_u32_ntt *t, uint32_t *values, uint32_t output_mod_factor);
void orc_u32_ntt_scalar_inverse(const orc_u32_ntt *t, uint32_t *values, uint32_t output_mod_factor);
void orc_u32_ntt_transform_slice(const orc_u32_ntt *t, uint32_t *poly);
void orc_u32_ntt_inverse_transform_slice(const orc_u32_ntt *t, uint32_t *values);
void orc_u32_ntt_lazy_transform_slice(const orc_u32_ntt *t, uint32_t *poly);
void orc_u32_ntt_lazy_inverse_transform_slice(const orc_u32_ntt *t, uint32_t *values);
void orc_u32_ntt_transform_monomial(const orc_u32_ntt *t, uint32_t coeff, size_t degree, uint32_t *values);
void orc_u32_ntt_transform_coeff_one_monomial(const orc_u32_ntt *t, size_t degree, uint32_t *values);
void orc_u32_ntt_transform_coeff_minus_one_monomial(const orc_u32_ntt *t, size_t degree, uint32_t *values);
void orc_u32_reduce_mul_slice_assign(uint32_t q, uint32_t *a, const uint32_t *b, size_t n);
void orc_u32_reduce_add_mul_slice_assign(uint32_t q, uint32_t *acc, const uint32_t *a, const uint32_t *b, size_t n);

typedef struct orc_dcrt orc_dcrt;
int orc_dcrt_new(uint32_t log_n, const uint64_t *moduli, size_t count, orc_dcrt **out);
void orc_dcrt_free(orc_dcrt *t);
size_t orc_dcrt_poly_length(const orc_dcrt *t);
size_t orc_dcrt_moduli_count(const orc_dcrt *t);
const orc_u64_ntt *orc_dcrt_table(const orc_dcrt *t, size_t i);
void orc_dcrt_transform_slice(const orc_dcrt *t, uint64_t *poly);
void orc_dcrt_inverse_transform_slice(const orc_dcrt *t, uint64_t *poly);
/* DcrtPolynomial::mul_assign / add_mul_assign (primus_poly/src/dcrt/mul.rs:176, mod.rs:105) */
void orc_dcrt_poly_mul_assign(const orc_dcrt *t, uint64_t *a, const uint64_t *b);
void orc_dcrt_poly_add_mul_assign(const orc_dcrt *t, uint64_t *acc, const uint64_t *a,
                                  const uint64_t *b);

/* GLWE butterfly (a, b) = (a + s, (a - s) * w): DcrtPolynomial::butterfly_mul_factor_to
 * (primus_poly/src/dcrt/mul.rs:15-30,196-222; w = L*N ShoupFactor pairs) and butterfly_mul_to
 * (primus_poly/src/dcrt/mod.rs:125-160; w = plain residues). */
void orc_dcrt_poly_butterfly_mul_factor_to(const orc_dcrt *t, uint64_t *a, const uint64_t *s,
                                           const uint64_t *w_pairs, uint64_t *b);
void orc_dcrt_poly_butterfly_mul_to(const orc_dcrt *t, uint64_t *a, const uint64_t *s, const uint64_t *w,
                                    uint64_t *b);

/* ---------------- CrtPolynomial / DcrtPolynomial element-wise family ----------------------
 * (primus_poly/src/crt/{add,sub,neg,mul}.rs, dcrt/inv.rs; CrtGlwe loops the same slices).  One RNS
 * polynomial of L limbs x n words per call; `factors` = L (value, quotient) pairs. */
uint64_t orc_reduce_neg(uint64_t q, uint64_t v);
void orc_crt_poly_add_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, const uint64_t *b,
                         uint64_t *out);
void orc_crt_poly_sub_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, const uint64_t *b,
                         uint64_t *out);
void orc_crt_poly_neg_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, uint64_t *out);
int orc_crt_poly_mul_scalar_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a,
                               const uint64_t *scalars, uint64_t *out);
int orc_crt_poly_add_mul_scalar_assign(const uint64_t *moduli, size_t L, size_t n, uint64_t *acc,
                                       const uint64_t *rhs, const uint64_t *scalars);
void orc_crt_poly_mul_factor_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a,
                                const uint64_t *factors, uint64_t *out);
void orc_crt_poly_add_mul_factor_assign(const uint64_t *moduli, size_t L, size_t n, uint64_t *acc,
                                        const uint64_t *rhs, const uint64_t *factors);
int orc_crt_poly_mul_monomial_assign(const uint64_t *moduli, size_t L, size_t n, uint64_t *data, size_t r);
int orc_dcrt_poly_inv_to(const uint64_t *moduli, size_t L, size_t n, const uint64_t *a, uint64_t *out);

/* ---------------- schoolbook negacyclic product (primus_poly/src/poly/mul.rs:107-134) --- */
void orc_naive_negacyclic_mul(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out,
                              size_t n);

/* ---------------- RNSBase<u64, BarrettModulus> (primus_rns/src/base.rs) ---------------- */
typedef struct orc_rns orc_rns;
int orc_rns_new(const uint64_t *moduli, size_t count, orc_rns **out);
void orc_rns_free(orc_rns *b);
size_t orc_rns_moduli_count(const orc_rns *b);
size_t orc_rns_value_len(const orc_rns *b);
const uint64_t *orc_rns_moduli_product(const orc_rns *b);
const uint64_t *orc_rns_punctured_product(const orc_rns *b);
void orc_rns_compose_to(const orc_rns *b, const uint64_t *residues, uint64_t *value);
void orc_rns_compose_multiple_values_to(const orc_rns *b, const uint64_t *multi_residues,
                                        uint64_t *big_uint_values, size_t value_count);
void orc_rns_decompose_to(const orc_rns *b, const uint64_t *value, uint64_t *residues);
void orc_rns_decompose_big_uint_values_to(const orc_rns *b, const uint64_t *big_uint_values,
                                          uint64_t *multi_residues, size_t value_count);
void orc_rns_wrapping_decompose_small_values_to(const orc_rns *b, const uint64_t *small_values,
                                                uint64_t *multi_residues, size_t value_count,
                                                uint64_t small_value_modulus);
void orc_rns_add_wrapping_decompose_small_values_scaled(const orc_rns *b, const uint64_t *small_values, uint64_t *acc,
                                                        size_t value_count, uint64_t small_value_modulus,
                                                        const uint64_t *factors);
void orc_rns_add_decompose_small_values_scaled(const orc_rns *b, const uint64_t *small_values, uint64_t *acc,
                                               size_t value_count, const uint64_t *factors);

/* ---------------- BigUintApproxSignedBasis<u64> (primus_decompose/src/big_integer) ----- */
typedef struct orc_basis orc_basis;
/* reverse_length == 0 means None (full chain). */
int orc_basis_new(const orc_rns *rns, uint32_t log_basis, size_t reverse_length,
                  orc_basis **out);
void orc_basis_free(orc_basis *b);
size_t orc_basis_decompose_length(const orc_basis *b);
uint32_t orc_basis_log_basis(const orc_basis *b);
uint32_t orc_basis_drop_bits(const orc_basis *b);
uint64_t orc_basis_basis_value(const orc_basis *b);
/* init mode: 0 Plain, 1 CarryOnly, 2 AdjustOnly, 3 AdjustAndCarry */
int orc_basis_init_mode(const orc_basis *b);
const uint64_t *orc_basis_threshold(const orc_basis *b);
const uint64_t *orc_basis_adjust_add(const orc_basis *b);
const uint64_t *orc_basis_scalars(const orc_basis *b);         /* ell * value_len */
const uint64_t *orc_basis_scalars_residue(const orc_basis *b); /* ell * moduli_count */
void orc_basis_init_value_carry_slice_inplace(const orc_basis *b, uint64_t *values,
                                              uint8_t *carries, size_t count);
void orc_basis_unsigned_decompose_slice_to(const orc_basis *b, size_t level,
                                           const uint64_t *values, uint64_t *digits,
                                           uint8_t *carries, size_t count);
void orc_basis_init_value_carry_slice_to(const orc_basis *b, const uint64_t *values, uint64_t *adjusted,
                                         uint8_t *carries, size_t count);
void orc_basis_decompose_slice_to(const orc_basis *b, size_t level, const uint64_t *values, uint64_t *decomposed,
                                  uint8_t *carries, size_t count);

/* ---------------- BaseConverter (primus_rns/src/converter.rs) ---------------- */
typedef struct orc_conv orc_conv;
int orc_conv_new(const orc_rns *in, const orc_rns *out, orc_conv **res); /* borrows both bases */
void orc_conv_free(orc_conv *c);
const uint64_t *orc_conv_matrix(const orc_conv *c);
void orc_conv_fast_convert(const orc_conv *c, const uint64_t *residues_in, uint64_t *residues_out,
                           uint64_t *scratch);
void orc_conv_fast_convert_array(const orc_conv *c, const uint64_t *crt_poly_in, uint64_t *crt_poly_out,
                                 size_t poly_length, uint64_t *scratch);
int orc_conv_exact_convert_array(const orc_conv *c, const uint64_t *crt_poly_in, uint64_t *crt_poly_out,
                                 size_t poly_length);

/* ---------------- RNS gadget external product (primus_lattice) ---------------- */
/* DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign, glwe/dcrt.rs:178-255.
 * acc: (k+1)*L*N words (DcrtGlwe), glev: ell*(k+1)*L*N words, crt_poly: L*N words. */
void orc_add_dcrt_glev_mul_crt_poly_assign(const orc_dcrt *table, const orc_rns *rns,
                                           const orc_basis *basis, size_t glwe_dimension,
                                           uint64_t *acc, const uint64_t *dcrt_glev,
                                           const uint64_t *crt_poly);
/* CrtGlwe::mul_dcrt_ggsw_to, glwe/crt.rs:200-227.  result stays in DCRT (NTT) form. */
void orc_add_dcrt_glev_mul_big_uint_poly_assign(const orc_dcrt *table, const orc_rns *rns, const orc_basis *basis,
                                                size_t k, uint64_t *acc, const uint64_t *dcrt_glev,
                                                const uint64_t *big_uint_poly);
void orc_mul_dcrt_ggsw_to(const orc_dcrt *table, const orc_rns *rns, const orc_basis *basis,
                          size_t glwe_dimension, const uint64_t *crt_glwe,
                          const uint64_t *dcrt_ggsw, uint64_t *result);

/* ---------------- the <u32> instantiations: RNSBase<u32>, BigUintApproxSignedBasis<u32>, the external product over
 * U32DcrtTable (pfhe_oracle_rns32.c: 32-bit limb arithmetic throughout, written against the reference's generic source) */
typedef struct orc_rns32 orc_rns32;
typedef struct orc_basis32 orc_basis32;
int orc_rns32_new(const uint32_t *moduli, size_t count, orc_rns32 **out);
void orc_rns32_free(orc_rns32 *b);
size_t orc_rns32_moduli_count(const orc_rns32 *b);
size_t orc_rns32_value_len(const orc_rns32 *b);
const uint32_t *orc_rns32_moduli_product(const orc_rns32 *b);
const uint32_t *orc_rns32_punctured_product(const orc_rns32 *b);
void orc_rns32_compose_to(const orc_rns32 *b, const uint32_t *residues, uint32_t *value);
void orc_rns32_compose_multiple_values_to(const orc_rns32 *b, const uint32_t *multi_residues, uint32_t *big_uint_values,
                                          size_t value_count);
void orc_rns32_decompose_to(const orc_rns32 *b, const uint32_t *value, uint32_t *residues);
void orc_rns32_decompose_big_uint_values_to(const orc_rns32 *b, const uint32_t *big_uint_values, uint32_t *multi_residues,
                                            size_t value_count);
void orc_rns32_wrapping_decompose_small_values_to(const orc_rns32 *b, const uint32_t *small_values, uint32_t *multi_residues,
                                                  size_t value_count, uint32_t small_value_modulus);
void orc_rns32_add_wrapping_decompose_small_values_scaled(const orc_rns32 *b, const uint32_t *small_values, uint32_t *acc,
                                                          size_t value_count, uint32_t small_value_modulus,
                                                          const uint32_t *factors);
void orc_rns32_add_decompose_small_values_scaled(const orc_rns32 *b, const uint32_t *small_values, uint32_t *acc,
                                                 size_t value_count, const uint32_t *factors);
/* BaseConverter<u32> (converter.rs with T = u32); borrows both bases */
typedef struct orc_conv32 orc_conv32;
int orc_conv32_new(const orc_rns32 *in, const orc_rns32 *out, orc_conv32 **res);
void orc_conv32_free(orc_conv32 *c);
const uint32_t *orc_conv32_matrix(const orc_conv32 *c);
void orc_conv32_fast_convert_array(const orc_conv32 *c, const uint32_t *crt_poly_in, uint32_t *crt_poly_out,
                                   size_t poly_length, uint32_t *scratch);
int orc_conv32_exact_convert_array(const orc_conv32 *c, const uint32_t *crt_poly_in, uint32_t *crt_poly_out,
                                   size_t poly_length);
int orc_basis32_new(const orc_rns32 *rns, uint32_t log_basis, size_t reverse_length, orc_basis32 **out);
void orc_basis32_free(orc_basis32 *b);
size_t orc_basis32_decompose_length(const orc_basis32 *b);
uint32_t orc_basis32_log_basis(const orc_basis32 *b);
uint32_t orc_basis32_drop_bits(const orc_basis32 *b);
uint32_t orc_basis32_basis_value(const orc_basis32 *b);
int orc_basis32_init_mode(const orc_basis32 *b);
const uint32_t *orc_basis32_scalars(const orc_basis32 *b);
const uint32_t *orc_basis32_scalars_residue(const orc_basis32 *b);
void orc_basis32_init_value_carry_slice_inplace(const orc_basis32 *b, uint32_t *values, uint8_t *carries, size_t count);
void orc_basis32_init_value_carry_slice_to(const orc_basis32 *b, const uint32_t *values, uint32_t *adjusted, uint8_t *carries,
                                           size_t count);
void orc_basis32_unsigned_decompose_slice_to(const orc_basis32 *b, size_t level, const uint32_t *values, uint32_t *digits,
                                             uint8_t *carries, size_t count);
void orc_basis32_decompose_slice_to(const orc_basis32 *b, size_t level, const uint32_t *values, uint32_t *decomposed,
                                    uint8_t *carries, size_t count);
/* `tables`: one orc_u32_ntt per modulus of the base, in base order (U32DcrtTable, dcrt/prime32.rs:11) */
void orc_add_dcrt32_glev_mul_crt_poly_assign(const orc_u32_ntt *const *tables, const orc_rns32 *rns, const orc_basis32 *basis,
                                             size_t k, uint32_t *acc, const uint32_t *dcrt_glev, const uint32_t *crt_poly);
void orc_add_dcrt32_glev_mul_big_uint_poly_assign(const orc_u32_ntt *const *tables, const orc_rns32 *rns,
                                                  const orc_basis32 *basis, size_t k, uint32_t *acc, const uint32_t *dcrt_glev,
                                                  const uint32_t *big_uint_poly);
void orc_mul_dcrt32_ggsw_to(const orc_u32_ntt *const *tables, const orc_rns32 *rns, const orc_basis32 *basis, size_t k,
                            const uint32_t *crt_glwe, const uint32_t *dcrt_ggsw, uint32_t *result);

#ifdef __cplusplus
}
#endif
#endif
