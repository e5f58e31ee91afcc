// pfhe.hpp — header-only C++17 mirror of the reference's operator interface over the C ABI
// (include/pfhe.h).  Same type names, method names and argument meaning as the Rust traits:
//   U64NttTable   — primus_ntt::NttTable for U64NttTable   (crates/primus_ntt/src/ntt/mod.rs:16-113)
//   U64DcrtTable  — primus_ntt::DcrtTable for U64DcrtTable (crates/primus_ntt/src/dcrt/mod.rs:19-135)
//   U32NttTable, U32DcrtTable — the u32 / low-q tables (ntt/prime32/table.rs, dcrt/prime32.rs)
//   RNSBase, BigUintApproxSignedBasis, DcrtGlevContext, mul_dcrt_ggsw_to (and RNSBase32, BaseConverter32, BigUintApproxSignedBasis32,
//   DcrtGlevContext32: the <u32> instantiation, u32 words in memory)
//                 — primus_rns / primus_decompose / primus_lattice entry points of the RNS
//                   gadget external product (crates/primus_lattice/src/glwe/crt.rs:200-227)
// Errors: constructors throw pfhe::Error carrying the pfhe_status (the reference returns
// Result<_, NttError>); in-place transforms throw on length mismatch where the reference
// debug_asserts.  Slices are (pointer, length-in-words) pairs, in place, like `&mut [u64]`.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pfhe.h"

namespace pfhe {

class Error : public std::runtime_error {
  public:
    Error(int status, const std::string &what) : std::runtime_error(what), status_(status) {}
    int status() const noexcept { return status_; }

  private:
    int status_;
};

inline void check(int status) {
    if (status != PFHE_OK) {
        std::string msg = pfhe_status_string(status);
        const char *detail = pfhe_last_error();
        if (detail && *detail) msg += std::string(": ") + detail;
        throw Error(status, msg);
    }
}

class U64NttTable {
  public:
    U64NttTable(uint32_t log_n, uint64_t modulus, int device = 0) { check(pfhe_ntt_create(log_n, modulus, device, &h_)); }
    ~U64NttTable() { pfhe_ntt_destroy(h_); }
    U64NttTable(U64NttTable &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    U64NttTable(const U64NttTable &) = delete;
    U64NttTable &operator=(const U64NttTable &) = delete;

    size_t poly_length() const { return pfhe_ntt_poly_length(h_); }
    size_t n() const { return poly_length(); }
    uint32_t log_n() const { return pfhe_ntt_log_n(h_); }
    uint64_t modulus() const { return pfhe_ntt_modulus(h_); }
    uint64_t root() const { return pfhe_ntt_root(h_); }
    uint64_t inv_root() const { return pfhe_ntt_inv_root(h_); }
    uint64_t inv_n() const { return pfhe_ntt_inv_n(h_); }

    void transform_slice(uint64_t *poly, size_t len) const { check(pfhe_ntt_transform_slice(h_, poly, len)); }
    void inverse_transform_slice(uint64_t *v, size_t len) const { check(pfhe_ntt_inverse_transform_slice(h_, v, len)); }
    void lazy_transform_slice(uint64_t *poly, size_t len) const { check(pfhe_ntt_lazy_transform_slice(h_, poly, len)); }
    void lazy_inverse_transform_slice(uint64_t *v, size_t len) const { check(pfhe_ntt_lazy_inverse_transform_slice(h_, v, len)); }
    void transform_monomial(uint64_t coeff, size_t degree, uint64_t *values, size_t len) const {
        check(pfhe_ntt_transform_monomial(h_, coeff, degree, values, len));
    }
    void transform_coeff_one_monomial(size_t degree, uint64_t *values, size_t len) const {
        check(pfhe_ntt_transform_coeff_one_monomial(h_, degree, values, len));
    }
    void transform_coeff_minus_one_monomial(size_t degree, uint64_t *values, size_t len) const {
        check(pfhe_ntt_transform_coeff_minus_one_monomial(h_, degree, values, len));
    }
    // device-resident batches (asynchronous on `stream`)
    void transform_dev(uint64_t *poly_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_ntt_transform_dev(h_, poly_dev, len, lazy, stream));
    }
    void inverse_transform_dev(uint64_t *v_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_ntt_inverse_transform_dev(h_, v_dev, len, lazy, stream));
    }
    const pfhe_ntt *handle() const { return h_; }

  private:
    pfhe_ntt *h_ = nullptr;
};

class U64DcrtTable {
  public:
    U64DcrtTable(uint32_t log_n, const std::vector<uint64_t> &moduli, int device = 0) {
        check(pfhe_dcrt_create(log_n, moduli.data(), moduli.size(), device, &h_));
    }
    ~U64DcrtTable() { pfhe_dcrt_destroy(h_); }
    U64DcrtTable(U64DcrtTable &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    U64DcrtTable(const U64DcrtTable &) = delete;
    U64DcrtTable &operator=(const U64DcrtTable &) = delete;

    size_t poly_length() const { return pfhe_dcrt_poly_length(h_); }
    size_t moduli_count() const { return pfhe_dcrt_moduli_count(h_); }
    size_t crt_poly_length() const { return pfhe_dcrt_crt_poly_length(h_); }
    uint64_t modulus(size_t i) const { return pfhe_dcrt_modulus(h_, i); }

    void transform_slice(uint64_t *poly, size_t len) const { check(pfhe_dcrt_transform_slice(h_, poly, len)); }
    void inverse_transform_slice(uint64_t *poly, size_t len) const { check(pfhe_dcrt_inverse_transform_slice(h_, poly, len)); }
    void lazy_transform_slice(uint64_t *poly, size_t len) const { check(pfhe_dcrt_lazy_transform_slice(h_, poly, len)); }
    void lazy_inverse_transform_slice(uint64_t *poly, size_t len) const { check(pfhe_dcrt_lazy_inverse_transform_slice(h_, poly, len)); }
    void transform_monomial(uint64_t coeff, size_t degree, uint64_t *values, size_t len) const {
        check(pfhe_dcrt_transform_monomial(h_, coeff, degree, values, len));
    }
    // device-pointer form (launches on `stream` only: capturable); minus_one selects -X^degree
    void transform_monomial_dev(uint64_t coeff, size_t degree, uint64_t *values_dev, size_t len, bool minus_one = false,
                                void *stream = nullptr) const {
        check(pfhe_dcrt_transform_monomial_dev(h_, coeff, degree, values_dev, len, minus_one ? 1 : 0, stream));
    }
    void transform_dev(uint64_t *poly_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_dcrt_transform_dev(h_, poly_dev, len, lazy, stream));
    }
    void inverse_transform_dev(uint64_t *poly_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_dcrt_inverse_transform_dev(h_, poly_dev, len, lazy, stream));
    }
    // DcrtPolynomial::mul_assign / add_mul_assign (crates/primus_poly/src/dcrt/mul.rs:176, mod.rs:105)
    void mul_assign_dev(uint64_t *a, size_t len_a, const uint64_t *b, size_t len_b, void *stream = nullptr) const {
        check(pfhe_dcrt_mul_assign_dev(h_, a, len_a, b, len_b, stream));
    }
    void add_mul_assign_dev(uint64_t *acc, const uint64_t *a, size_t len_a, const uint64_t *b, size_t len_b,
                            void *stream = nullptr) const {
        check(pfhe_dcrt_add_mul_assign_dev(h_, acc, a, len_a, b, len_b, stream));
    }
    // CrtRlwe::mul_dcrt_polynomial_to + into_coeff_form (crates/primus_lattice/src/rlwe/crt.rs:42-65)
    void mul_dcrt_polynomial_dev(uint64_t *crt_poly, size_t len, const uint64_t *dcrt_poly, size_t len_b,
                                 void *stream = nullptr) const {
        check(pfhe_dcrt_mul_dcrt_polynomial_dev(h_, crt_poly, len, dcrt_poly, len_b, stream));
    }
    // CrtPolynomial / DcrtPolynomial / CrtGlwe element-wise family (crates/primus_poly/src/crt/{add,sub,neg,mul}.rs,
    // dcrt/inv.rs; primus_lattice/src/macros/mod.rs:367-531, glwe/crt.rs:59-175).  `out` may alias `a`.
    void add_to_dev(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, void *stream = nullptr) const {
        check(pfhe_dcrt_add_to_dev(h_, a, b, out, len, stream));
    }
    void sub_to_dev(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, void *stream = nullptr) const {
        check(pfhe_dcrt_sub_to_dev(h_, a, b, out, len, stream));
    }
    void neg_to_dev(const uint64_t *a, uint64_t *out, size_t len, void *stream = nullptr) const {
        check(pfhe_dcrt_neg_to_dev(h_, a, out, len, stream));
    }
    void mul_scalar_to_dev(const uint64_t *a, const std::vector<uint64_t> &scalars, uint64_t *out, size_t len,
                           void *stream = nullptr) const {
        require_count(scalars.size(), moduli_count());
        check(pfhe_dcrt_mul_scalar_to_dev(h_, a, scalars.data(), out, len, stream));
    }
    void add_mul_scalar_assign_dev(uint64_t *acc, const uint64_t *rhs, const std::vector<uint64_t> &scalars, size_t len,
                                   void *stream = nullptr) const {
        require_count(scalars.size(), moduli_count());
        check(pfhe_dcrt_add_mul_scalar_assign_dev(h_, acc, rhs, scalars.data(), len, stream));
    }
    // factors: (value, quotient) per modulus, ShoupFactor (crates/primus_factor/src/shoup_factor/mod.rs:22)
    void mul_factor_to_dev(const uint64_t *a, const std::vector<uint64_t> &factors, uint64_t *out, size_t len,
                           void *stream = nullptr) const {
        require_count(factors.size(), 2 * moduli_count());
        check(pfhe_dcrt_mul_factor_to_dev(h_, a, factors.data(), out, len, stream));
    }
    void add_mul_factor_assign_dev(uint64_t *acc, const uint64_t *rhs, const std::vector<uint64_t> &factors, size_t len,
                                   void *stream = nullptr) const {
        require_count(factors.size(), 2 * moduli_count());
        check(pfhe_dcrt_add_mul_factor_assign_dev(h_, acc, rhs, factors.data(), len, stream));
    }
    void mul_monomial_to_dev(const uint64_t *a, size_t r, uint64_t *out, size_t len, void *stream = nullptr) const {
        check(pfhe_dcrt_mul_monomial_to_dev(h_, a, r, out, len, stream));
    }
    void mul_monomial_assign_dev(uint64_t *data, size_t r, size_t len, void *stream = nullptr) const {
        check(pfhe_dcrt_mul_monomial_assign_dev(h_, data, r, len, stream));
    }
    // throws Error(PFHE_ERR_NO_INVERSE) where the reference panics
    void inv_to_dev(const uint64_t *a, uint64_t *out, size_t len, void *stream = nullptr) const {
        check(pfhe_dcrt_inv_to_dev(h_, a, out, len, stream));
    }
    const pfhe_dcrt *handle() const { return h_; }

  private:
    static void require_count(size_t got, size_t want) {
        if (got != want) throw Error(PFHE_ERR_BAD_LENGTH, "expected one entry per modulus");
    }
    pfhe_dcrt *h_ = nullptr;
};

// primus_ntt::U32NttTable (crates/primus_ntt/src/ntt/prime32/table.rs:37) — q < 2^30, u32 data
class U32NttTable {
  public:
    U32NttTable(uint32_t log_n, uint32_t modulus, int device = 0) { check(pfhe_ntt32_create(log_n, modulus, device, &h_)); }
    ~U32NttTable() { pfhe_ntt32_destroy(h_); }
    U32NttTable(U32NttTable &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    U32NttTable(const U32NttTable &) = delete;
    U32NttTable &operator=(const U32NttTable &) = delete;

    size_t poly_length() const { return pfhe_ntt32_poly_length(h_); }
    size_t n() const { return poly_length(); }
    uint32_t log_n() const { return pfhe_ntt32_log_n(h_); }
    uint32_t modulus() const { return pfhe_ntt32_modulus(h_); }
    uint32_t root() const { return pfhe_ntt32_root(h_); }
    uint32_t inv_root() const { return pfhe_ntt32_inv_root(h_); }
    uint32_t inv_n() const { return pfhe_ntt32_inv_n(h_); }

    void transform_slice(uint32_t *poly, size_t len) const { check(pfhe_ntt32_transform_slice(h_, poly, len)); }
    void inverse_transform_slice(uint32_t *v, size_t len) const { check(pfhe_ntt32_inverse_transform_slice(h_, v, len)); }
    void lazy_transform_slice(uint32_t *poly, size_t len) const { check(pfhe_ntt32_lazy_transform_slice(h_, poly, len)); }
    void lazy_inverse_transform_slice(uint32_t *v, size_t len) const { check(pfhe_ntt32_lazy_inverse_transform_slice(h_, v, len)); }
    void transform_monomial(uint32_t coeff, size_t degree, uint32_t *values, size_t len) const {
        check(pfhe_ntt32_transform_monomial(h_, coeff, degree, values, len));
    }
    void transform_coeff_one_monomial(size_t degree, uint32_t *values, size_t len) const {
        check(pfhe_ntt32_transform_coeff_one_monomial(h_, degree, values, len));
    }
    void transform_coeff_minus_one_monomial(size_t degree, uint32_t *values, size_t len) const {
        check(pfhe_ntt32_transform_coeff_minus_one_monomial(h_, degree, values, len));
    }
    void transform_dev(uint32_t *poly_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_ntt32_transform_dev(h_, poly_dev, len, lazy, stream));
    }
    void inverse_transform_dev(uint32_t *v_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_ntt32_inverse_transform_dev(h_, v_dev, len, lazy, stream));
    }
    const pfhe_ntt32 *handle() const { return h_; }

  private:
    pfhe_ntt32 *h_ = nullptr;
};

// primus_ntt::U32DcrtTable (crates/primus_ntt/src/dcrt/prime32.rs:11)
class U32DcrtTable {
  public:
    U32DcrtTable(uint32_t log_n, const std::vector<uint32_t> &moduli, int device = 0) {
        check(pfhe_dcrt32_create(log_n, moduli.data(), moduli.size(), device, &h_));
    }
    ~U32DcrtTable() { pfhe_dcrt32_destroy(h_); }
    U32DcrtTable(U32DcrtTable &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    U32DcrtTable(const U32DcrtTable &) = delete;
    U32DcrtTable &operator=(const U32DcrtTable &) = delete;

    size_t poly_length() const { return pfhe_dcrt32_poly_length(h_); }
    size_t moduli_count() const { return pfhe_dcrt32_moduli_count(h_); }
    size_t crt_poly_length() const { return pfhe_dcrt32_crt_poly_length(h_); }
    uint32_t modulus(size_t i) const { return pfhe_dcrt32_modulus(h_, i); }

    void transform_slice(uint32_t *poly, size_t len) const { check(pfhe_dcrt32_transform_slice(h_, poly, len)); }
    void inverse_transform_slice(uint32_t *poly, size_t len) const { check(pfhe_dcrt32_inverse_transform_slice(h_, poly, len)); }
    void lazy_transform_slice(uint32_t *poly, size_t len) const { check(pfhe_dcrt32_lazy_transform_slice(h_, poly, len)); }
    void lazy_inverse_transform_slice(uint32_t *poly, size_t len) const { check(pfhe_dcrt32_lazy_inverse_transform_slice(h_, poly, len)); }
    void transform_monomial(uint32_t coeff, size_t degree, uint32_t *values, size_t len) const {
        check(pfhe_dcrt32_transform_monomial(h_, coeff, degree, values, len));
    }
    void transform_dev(uint32_t *poly_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_dcrt32_transform_dev(h_, poly_dev, len, lazy, stream));
    }
    void inverse_transform_dev(uint32_t *poly_dev, size_t len, bool lazy = false, void *stream = nullptr) const {
        check(pfhe_dcrt32_inverse_transform_dev(h_, poly_dev, len, lazy, stream));
    }
    void mul_assign_dev(uint32_t *a, size_t len_a, const uint32_t *b, size_t len_b, void *stream = nullptr) const {
        check(pfhe_dcrt32_mul_assign_dev(h_, a, len_a, b, len_b, stream));
    }
    void add_mul_assign_dev(uint32_t *acc, const uint32_t *a, size_t len_a, const uint32_t *b, size_t len_b,
                            void *stream = nullptr) const {
        check(pfhe_dcrt32_add_mul_assign_dev(h_, acc, a, len_a, b, len_b, stream));
    }
    const pfhe_dcrt32 *handle() const { return h_; }

  private:
    pfhe_dcrt32 *h_ = nullptr;
};

class RNSBase {
  public:
    explicit RNSBase(const std::vector<uint64_t> &moduli, int device = 0) {
        check(pfhe_rns_create(moduli.data(), moduli.size(), device, &h_));
    }
    ~RNSBase() { pfhe_rns_destroy(h_); }
    RNSBase(const RNSBase &) = delete;
    RNSBase &operator=(const RNSBase &) = delete;
    size_t moduli_count() const { return pfhe_rns_moduli_count(h_); }
    size_t big_uint_value_len() const { return pfhe_rns_big_uint_value_len(h_); }
    std::vector<uint64_t> moduli_product() const {
        std::vector<uint64_t> q(big_uint_value_len());
        check(pfhe_rns_moduli_product(h_, q.data(), q.size()));
        return q;
    }
    void compose_multiple_values_to(const uint64_t *multi_residues, size_t len_in, uint64_t *big_uint_values,
                                    size_t len_out, size_t value_count) const {
        check(pfhe_rns_compose_multiple_values_to(h_, multi_residues, len_in, big_uint_values, len_out, value_count));
    }
    void wrapping_decompose_small_values_to(const uint64_t *small_values, size_t value_count, uint64_t *multi_residues,
                                            size_t len_out, uint64_t small_value_modulus) const {
        check(pfhe_rns_wrapping_decompose_small_values_to(h_, small_values, value_count, multi_residues, len_out,
                                                          small_value_modulus));
    }
    // RNSBase::add_wrapping_decompose_small_values_scaled / add_decompose_small_values_scaled
    // (crates/primus_rns/src/base.rs:326-416); factors = (value, quotient) per modulus
    void add_wrapping_decompose_small_values_scaled(const uint64_t *small_values, size_t value_count, uint64_t *acc,
                                                    size_t len_acc, uint64_t small_value_modulus,
                                                    const std::vector<uint64_t> &factors) const {
        if (factors.size() != 2 * moduli_count()) throw Error(PFHE_ERR_BAD_LENGTH, "expected one factor per modulus");
        check(pfhe_rns_add_wrapping_decompose_small_values_scaled(h_, small_values, value_count, acc, len_acc,
                                                                  small_value_modulus, factors.data()));
    }
    void add_decompose_small_values_scaled(const uint64_t *small_values, size_t value_count, uint64_t *acc, size_t len_acc,
                                           const std::vector<uint64_t> &factors) const {
        if (factors.size() != 2 * moduli_count()) throw Error(PFHE_ERR_BAD_LENGTH, "expected one factor per modulus");
        check(pfhe_rns_add_decompose_small_values_scaled(h_, small_values, value_count, acc, len_acc, factors.data()));
    }
    // RNSBase::decompose_big_uint_values_to (crates/primus_rns/src/base.rs:457-481)
    void decompose_big_uint_values_to(const uint64_t *big_uint_values, size_t len_in, uint64_t *multi_residues,
                                      size_t len_out, size_t value_count) const {
        check(pfhe_rns_decompose_big_uint_values_to(h_, big_uint_values, len_in, multi_residues, len_out, value_count));
    }
    const pfhe_rns *handle() const { return h_; }

  private:
    pfhe_rns *h_ = nullptr;
};

// primus_rns::BaseConverter (crates/primus_rns/src/converter.rs:21): modulus-major arrays; the reference's
// `scratch` argument has no counterpart
class BaseConverter {
  public:
    BaseConverter(const RNSBase &input_base, const RNSBase &output_base) {
        check(pfhe_conv_create(input_base.handle(), output_base.handle(), &h_));
    }
    ~BaseConverter() { pfhe_conv_destroy(h_); }
    BaseConverter(const BaseConverter &) = delete;
    BaseConverter &operator=(const BaseConverter &) = delete;
    size_t input_moduli_count() const { return pfhe_conv_input_moduli_count(h_); }
    size_t output_moduli_count() const { return pfhe_conv_output_moduli_count(h_); }
    void fast_convert_array(const uint64_t *crt_poly_in, size_t len_in, uint64_t *crt_poly_out, size_t len_out,
                            size_t poly_length) const {
        check(pfhe_conv_fast_convert_array(h_, crt_poly_in, len_in, crt_poly_out, len_out, poly_length));
    }
    void exact_convert_array(const uint64_t *crt_poly_in, size_t len_in, uint64_t *crt_poly_out, size_t len_out,
                             size_t poly_length) const {
        check(pfhe_conv_exact_convert_array(h_, crt_poly_in, len_in, crt_poly_out, len_out, poly_length));
    }
    void fast_convert_array_dev(const uint64_t *in_dev, size_t len_in, uint64_t *out_dev, size_t len_out,
                                size_t poly_length, void *stream = nullptr) const {
        check(pfhe_conv_fast_convert_array_dev(h_, in_dev, len_in, out_dev, len_out, poly_length, stream));
    }
    void exact_convert_array_dev(const uint64_t *in_dev, size_t len_in, uint64_t *out_dev, size_t len_out,
                                 size_t poly_length, void *stream = nullptr) const {
        check(pfhe_conv_exact_convert_array_dev(h_, in_dev, len_in, out_dev, len_out, poly_length, stream));
    }

  private:
    pfhe_conv *h_ = nullptr;
};

class BigUintApproxSignedBasis {
  public:
    BigUintApproxSignedBasis(const RNSBase &base, uint32_t log_basis, size_t reverse_length = 0) {
        check(pfhe_basis_create(base.handle(), log_basis, reverse_length, &h_));
    }
    ~BigUintApproxSignedBasis() { pfhe_basis_destroy(h_); }
    BigUintApproxSignedBasis(const BigUintApproxSignedBasis &) = delete;
    BigUintApproxSignedBasis &operator=(const BigUintApproxSignedBasis &) = delete;
    size_t decompose_length() const { return pfhe_basis_decompose_length(h_); }
    uint32_t log_basis() const { return pfhe_basis_log_basis(h_); }
    uint32_t drop_bits() const { return pfhe_basis_drop_bits(h_); }
    uint64_t basis_value() const { return pfhe_basis_basis_value(h_); }
    void init_value_carry_slice_inplace(uint64_t *values, size_t len, uint8_t *carries, size_t count) const {
        check(pfhe_basis_init_value_carry_slice_inplace(h_, values, len, carries, count));
    }
    void unsigned_decompose_slice_to(size_t level, const uint64_t *values, size_t len, uint64_t *digits,
                                     uint8_t *carries, size_t count) const {
        check(pfhe_basis_unsigned_decompose_slice_to(h_, level, values, len, digits, carries, count));
    }
    // decomposer_iter().nth(level).decompose_slice_to (big_integer/common.rs:289-306): signed digit as a residue mod Q
    void decompose_slice_to(size_t level, const uint64_t *values, size_t len, uint64_t *decomposed_values, size_t len_out,
                            uint8_t *carries, size_t count) const {
        check(pfhe_basis_decompose_slice_to(h_, level, values, len, decomposed_values, len_out, carries, count));
    }
    const pfhe_basis *handle() const { return h_; }

  private:
    pfhe_basis *h_ = nullptr;
};

// DcrtGlevContext (crates/primus_lattice/src/context/glev.rs:4-68) + the handles the reference
// passes next to it.  One holder at a time, like the `&mut` it mirrors: a call from a second thread while one is inside throws
// (PFHE_ERR_BUSY, "plan in use"); successive calls on different streams are ordered by the library.
class DcrtGlevContext {
  public:
    DcrtGlevContext(const U64DcrtTable &table, const RNSBase &base, const BigUintApproxSignedBasis &basis,
                    size_t glwe_dimension = 1, size_t chunk = 0) {
        check(pfhe_extprod_plan_create(table.handle(), base.handle(), basis.handle(), glwe_dimension, chunk, &h_));
    }
    ~DcrtGlevContext() { pfhe_extprod_plan_destroy(h_); }
    DcrtGlevContext(const DcrtGlevContext &) = delete;
    DcrtGlevContext &operator=(const DcrtGlevContext &) = delete;
    pfhe_extprod_plan *handle() const { return h_; }
    bool in_use() const { return pfhe_extprod_plan_in_use(h_) != 0; }  // some thread is inside a call on this context

  private:
    pfhe_extprod_plan *h_ = nullptr;
};

// CrtGlwe::mul_dcrt_ggsw_to (crates/primus_lattice/src/glwe/crt.rs:200-227), host slices
inline void mul_dcrt_ggsw_to(const uint64_t *crt_glwe, size_t len_glwe, const uint64_t *dcrt_ggsw, size_t len_ggsw,
                             uint64_t *result, size_t len_result, DcrtGlevContext &context,
                             bool into_coeff_form = false) {
    check(pfhe_extprod_mul_dcrt_ggsw_to(context.handle(), crt_glwe, len_glwe, dcrt_ggsw, len_ggsw, result, len_result,
                                        into_coeff_form));
}

// device-resident batches
inline void mul_dcrt_ggsw_to_dev(const uint64_t *crt_glwe_dev, size_t len_glwe, const uint64_t *dcrt_ggsw_dev,
                                 size_t len_ggsw, uint64_t *result_dev, size_t len_result, DcrtGlevContext &context,
                                 bool into_coeff_form = false, void *stream = nullptr) {
    check(pfhe_extprod_mul_dcrt_ggsw_to_dev(context.handle(), crt_glwe_dev, len_glwe, dcrt_ggsw_dev, len_ggsw,
                                            result_dev, len_result, into_coeff_form, stream));
}

// ---- the <u32> instantiation of the same operators (RNSBase<u32>, BigUintApproxSignedBasis<u32> — the type the
// reference's tests/big_uint.rs:13 runs — and CrtGlwe<u32>::mul_dcrt_ggsw_to over a U32DcrtTable).  Words are
// uint32_t in memory (residues, digits, limbs of big integers); moduli below 2^30, log_basis below 32; up to 32 moduli.
class RNSBase32 {
  public:
    explicit RNSBase32(const std::vector<uint32_t> &moduli, int device = 0) {
        check(pfhe_rns32_create(moduli.data(), moduli.size(), device, &h_));
    }
    ~RNSBase32() { pfhe_rns32_destroy(h_); }
    RNSBase32(const RNSBase32 &) = delete;
    RNSBase32 &operator=(const RNSBase32 &) = delete;
    size_t moduli_count() const { return pfhe_rns32_moduli_count(h_); }
    size_t big_uint_value_len() const { return pfhe_rns32_big_uint_value_len(h_); }
    std::vector<uint32_t> moduli_product() const {
        std::vector<uint32_t> q(big_uint_value_len());
        check(pfhe_rns32_moduli_product(h_, q.data(), q.size()));
        return q;
    }
    void compose_multiple_values_to(const uint32_t *multi_residues, size_t len_in, uint32_t *big_uint_values,
                                    size_t len_out, size_t value_count) const {
        check(pfhe_rns32_compose_multiple_values_to(h_, multi_residues, len_in, big_uint_values, len_out, value_count));
    }
    void wrapping_decompose_small_values_to(const uint32_t *small_values, size_t value_count, uint32_t *multi_residues,
                                            size_t len_out, uint32_t small_value_modulus) const {
        check(pfhe_rns32_wrapping_decompose_small_values_to(h_, small_values, value_count, multi_residues, len_out,
                                                            small_value_modulus));
    }
    void add_wrapping_decompose_small_values_scaled(const uint32_t *small_values, size_t value_count, uint32_t *acc,
                                                    size_t len_acc, uint32_t small_value_modulus,
                                                    const std::vector<uint32_t> &factors) const {
        if (factors.size() != 2 * moduli_count()) throw Error(PFHE_ERR_BAD_LENGTH, "expected one factor per modulus");
        check(pfhe_rns32_add_wrapping_decompose_small_values_scaled(h_, small_values, value_count, acc, len_acc,
                                                                    small_value_modulus, factors.data()));
    }
    void add_decompose_small_values_scaled(const uint32_t *small_values, size_t value_count, uint32_t *acc, size_t len_acc,
                                           const std::vector<uint32_t> &factors) const {
        if (factors.size() != 2 * moduli_count()) throw Error(PFHE_ERR_BAD_LENGTH, "expected one factor per modulus");
        check(pfhe_rns32_add_decompose_small_values_scaled(h_, small_values, value_count, acc, len_acc, factors.data()));
    }
    void decompose_big_uint_values_to(const uint32_t *big_uint_values, size_t len_in, uint32_t *multi_residues,
                                      size_t len_out, size_t value_count) const {
        check(pfhe_rns32_decompose_big_uint_values_to(h_, big_uint_values, len_in, multi_residues, len_out, value_count));
    }
    const pfhe_rns32 *handle() const { return h_; }

  private:
    pfhe_rns32 *h_ = nullptr;
};

// primus_rns::BaseConverter<u32, BarrettModulus<u32>> (converter.rs:21, generic over T: FheUint)
class BaseConverter32 {
  public:
    BaseConverter32(const RNSBase32 &input_base, const RNSBase32 &output_base) {
        check(pfhe_conv32_create(input_base.handle(), output_base.handle(), &h_));
    }
    ~BaseConverter32() { pfhe_conv32_destroy(h_); }
    BaseConverter32(const BaseConverter32 &) = delete;
    BaseConverter32 &operator=(const BaseConverter32 &) = delete;
    size_t input_moduli_count() const { return pfhe_conv32_input_moduli_count(h_); }
    size_t output_moduli_count() const { return pfhe_conv32_output_moduli_count(h_); }
    void fast_convert_array(const uint32_t *crt_poly_in, size_t len_in, uint32_t *crt_poly_out, size_t len_out,
                            size_t poly_length) const {
        check(pfhe_conv32_fast_convert_array(h_, crt_poly_in, len_in, crt_poly_out, len_out, poly_length));
    }
    void exact_convert_array(const uint32_t *crt_poly_in, size_t len_in, uint32_t *crt_poly_out, size_t len_out,
                             size_t poly_length) const {
        check(pfhe_conv32_exact_convert_array(h_, crt_poly_in, len_in, crt_poly_out, len_out, poly_length));
    }
    void fast_convert_array_dev(const uint32_t *in_dev, size_t len_in, uint32_t *out_dev, size_t len_out,
                                size_t poly_length, void *stream = nullptr) const {
        check(pfhe_conv32_fast_convert_array_dev(h_, in_dev, len_in, out_dev, len_out, poly_length, stream));
    }
    void exact_convert_array_dev(const uint32_t *in_dev, size_t len_in, uint32_t *out_dev, size_t len_out,
                                 size_t poly_length, void *stream = nullptr) const {
        check(pfhe_conv32_exact_convert_array_dev(h_, in_dev, len_in, out_dev, len_out, poly_length, stream));
    }

  private:
    pfhe_conv32 *h_ = nullptr;
};

class BigUintApproxSignedBasis32 {
  public:
    BigUintApproxSignedBasis32(const RNSBase32 &base, uint32_t log_basis, size_t reverse_length = 0) {
        check(pfhe_basis32_create(base.handle(), log_basis, reverse_length, &h_));
    }
    ~BigUintApproxSignedBasis32() { pfhe_basis32_destroy(h_); }
    BigUintApproxSignedBasis32(const BigUintApproxSignedBasis32 &) = delete;
    BigUintApproxSignedBasis32 &operator=(const BigUintApproxSignedBasis32 &) = delete;
    size_t decompose_length() const { return pfhe_basis32_decompose_length(h_); }
    uint32_t log_basis() const { return pfhe_basis32_log_basis(h_); }
    uint32_t drop_bits() const { return pfhe_basis32_drop_bits(h_); }
    uint32_t basis_value() const { return pfhe_basis32_basis_value(h_); }
    void init_value_carry_slice_inplace(uint32_t *values, size_t len, uint8_t *carries, size_t count) const {
        check(pfhe_basis32_init_value_carry_slice_inplace(h_, values, len, carries, count));
    }
    void unsigned_decompose_slice_to(size_t level, const uint32_t *values, size_t len, uint32_t *digits,
                                     uint8_t *carries, size_t count) const {
        check(pfhe_basis32_unsigned_decompose_slice_to(h_, level, values, len, digits, carries, count));
    }
    void decompose_slice_to(size_t level, const uint32_t *values, size_t len, uint32_t *decomposed_values, size_t len_out,
                            uint8_t *carries, size_t count) const {
        check(pfhe_basis32_decompose_slice_to(h_, level, values, len, decomposed_values, len_out, carries, count));
    }
    const pfhe_basis32 *handle() const { return h_; }

  private:
    pfhe_basis32 *h_ = nullptr;
};

// DcrtGlevContext over a U32DcrtTable; one holder at a time like DcrtGlevContext
class DcrtGlevContext32 {
  public:
    DcrtGlevContext32(const U32DcrtTable &table, const RNSBase32 &base, const BigUintApproxSignedBasis32 &basis,
                      size_t glwe_dimension = 1, size_t chunk = 0) {
        check(pfhe_extprod32_plan_create(table.handle(), base.handle(), basis.handle(), glwe_dimension, chunk, &h_));
    }
    ~DcrtGlevContext32() { pfhe_extprod32_plan_destroy(h_); }
    DcrtGlevContext32(const DcrtGlevContext32 &) = delete;
    DcrtGlevContext32 &operator=(const DcrtGlevContext32 &) = delete;
    pfhe_extprod32_plan *handle() const { return h_; }
    bool in_use() const { return pfhe_extprod32_plan_in_use(h_) != 0; }
    size_t scratch_bytes() const { return pfhe_extprod32_plan_scratch_bytes(h_); }

  private:
    pfhe_extprod32_plan *h_ = nullptr;
};

// CrtGlwe<u32>::mul_dcrt_ggsw_to (crates/primus_lattice/src/glwe/crt.rs:200-227)
inline void mul_dcrt_ggsw_to(const uint32_t *crt_glwe, size_t len_glwe, const uint32_t *dcrt_ggsw, size_t len_ggsw,
                             uint32_t *result, size_t len_result, DcrtGlevContext32 &context,
                             bool into_coeff_form = false) {
    check(pfhe_extprod32_mul_dcrt_ggsw_to(context.handle(), crt_glwe, len_glwe, dcrt_ggsw, len_ggsw, result, len_result,
                                          into_coeff_form));
}

inline void mul_dcrt_ggsw_to_dev(const uint32_t *crt_glwe_dev, size_t len_glwe, const uint32_t *dcrt_ggsw_dev,
                                 size_t len_ggsw, uint32_t *result_dev, size_t len_result, DcrtGlevContext32 &context,
                                 bool into_coeff_form = false, void *stream = nullptr) {
    check(pfhe_extprod32_mul_dcrt_ggsw_to_dev(context.handle(), crt_glwe_dev, len_glwe, dcrt_ggsw_dev, len_ggsw,
                                              result_dev, len_result, into_coeff_form, stream));
}

}  // namespace pfhe
