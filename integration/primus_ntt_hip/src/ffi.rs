//! Raw declarations of the C ABI (include/pfhe.h).  Only what the trait implementations and the
//! batched external-product wrapper need; the rest of pfhe.h binds the same way.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0] } )* } }
opaque!(pfhe_ntt, pfhe_dcrt, pfhe_ntt32, pfhe_dcrt32, pfhe_rns, pfhe_basis, pfhe_extprod_plan, pfhe_conv);

pub const PFHE_OK: c_int = 0;
pub const PFHE_ERR_NO_INVERSE: c_int = 37;

unsafe extern "C" {
    pub fn pfhe_last_error() -> *const c_char;
    pub fn pfhe_device_malloc(device: c_int, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn pfhe_device_free(device: c_int, ptr: *mut c_void) -> c_int;
    pub fn pfhe_memcpy_h2d(device: c_int, dst: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_memcpy_d2h(device: c_int, dst: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    // host-pointer entry points stage through a per-device pool (no allocation per call): diagnostics
    pub fn pfhe_debug_alloc_count() -> u64;
    pub fn pfhe_staging_release(device: c_int) -> c_int;

    // U64NttTable
    pub fn pfhe_ntt_create(log_n: u32, modulus: u64, device: c_int, out: *mut *mut pfhe_ntt) -> c_int;
    pub fn pfhe_ntt_destroy(t: *mut pfhe_ntt);
    pub fn pfhe_ntt_poly_length(t: *const pfhe_ntt) -> usize;
    pub fn pfhe_ntt_modulus(t: *const pfhe_ntt) -> u64;
    pub fn pfhe_ntt_log_n(t: *const pfhe_ntt) -> u32;
    pub fn pfhe_ntt_root(t: *const pfhe_ntt) -> u64;
    pub fn pfhe_ntt_inv_root(t: *const pfhe_ntt) -> u64;
    pub fn pfhe_ntt_inv_n(t: *const pfhe_ntt) -> u64;
    pub fn pfhe_ntt_transform_slice(t: *const pfhe_ntt, poly: *mut u64, len: usize) -> c_int;
    pub fn pfhe_ntt_inverse_transform_slice(t: *const pfhe_ntt, values: *mut u64, len: usize) -> c_int;
    pub fn pfhe_ntt_lazy_transform_slice(t: *const pfhe_ntt, poly: *mut u64, len: usize) -> c_int;
    pub fn pfhe_ntt_lazy_inverse_transform_slice(t: *const pfhe_ntt, values: *mut u64, len: usize) -> c_int;
    pub fn pfhe_ntt_transform_monomial(t: *const pfhe_ntt, coeff: u64, degree: usize, values: *mut u64, len: usize) -> c_int;
    pub fn pfhe_ntt_transform_coeff_one_monomial(t: *const pfhe_ntt, degree: usize, values: *mut u64, len: usize) -> c_int;
    pub fn pfhe_ntt_transform_coeff_minus_one_monomial(t: *const pfhe_ntt, degree: usize, values: *mut u64, len: usize) -> c_int;

    // U64DcrtTable
    pub fn pfhe_dcrt_create(log_n: u32, moduli: *const u64, count: usize, device: c_int, out: *mut *mut pfhe_dcrt) -> c_int;
    pub fn pfhe_dcrt_destroy(t: *mut pfhe_dcrt);
    pub fn pfhe_dcrt_transform_slice(t: *const pfhe_dcrt, poly: *mut u64, len: usize) -> c_int;
    pub fn pfhe_dcrt_inverse_transform_slice(t: *const pfhe_dcrt, poly: *mut u64, len: usize) -> c_int;
    pub fn pfhe_dcrt_lazy_transform_slice(t: *const pfhe_dcrt, poly: *mut u64, len: usize) -> c_int;
    pub fn pfhe_dcrt_lazy_inverse_transform_slice(t: *const pfhe_dcrt, poly: *mut u64, len: usize) -> c_int;
    pub fn pfhe_dcrt_transform_dev(t: *const pfhe_dcrt, poly_dev: *mut u64, len: usize, lazy: c_int, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_inverse_transform_dev(t: *const pfhe_dcrt, poly_dev: *mut u64, len: usize, lazy: c_int, stream: *mut c_void) -> c_int;

    // element-wise family on device-resident CRT polynomials / GLWE ciphertexts (len = total words)
    pub fn pfhe_dcrt_add_to_dev(t: *const pfhe_dcrt, a: *const u64, b: *const u64, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_sub_to_dev(t: *const pfhe_dcrt, a: *const u64, b: *const u64, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_neg_to_dev(t: *const pfhe_dcrt, a: *const u64, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_mul_scalar_to_dev(t: *const pfhe_dcrt, a: *const u64, scalars: *const u64, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_add_mul_scalar_assign_dev(t: *const pfhe_dcrt, acc: *mut u64, rhs: *const u64, scalars: *const u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_mul_factor_to_dev(t: *const pfhe_dcrt, a: *const u64, factors: *const u64, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_add_mul_factor_assign_dev(t: *const pfhe_dcrt, acc: *mut u64, rhs: *const u64, factors: *const u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_mul_monomial_to_dev(t: *const pfhe_dcrt, a: *const u64, r: usize, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_mul_monomial_assign_dev(t: *const pfhe_dcrt, data: *mut u64, r: usize, len: usize, stream: *mut c_void) -> c_int;
    pub fn pfhe_dcrt_inv_to_dev(t: *const pfhe_dcrt, a: *const u64, out: *mut u64, len: usize, stream: *mut c_void) -> c_int;

    // U32NttTable (same shape with u32 words)
    pub fn pfhe_ntt32_create(log_n: u32, modulus: u32, device: c_int, out: *mut *mut pfhe_ntt32) -> c_int;
    pub fn pfhe_ntt32_destroy(t: *mut pfhe_ntt32);
    pub fn pfhe_ntt32_poly_length(t: *const pfhe_ntt32) -> usize;
    pub fn pfhe_ntt32_transform_slice(t: *const pfhe_ntt32, poly: *mut u32, len: usize) -> c_int;
    pub fn pfhe_ntt32_inverse_transform_slice(t: *const pfhe_ntt32, values: *mut u32, len: usize) -> c_int;
    pub fn pfhe_ntt32_lazy_transform_slice(t: *const pfhe_ntt32, poly: *mut u32, len: usize) -> c_int;
    pub fn pfhe_ntt32_lazy_inverse_transform_slice(t: *const pfhe_ntt32, values: *mut u32, len: usize) -> c_int;
    pub fn pfhe_ntt32_transform_monomial(t: *const pfhe_ntt32, coeff: u32, degree: usize, values: *mut u32, len: usize) -> c_int;
    pub fn pfhe_ntt32_transform_coeff_one_monomial(t: *const pfhe_ntt32, degree: usize, values: *mut u32, len: usize) -> c_int;
    pub fn pfhe_ntt32_transform_coeff_minus_one_monomial(t: *const pfhe_ntt32, degree: usize, values: *mut u32, len: usize) -> c_int;

    // RNS gadget external product (batched, device resident)
    pub fn pfhe_rns_create(moduli: *const u64, count: usize, device: c_int, out: *mut *mut pfhe_rns) -> c_int;
    pub fn pfhe_rns_destroy(r: *mut pfhe_rns);
    pub fn pfhe_basis_create(base: *const pfhe_rns, log_basis: u32, reverse_length: usize, out: *mut *mut pfhe_basis) -> c_int;
    pub fn pfhe_basis_destroy(b: *mut pfhe_basis);
    pub fn pfhe_extprod_plan_create(table: *const pfhe_dcrt, base: *const pfhe_rns, basis: *const pfhe_basis,
                                    glwe_dimension: usize, chunk: usize, out: *mut *mut pfhe_extprod_plan) -> c_int;
    pub fn pfhe_extprod_plan_destroy(p: *mut pfhe_extprod_plan);
    /// 1 while some thread is inside a pfhe_extprod_* call on the plan (a second thread gets PFHE_ERR_BAD_ARGUMENT)
    pub fn pfhe_extprod_plan_in_use(p: *const pfhe_extprod_plan) -> c_int;
    pub fn pfhe_extprod_mul_dcrt_ggsw_to_dev(plan: *mut pfhe_extprod_plan, crt_glwe_dev: *const u64, len_glwe: usize,
                                             dcrt_ggsw_dev: *const u64, len_ggsw: usize, result_dev: *mut u64,
                                             len_result: usize, into_coeff_form: c_int, stream: *mut c_void) -> c_int;
}
