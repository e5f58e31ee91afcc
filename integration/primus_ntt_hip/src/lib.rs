//! `NttTable` / `DcrtTable` implementations that delegate to hand-written HIP kernels for MI355X
//! through the C ABI of libpfhe_hip.so (include/pfhe.h).  UNTESTED SOURCE — see Cargo.toml.
//!
//! Replaces, behind the same traits:
//!   * `U64NttTable`  (crates/primus_ntt/src/ntt/prime64/table.rs:41)  -> [`HipNttTable`]
//!   * `U64DcrtTable` (crates/primus_ntt/src/dcrt/prime64.rs:11)       -> [`HipDcrtTable`]
//!   * `U32NttTable`  (crates/primus_ntt/src/ntt/prime32/table.rs:37)  -> [`HipU32NttTable`]
//! so that every function of primus_lattice that is generic over `Table: NttTable` / `DcrtTable`
//! runs on the GPU unchanged (one host->device->host round trip per `&mut [T]` call), and adds the
//! batched, device-resident external product as [`HipExternalProduct`].
mod ffi;

use core::ffi::{c_int, CStr};

use primus_data::{DataMut, RawData};
use primus_ntt::{DcrtTable, NttError, NttTable};
use primus_poly::{CrtPolynomial, DcrtPolynomial, NttPolynomial, Polynomial};
use primus_reduce::FieldContext;

fn device() -> c_int {
    std::env::var("PFHE_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0)
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::pfhe_last_error()) }.to_string_lossy().into_owned()
}

/// pfhe_status -> NttError (crates/primus_ntt/src/error.rs:7-49; include/pfhe.h status codes 1..5).
fn status_to_err<T: From<u32> + Copy>(code: c_int, n: usize, q: T, max_bits: u32) -> NttError<T> {
    match code {
        1 => NttError::NoPrimitiveRoot { degree: T::from(2 * n as u32), modulus: q },
        2 => NttError::DegreeConversionErr { degree: n, modulus: q },
        3 => NttError::DegreeTooLarge { degree: n, modulus: q },
        5 => NttError::ModulusTooLarge { modulus: q, max_bits },
        _ => NttError::NttTableErr, // incl. PFHE_ERR_NO_DEVICE / PFHE_ERR_HIP: no CPU fallback
    }
}

/// The reference's transforms are infallible (`table.rs:541-563` only debug_asserts lengths); a
/// non-zero status here is a length mismatch or a device failure, i.e. a state the reference would
/// panic on as well.
fn expect_ok(rc: c_int, what: &str) {
    assert_eq!(rc, ffi::PFHE_OK, "{what}: {}", last_error());
}

// ------------------------------------------------------------------------------------------------
// U64NttTable
// ------------------------------------------------------------------------------------------------
pub struct HipNttTable {
    h: *mut ffi::pfhe_ntt,
}
// handles are immutable after creation and usable from several host threads (pfhe.h "Conventions")
unsafe impl Send for HipNttTable {}
unsafe impl Sync for HipNttTable {}
impl Drop for HipNttTable {
    fn drop(&mut self) {
        unsafe { ffi::pfhe_ntt_destroy(self.h) }
    }
}

/// The inherent getters of `U64NttTable` (crates/primus_ntt/src/ntt/prime64/table.rs:127-161).
impl HipNttTable {
    pub fn modulus(&self) -> u64 {
        unsafe { ffi::pfhe_ntt_modulus(self.h) }
    }
    pub fn log_n(&self) -> u32 {
        unsafe { ffi::pfhe_ntt_log_n(self.h) }
    }
    pub fn n(&self) -> usize {
        unsafe { ffi::pfhe_ntt_poly_length(self.h) }
    }
    pub fn root(&self) -> u64 {
        unsafe { ffi::pfhe_ntt_root(self.h) }
    }
    pub fn inv_root(&self) -> u64 {
        unsafe { ffi::pfhe_ntt_inv_root(self.h) }
    }
    pub fn inv_n(&self) -> u64 {
        unsafe { ffi::pfhe_ntt_inv_n(self.h) }
    }
}

impl NttTable for HipNttTable {
    type ValueT = u64;

    fn new<M: FieldContext<u64>>(log_n: u32, modulus: M) -> Result<Self, NttError<u64>> {
        let q = modulus.value().ok_or(NttError::NttTableErr)?; // table.rs:313-315
        let mut h = core::ptr::null_mut();
        match unsafe { ffi::pfhe_ntt_create(log_n, q, device(), &mut h) } {
            ffi::PFHE_OK => Ok(Self { h }),
            e => Err(status_to_err(e, 1usize << log_n, q, 62)),
        }
    }
    fn poly_length(&self) -> usize {
        unsafe { ffi::pfhe_ntt_poly_length(self.h) }
    }
    fn transform_inplace<S: RawData<Elem = u64> + DataMut>(&self, mut poly: Polynomial<S>) -> NttPolynomial<S> {
        self.transform_slice(poly.as_mut_slice()); // table.rs:523-530
        NttPolynomial::new(poly.0)
    }
    fn inverse_transform_inplace<S: RawData<Elem = u64> + DataMut>(&self, mut values: NttPolynomial<S>) -> Polynomial<S> {
        self.inverse_transform_slice(values.as_mut_slice()); // table.rs:532-539
        Polynomial::new(values.0)
    }
    fn lazy_transform_slice(&self, poly: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_lazy_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "lazy_transform_slice")
    }
    fn transform_slice(&self, poly: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "transform_slice")
    }
    fn lazy_inverse_transform_slice(&self, values: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_lazy_inverse_transform_slice(self.h, values.as_mut_ptr(), values.len()) },
                  "lazy_inverse_transform_slice")
    }
    fn inverse_transform_slice(&self, values: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_inverse_transform_slice(self.h, values.as_mut_ptr(), values.len()) },
                  "inverse_transform_slice")
    }
    fn transform_monomial(&self, coeff: u64, degree: usize, values: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_transform_monomial(self.h, coeff, degree, values.as_mut_ptr(), values.len()) },
                  "transform_monomial")
    }
    fn transform_coeff_one_monomial(&self, degree: usize, values: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_transform_coeff_one_monomial(self.h, degree, values.as_mut_ptr(), values.len()) },
                  "transform_coeff_one_monomial")
    }
    fn transform_coeff_minus_one_monomial(&self, degree: usize, values: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_ntt_transform_coeff_minus_one_monomial(self.h, degree, values.as_mut_ptr(), values.len()) },
                  "transform_coeff_minus_one_monomial")
    }
}

// ------------------------------------------------------------------------------------------------
// U64DcrtTable
// ------------------------------------------------------------------------------------------------
pub struct HipDcrtTable {
    h: *mut ffi::pfhe_dcrt,
    limbs: Vec<HipNttTable>, // DcrtTable::ntt_tables() hands out per-limb tables (dcrt/mod.rs:31-35)
    poly_length: usize,
}
unsafe impl Send for HipDcrtTable {}
unsafe impl Sync for HipDcrtTable {}
impl Drop for HipDcrtTable {
    fn drop(&mut self) {
        unsafe { ffi::pfhe_dcrt_destroy(self.h) }
    }
}
impl HipDcrtTable {
    /// Raw handle for the device-resident entry points (`pfhe_dcrt_*_dev`, `pfhe_extprod_*`).
    pub fn handle(&self) -> *const ffi::pfhe_dcrt {
        self.h
    }

    // Device-resident element-wise forms of CrtGlwe / CrtPolynomial methods for a batch in one slice
    // (crates/primus_lattice/src/macros/mod.rs:367-531, glwe/crt.rs:59-175, primus_poly/src/crt/mul.rs:102-127).
    // All pointers are device pointers to `len` words; `out` may alias `a`.

    /// `CrtGlwe::add_element_wise_to` / `add_element_wise_assign`.
    pub unsafe fn add_element_wise_to_dev(&self, a: *const u64, b: *const u64, out: *mut u64, len: usize,
                                          stream: *mut core::ffi::c_void) -> Result<(), c_int> {
        status(unsafe { ffi::pfhe_dcrt_add_to_dev(self.h, a, b, out, len, stream) })
    }
    /// `CrtGlwe::sub_element_wise_to` / `sub_element_wise_assign`.
    pub unsafe fn sub_element_wise_to_dev(&self, a: *const u64, b: *const u64, out: *mut u64, len: usize,
                                          stream: *mut core::ffi::c_void) -> Result<(), c_int> {
        status(unsafe { ffi::pfhe_dcrt_sub_to_dev(self.h, a, b, out, len, stream) })
    }
    /// `CrtGlwe::mul_scalar_to` / `mul_scalar_assign` (`scalar_residue`: one residue per modulus, on the host).
    pub unsafe fn mul_scalar_to_dev(&self, a: *const u64, scalar_residue: &[u64], out: *mut u64, len: usize,
                                    stream: *mut core::ffi::c_void) -> Result<(), c_int> {
        assert_eq!(scalar_residue.len(), self.limbs.len());
        status(unsafe { ffi::pfhe_dcrt_mul_scalar_to_dev(self.h, a, scalar_residue.as_ptr(), out, len, stream) })
    }
    /// `CrtGlwe::mul_monic_monomial_assign(r)` written to a second buffer (`self * X^r`, `r < 2N`).
    pub unsafe fn mul_monic_monomial_to_dev(&self, a: *const u64, r: usize, out: *mut u64, len: usize,
                                            stream: *mut core::ffi::c_void) -> Result<(), c_int> {
        status(unsafe { ffi::pfhe_dcrt_mul_monomial_to_dev(self.h, a, r, out, len, stream) })
    }
    /// `DcrtPolynomial::inv_to`; `Err(PFHE_ERR_NO_INVERSE)` where the reference panics.
    pub unsafe fn inv_to_dev(&self, a: *const u64, out: *mut u64, len: usize, stream: *mut core::ffi::c_void)
                             -> Result<(), c_int> {
        status(unsafe { ffi::pfhe_dcrt_inv_to_dev(self.h, a, out, len, stream) })
    }
}

fn status(rc: c_int) -> Result<(), c_int> {
    if rc == ffi::PFHE_OK { Ok(()) } else { Err(rc) }
}

impl DcrtTable for HipDcrtTable {
    type ValueT = u64;
    type NttTables = HipNttTable;

    fn new<M: FieldContext<u64>>(log_n: u32, moduli: &[M]) -> Result<Self, NttError<u64>> {
        let qs: Vec<u64> = moduli.iter().map(|m| m.value().ok_or(NttError::NttTableErr)).collect::<Result<_, _>>()?;
        let limbs = moduli.iter().map(|m| HipNttTable::new(log_n, *m)).collect::<Result<Vec<_>, _>>()?;
        let mut h = core::ptr::null_mut();
        match unsafe { ffi::pfhe_dcrt_create(log_n, qs.as_ptr(), qs.len(), device(), &mut h) } {
            ffi::PFHE_OK => Ok(Self { h, limbs, poly_length: 1usize << log_n }),
            e => Err(status_to_err(e, 1usize << log_n, qs.first().copied().unwrap_or(0), 62)),
        }
    }
    fn ntt_tables(&self) -> &[HipNttTable] {
        &self.limbs
    }
    fn iter(&self) -> std::slice::Iter<'_, HipNttTable> {
        self.limbs.iter()
    }
    fn poly_length(&self) -> usize {
        self.poly_length
    }
    fn moduli_count(&self) -> usize {
        self.limbs.len()
    }
    fn crt_poly_length(&self) -> usize {
        self.poly_length * self.limbs.len()
    }
    fn transform_inplace<S: RawData<Elem = u64> + DataMut>(&self, mut crt_poly: CrtPolynomial<S>) -> DcrtPolynomial<S> {
        self.transform_slice(crt_poly.as_mut_slice()); // one launch for all limbs (dcrt/prime64.rs:71-83 loops)
        DcrtPolynomial::new(crt_poly.0)
    }
    fn inverse_transform_inplace<S: RawData<Elem = u64> + DataMut>(&self, mut dcrt_poly: DcrtPolynomial<S>) -> CrtPolynomial<S> {
        self.inverse_transform_slice(dcrt_poly.as_mut_slice());
        CrtPolynomial::new(dcrt_poly.0)
    }
    fn lazy_transform_slice(&self, poly: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_dcrt_lazy_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "dcrt lazy_transform_slice")
    }
    fn transform_slice(&self, poly: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_dcrt_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "dcrt transform_slice")
    }
    fn lazy_inverse_transform_slice(&self, poly: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_dcrt_lazy_inverse_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) },
                  "dcrt lazy_inverse_transform_slice")
    }
    fn inverse_transform_slice(&self, poly: &mut [u64]) {
        expect_ok(unsafe { ffi::pfhe_dcrt_inverse_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) },
                  "dcrt inverse_transform_slice")
    }
    // transform_monomial & co: the trait's provided methods (dcrt/mod.rs:107-134) call the per-limb tables
}

// ------------------------------------------------------------------------------------------------
// U32NttTable
// ------------------------------------------------------------------------------------------------
pub struct HipU32NttTable {
    h: *mut ffi::pfhe_ntt32,
}
unsafe impl Send for HipU32NttTable {}
unsafe impl Sync for HipU32NttTable {}
impl Drop for HipU32NttTable {
    fn drop(&mut self) {
        unsafe { ffi::pfhe_ntt32_destroy(self.h) }
    }
}

impl NttTable for HipU32NttTable {
    type ValueT = u32;

    fn new<M: FieldContext<u32>>(log_n: u32, modulus: M) -> Result<Self, NttError<u32>> {
        let q = modulus.value().ok_or(NttError::NttTableErr)?;
        let mut h = core::ptr::null_mut();
        match unsafe { ffi::pfhe_ntt32_create(log_n, q, device(), &mut h) } {
            ffi::PFHE_OK => Ok(Self { h }),
            e => Err(status_to_err(e, 1usize << log_n, q, 30)), // prime32/table.rs:195-200
        }
    }
    fn poly_length(&self) -> usize {
        unsafe { ffi::pfhe_ntt32_poly_length(self.h) }
    }
    fn transform_inplace<S: RawData<Elem = u32> + DataMut>(&self, mut poly: Polynomial<S>) -> NttPolynomial<S> {
        self.transform_slice(poly.as_mut_slice());
        NttPolynomial::new(poly.0)
    }
    fn inverse_transform_inplace<S: RawData<Elem = u32> + DataMut>(&self, mut values: NttPolynomial<S>) -> Polynomial<S> {
        self.inverse_transform_slice(values.as_mut_slice());
        Polynomial::new(values.0)
    }
    fn lazy_transform_slice(&self, poly: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_lazy_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "u32 lazy_transform_slice")
    }
    fn transform_slice(&self, poly: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "u32 transform_slice")
    }
    fn lazy_inverse_transform_slice(&self, values: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_lazy_inverse_transform_slice(self.h, values.as_mut_ptr(), values.len()) },
                  "u32 lazy_inverse_transform_slice")
    }
    fn inverse_transform_slice(&self, values: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_inverse_transform_slice(self.h, values.as_mut_ptr(), values.len()) },
                  "u32 inverse_transform_slice")
    }
    fn transform_monomial(&self, coeff: u32, degree: usize, values: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_transform_monomial(self.h, coeff, degree, values.as_mut_ptr(), values.len()) },
                  "u32 transform_monomial")
    }
    fn transform_coeff_one_monomial(&self, degree: usize, values: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_transform_coeff_one_monomial(self.h, degree, values.as_mut_ptr(), values.len()) },
                  "u32 transform_coeff_one_monomial")
    }
    fn transform_coeff_minus_one_monomial(&self, degree: usize, values: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_ntt32_transform_coeff_minus_one_monomial(self.h, degree, values.as_mut_ptr(), values.len()) },
                  "u32 transform_coeff_minus_one_monomial")
    }
}

// ------------------------------------------------------------------------------------------------
// U32DcrtTable (crates/primus_ntt/src/dcrt/prime32.rs:11-128)
// ------------------------------------------------------------------------------------------------
pub struct HipU32DcrtTable {
    h: *mut ffi::pfhe_dcrt32,
    limbs: Vec<HipU32NttTable>,
    poly_length: usize,
}
unsafe impl Send for HipU32DcrtTable {}
unsafe impl Sync for HipU32DcrtTable {}
impl Drop for HipU32DcrtTable {
    fn drop(&mut self) {
        unsafe { ffi::pfhe_dcrt32_destroy(self.h) }
    }
}
impl HipU32DcrtTable {
    /// Raw handle for the device-resident entry points (`pfhe_dcrt32_*_dev`, `pfhe_extprod32_*`).
    pub fn handle(&self) -> *const ffi::pfhe_dcrt32 {
        self.h
    }
}

impl DcrtTable for HipU32DcrtTable {
    type ValueT = u32;
    type NttTables = HipU32NttTable;

    fn new<M: FieldContext<u32>>(log_n: u32, moduli: &[M]) -> Result<Self, NttError<u32>> {
        let qs: Vec<u32> = moduli.iter().map(|m| m.value().ok_or(NttError::NttTableErr)).collect::<Result<_, _>>()?;
        let limbs = moduli.iter().map(|m| HipU32NttTable::new(log_n, *m)).collect::<Result<Vec<_>, _>>()?;
        let mut h = core::ptr::null_mut();
        match unsafe { ffi::pfhe_dcrt32_create(log_n, qs.as_ptr(), qs.len(), device(), &mut h) } {
            ffi::PFHE_OK => Ok(Self { h, limbs, poly_length: 1usize << log_n }),
            e => Err(status_to_err(e, 1usize << log_n, qs.first().copied().unwrap_or(0), 30)),
        }
    }
    fn ntt_tables(&self) -> &[HipU32NttTable] {
        &self.limbs
    }
    fn iter(&self) -> std::slice::Iter<'_, HipU32NttTable> {
        self.limbs.iter()
    }
    fn poly_length(&self) -> usize {
        self.poly_length
    }
    fn moduli_count(&self) -> usize {
        self.limbs.len()
    }
    fn crt_poly_length(&self) -> usize {
        self.poly_length * self.limbs.len()
    }
    fn transform_inplace<S: RawData<Elem = u32> + DataMut>(&self, mut crt_poly: CrtPolynomial<S>) -> DcrtPolynomial<S> {
        self.transform_slice(crt_poly.as_mut_slice());
        DcrtPolynomial::new(crt_poly.0)
    }
    fn inverse_transform_inplace<S: RawData<Elem = u32> + DataMut>(&self, mut dcrt_poly: DcrtPolynomial<S>) -> CrtPolynomial<S> {
        self.inverse_transform_slice(dcrt_poly.as_mut_slice());
        CrtPolynomial::new(dcrt_poly.0)
    }
    fn lazy_transform_slice(&self, poly: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_dcrt32_lazy_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "dcrt32 lazy_transform_slice")
    }
    fn transform_slice(&self, poly: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_dcrt32_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) }, "dcrt32 transform_slice")
    }
    fn lazy_inverse_transform_slice(&self, poly: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_dcrt32_lazy_inverse_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) },
                  "dcrt32 lazy_inverse_transform_slice")
    }
    fn inverse_transform_slice(&self, poly: &mut [u32]) {
        expect_ok(unsafe { ffi::pfhe_dcrt32_inverse_transform_slice(self.h, poly.as_mut_ptr(), poly.len()) },
                  "dcrt32 inverse_transform_slice")
    }
}

// ------------------------------------------------------------------------------------------------
// Batched, device-resident RNS gadget external product:
//   CrtGlwe::mul_dcrt_ggsw_to (crates/primus_lattice/src/glwe/crt.rs:200-227) for `batch` ciphertexts
// ------------------------------------------------------------------------------------------------
/// Owns what the reference passes as `&BigUintApproxSignedBasis`, `&RNSBase` and `&mut DcrtGlevContext`
/// (scratch): one per stream, like the `&mut` context it mirrors.
pub struct HipExternalProduct {
    rns: *mut ffi::pfhe_rns,
    basis: *mut ffi::pfhe_basis,
    plan: *mut ffi::pfhe_extprod_plan,
}
impl HipExternalProduct {
    pub fn new(table: &HipDcrtTable, moduli: &[u64], log_basis: u32, glwe_dimension: usize) -> Result<Self, c_int> {
        let (mut rns, mut basis, mut plan) = (core::ptr::null_mut(), core::ptr::null_mut(), core::ptr::null_mut());
        unsafe {
            let rc = ffi::pfhe_rns_create(moduli.as_ptr(), moduli.len(), device(), &mut rns);
            if rc != ffi::PFHE_OK { return Err(rc); }
            let rc = ffi::pfhe_basis_create(rns, log_basis, 0, &mut basis);
            if rc != ffi::PFHE_OK { ffi::pfhe_rns_destroy(rns); return Err(rc); }
            let rc = ffi::pfhe_extprod_plan_create(table.handle(), rns, basis, glwe_dimension, 0, &mut plan);
            if rc != ffi::PFHE_OK { ffi::pfhe_basis_destroy(basis); ffi::pfhe_rns_destroy(rns); return Err(rc); }
        }
        Ok(Self { rns, basis, plan })
    }
    /// `crt_glwe_dev`: batch x (k+1) x L x N words on the device; `dcrt_ggsw_dev`: one GGSW
    /// ((k+1) x ell x (k+1) x L x N words, shared) or one per ciphertext; `result_dev` like `crt_glwe_dev`.
    pub unsafe fn mul_dcrt_ggsw_to_dev(&mut self, crt_glwe_dev: *const u64, len_glwe: usize, dcrt_ggsw_dev: *const u64,
                                       len_ggsw: usize, result_dev: *mut u64, into_coeff_form: bool,
                                       stream: *mut core::ffi::c_void) -> Result<(), c_int> {
        match unsafe { ffi::pfhe_extprod_mul_dcrt_ggsw_to_dev(self.plan, crt_glwe_dev, len_glwe, dcrt_ggsw_dev, len_ggsw,
                                                             result_dev, len_glwe, into_coeff_form as c_int, stream) } {
            ffi::PFHE_OK => Ok(()),
            e => Err(e),
        }
    }
}
impl Drop for HipExternalProduct {
    fn drop(&mut self) {
        unsafe {
            ffi::pfhe_extprod_plan_destroy(self.plan);
            ffi::pfhe_basis_destroy(self.basis);
            ffi::pfhe_rns_destroy(self.rns);
        }
    }
}

/// The same over `U32DcrtTable`: `CrtGlwe::<u32>::mul_dcrt_ggsw_to` with `RNSBase<u32>` and
/// `BigUintApproxSignedBasis<u32>` (u32 words on the device; moduli below 2^30, log_basis below 32).
pub struct HipExternalProduct32 {
    rns: *mut ffi::pfhe_rns32,
    basis: *mut ffi::pfhe_basis32,
    plan: *mut ffi::pfhe_extprod32_plan,
}
impl HipExternalProduct32 {
    pub fn new(table: &HipU32DcrtTable, moduli: &[u32], log_basis: u32, glwe_dimension: usize) -> Result<Self, c_int> {
        let (mut rns, mut basis, mut plan) = (core::ptr::null_mut(), core::ptr::null_mut(), core::ptr::null_mut());
        unsafe {
            let rc = ffi::pfhe_rns32_create(moduli.as_ptr(), moduli.len(), device(), &mut rns);
            if rc != ffi::PFHE_OK { return Err(rc); }
            let rc = ffi::pfhe_basis32_create(rns, log_basis, 0, &mut basis);
            if rc != ffi::PFHE_OK { ffi::pfhe_rns32_destroy(rns); return Err(rc); }
            let rc = ffi::pfhe_extprod32_plan_create(table.handle(), rns, basis, glwe_dimension, 0, &mut plan);
            if rc != ffi::PFHE_OK { ffi::pfhe_basis32_destroy(basis); ffi::pfhe_rns32_destroy(rns); return Err(rc); }
        }
        Ok(Self { rns, basis, plan })
    }
    pub unsafe fn mul_dcrt_ggsw_to_dev(&mut self, crt_glwe_dev: *const u32, len_glwe: usize, dcrt_ggsw_dev: *const u32,
                                       len_ggsw: usize, result_dev: *mut u32, into_coeff_form: bool,
                                       stream: *mut core::ffi::c_void) -> Result<(), c_int> {
        status(unsafe { ffi::pfhe_extprod32_mul_dcrt_ggsw_to_dev(self.plan, crt_glwe_dev, len_glwe, dcrt_ggsw_dev, len_ggsw,
                                                                 result_dev, len_glwe, into_coeff_form as c_int, stream) })
    }
}
impl Drop for HipExternalProduct32 {
    fn drop(&mut self) {
        unsafe {
            ffi::pfhe_extprod32_plan_destroy(self.plan);
            ffi::pfhe_basis32_destroy(self.basis);
            ffi::pfhe_rns32_destroy(self.rns);
        }
    }
}
