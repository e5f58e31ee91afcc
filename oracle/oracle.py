"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY (see pfhe_oracle.h): imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


def build(native: bool = False, out_dir: str | None = None) -> str:
    """Compile liboracle.so (gcc).  native=True builds a -march=native copy into out_dir."""
    if not native:
        subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"], check=True)
        return os.path.join(_HERE, "liboracle.so")
    out_dir = out_dir or _HERE
    out = os.path.join(out_dir, "liboracle_native.so")
    subprocess.run(
        ["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o", out,
         os.path.join(_HERE, "pfhe_oracle.c"), os.path.join(_HERE, "pfhe_oracle_avx512.c"),
         os.path.join(_HERE, "pfhe_oracle_rns32.c")],
        check=True,
    )
    return out


def _load(path: str | None = None) -> C.CDLL:
    path = path or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    u64, sz, u32, vp, ci = C.c_uint64, C.c_size_t, C.c_uint32, C.c_void_p, C.c_int

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_reduce_once", u64, u64, u64)
    sig("orc_shoup_quotient", u64, u64, u64)
    sig("orc_mul_mod_lazy", u64, u64, u64, u64, u64)
    sig("orc_mul_mod_lazy32", u64, u64, u64, u64, u64)
    sig("orc_shoup_mul", u64, u64, u64, u64, u64)
    sig("orc_barrett_new", ci, u64, vp)
    sig("orc_barrett_lazy_reduce_wide", u64, vp, u64, u64)
    sig("orc_barrett_reduce_wide", u64, vp, u64, u64)
    sig("orc_barrett_reduce", u64, vp, u64)
    sig("orc_barrett_mul", u64, vp, u64, u64)
    sig("orc_barrett_mul_add", u64, vp, u64, u64, u64)
    sig("orc_reduce_add", u64, u64, u64, u64)
    sig("orc_reduce_sub", u64, u64, u64, u64)
    sig("orc_pow_mod", u64, u64, u64, u64)
    sig("orc_inv_mod", u64, u64, u64)
    sig("orc_reduce_mul_slice_assign", None, u64, _u64p, _u64p, sz)
    sig("orc_reduce_add_mul_slice_assign", None, u64, _u64p, _u64p, _u64p, sz)
    sig("orc_minimal_primitive_root", ci, u32, u64, _u64p)

    sig("orc_u64_ntt_new", ci, u32, u64, C.POINTER(vp))
    sig("orc_u64_ntt_free", None, vp)
    for g in ("n",):
        sig("orc_u64_ntt_" + g, sz, vp)
    for g in ("modulus", "root", "inv_root", "inv_n", "inv_n_w"):
        sig("orc_u64_ntt_" + g, u64, vp)
    for g in ("roots", "roots_precon64", "inv_roots", "inv_roots_precon64", "ordinal_roots"):
        sig("orc_u64_ntt_" + g, _u64p, vp)
    sig("orc_u64_ntt_scalar_forward", None, vp, _u64p, u32, u32)
    sig("orc_u64_ntt_scalar_inverse", None, vp, _u64p, u32, u32)
    for g in ("transform_slice", "inverse_transform_slice", "lazy_transform_slice",
              "lazy_inverse_transform_slice"):
        sig("orc_u64_ntt_" + g, None, vp, _u64p)
        sig("orc_uint_ntt_" + g, None, vp, _u64p)
    sig("orc_avx512_available", ci)
    sig("orc_set_vector_backend", None, ci)
    sig("orc_get_vector_backend", ci)
    sig("orc_u64_ntt_forward_avx512", ci, vp, _u64p, ci)
    sig("orc_u64_ntt_inverse_avx512", ci, vp, _u64p, ci)
    sig("orc_avx512_ifma_available", ci)
    sig("orc_u64_ntt_forward_avx512_shift", ci, vp, _u64p, ci, ci)
    sig("orc_u64_ntt_inverse_avx512_shift", ci, vp, _u64p, ci, ci)
    sig("orc_u64_ntt_forward_avx512_batch", ci, vp, _u64p, sz, ci, ci)
    sig("orc_u64_ntt_transform_monomial", None, vp, u64, sz, _u64p)
    sig("orc_u64_ntt_transform_coeff_one_monomial", None, vp, sz, _u64p)
    sig("orc_u64_ntt_transform_coeff_minus_one_monomial", None, vp, sz, _u64p)
    sig("orc_uint_ntt_new", ci, u32, u64, C.POINTER(vp))
    sig("orc_uint_ntt_free", None, vp)
    sig("orc_uint_ntt_transform_monomial", None, vp, u64, sz, _u64p)

    sig("orc_u32_mul_mod_lazy", u32, u32, u32, u32, u32)
    sig("orc_u32_ntt_new", ci, u32, u32, C.POINTER(vp))
    sig("orc_u32_ntt_free", None, vp)
    sig("orc_u32_ntt_n", sz, vp)
    for g in ("modulus", "root", "inv_root", "inv_n", "inv_n_w"):
        sig("orc_u32_ntt_" + g, u32, vp)
    for g in ("roots", "inv_roots"):
        sig("orc_u32_ntt_" + g, _u32p, vp)
    sig("orc_u32_ntt_scalar_forward", None, vp, _u32p, u32)
    sig("orc_u32_ntt_scalar_inverse", None, vp, _u32p, u32)
    for g in ("transform_slice", "inverse_transform_slice", "lazy_transform_slice",
              "lazy_inverse_transform_slice"):
        sig("orc_u32_ntt_" + g, None, vp, _u32p)
    sig("orc_u32_ntt_transform_monomial", None, vp, u32, sz, _u32p)
    sig("orc_u32_ntt_transform_coeff_one_monomial", None, vp, sz, _u32p)
    sig("orc_u32_ntt_transform_coeff_minus_one_monomial", None, vp, sz, _u32p)
    sig("orc_u32_reduce_mul_slice_assign", None, u32, _u32p, _u32p, sz)
    sig("orc_u32_reduce_add_mul_slice_assign", None, u32, _u32p, _u32p, _u32p, sz)

    sig("orc_dcrt_new", ci, u32, _u64p, sz, C.POINTER(vp))
    sig("orc_dcrt_free", None, vp)
    sig("orc_dcrt_poly_length", sz, vp)
    sig("orc_dcrt_moduli_count", sz, vp)
    sig("orc_dcrt_table", vp, vp, sz)
    sig("orc_dcrt_transform_slice", None, vp, _u64p)
    sig("orc_dcrt_inverse_transform_slice", None, vp, _u64p)
    sig("orc_dcrt_poly_mul_assign", None, vp, _u64p, _u64p)
    sig("orc_dcrt_poly_add_mul_assign", None, vp, _u64p, _u64p, _u64p)
    sig("orc_naive_negacyclic_mul", None, u64, _u64p, _u64p, _u64p, sz)
    sig("orc_reduce_neg", u64, u64, u64)
    sig("orc_crt_poly_add_to", None, _u64p, sz, sz, _u64p, _u64p, _u64p)
    sig("orc_crt_poly_sub_to", None, _u64p, sz, sz, _u64p, _u64p, _u64p)
    sig("orc_crt_poly_neg_to", None, _u64p, sz, sz, _u64p, _u64p)
    sig("orc_crt_poly_mul_scalar_to", ci, _u64p, sz, sz, _u64p, _u64p, _u64p)
    sig("orc_crt_poly_add_mul_scalar_assign", ci, _u64p, sz, sz, _u64p, _u64p, _u64p)
    sig("orc_crt_poly_mul_factor_to", None, _u64p, sz, sz, _u64p, _u64p, _u64p)
    sig("orc_crt_poly_add_mul_factor_assign", None, _u64p, sz, sz, _u64p, _u64p, _u64p)
    sig("orc_crt_poly_mul_monomial_assign", ci, _u64p, sz, sz, _u64p, sz)
    sig("orc_dcrt_poly_inv_to", ci, _u64p, sz, sz, _u64p, _u64p)
    sig("orc_dcrt_poly_butterfly_mul_factor_to", None, vp, _u64p, _u64p, _u64p, _u64p)
    sig("orc_dcrt_poly_butterfly_mul_to", None, vp, _u64p, _u64p, _u64p, _u64p)

    sig("orc_rns_new", ci, _u64p, sz, C.POINTER(vp))
    sig("orc_rns_free", None, vp)
    sig("orc_rns_moduli_count", sz, vp)
    sig("orc_rns_value_len", sz, vp)
    sig("orc_rns_moduli_product", _u64p, vp)
    sig("orc_rns_punctured_product", _u64p, vp)
    sig("orc_rns_compose_to", None, vp, _u64p, _u64p)
    sig("orc_rns_compose_multiple_values_to", None, vp, _u64p, _u64p, sz)
    sig("orc_rns_decompose_to", None, vp, _u64p, _u64p)
    sig("orc_rns_decompose_big_uint_values_to", None, vp, _u64p, _u64p, sz)
    sig("orc_rns_wrapping_decompose_small_values_to", None, vp, _u64p, _u64p, sz, u64)
    sig("orc_rns_add_wrapping_decompose_small_values_scaled", None, vp, _u64p, _u64p, sz, u64, _u64p)
    sig("orc_rns_add_decompose_small_values_scaled", None, vp, _u64p, _u64p, sz, _u64p)

    sig("orc_conv_new", ci, vp, vp, C.POINTER(vp))
    sig("orc_conv_free", None, vp)
    sig("orc_conv_matrix", _u64p, vp)
    sig("orc_conv_fast_convert", None, vp, _u64p, _u64p, _u64p)
    sig("orc_conv_fast_convert_array", None, vp, _u64p, _u64p, sz, _u64p)
    sig("orc_conv_exact_convert_array", ci, vp, _u64p, _u64p, sz)

    sig("orc_basis_new", ci, vp, u32, sz, C.POINTER(vp))
    sig("orc_basis_free", None, vp)
    sig("orc_basis_decompose_length", sz, vp)
    sig("orc_basis_log_basis", u32, vp)
    sig("orc_basis_drop_bits", u32, vp)
    sig("orc_basis_basis_value", u64, vp)
    sig("orc_basis_init_mode", ci, vp)
    for g in ("threshold", "adjust_add", "scalars", "scalars_residue"):
        sig("orc_basis_" + g, _u64p, vp)
    sig("orc_basis_init_value_carry_slice_inplace", None, vp, _u64p, _u8p, sz)
    sig("orc_basis_unsigned_decompose_slice_to", None, vp, sz, _u64p, _u64p, _u8p, sz)
    sig("orc_basis_init_value_carry_slice_to", None, vp, _u64p, _u64p, _u8p, sz)
    sig("orc_basis_decompose_slice_to", None, vp, sz, _u64p, _u64p, _u8p, sz)
    sig("orc_add_dcrt_glev_mul_crt_poly_assign", None, vp, vp, vp, sz, _u64p, _u64p, _u64p)
    sig("orc_add_dcrt_glev_mul_big_uint_poly_assign", None, vp, vp, vp, sz, _u64p, _u64p, _u64p)
    sig("orc_mul_dcrt_ggsw_to", None, vp, vp, vp, sz, _u64p, _u64p, _u64p)
    # the <u32> instantiations (pfhe_oracle_rns32.c)
    u32p = C.POINTER(C.c_uint32)
    sig("orc_rns32_new", ci, u32p, sz, C.POINTER(vp))
    sig("orc_rns32_free", None, vp)
    sig("orc_rns32_moduli_count", sz, vp)
    sig("orc_rns32_value_len", sz, vp)
    sig("orc_rns32_moduli_product", u32p, vp)
    sig("orc_rns32_punctured_product", u32p, vp)
    sig("orc_rns32_compose_to", None, vp, u32p, u32p)
    sig("orc_rns32_compose_multiple_values_to", None, vp, u32p, u32p, sz)
    sig("orc_rns32_decompose_to", None, vp, u32p, u32p)
    sig("orc_rns32_decompose_big_uint_values_to", None, vp, u32p, u32p, sz)
    sig("orc_conv32_new", ci, vp, vp, C.POINTER(vp))
    sig("orc_conv32_free", None, vp)
    sig("orc_conv32_matrix", u32p, vp)
    sig("orc_conv32_fast_convert_array", None, vp, u32p, u32p, sz, u32p)
    sig("orc_conv32_exact_convert_array", ci, vp, u32p, u32p, sz)
    sig("orc_rns32_wrapping_decompose_small_values_to", None, vp, u32p, u32p, sz, u32)
    sig("orc_rns32_add_wrapping_decompose_small_values_scaled", None, vp, u32p, u32p, sz, u32, u32p)
    sig("orc_rns32_add_decompose_small_values_scaled", None, vp, u32p, u32p, sz, u32p)
    sig("orc_basis32_new", ci, vp, u32, sz, C.POINTER(vp))
    sig("orc_basis32_free", None, vp)
    sig("orc_basis32_decompose_length", sz, vp)
    for g in ("log_basis", "drop_bits", "basis_value"):
        sig("orc_basis32_" + g, u32, vp)
    sig("orc_basis32_init_mode", ci, vp)
    sig("orc_basis32_scalars", u32p, vp)
    sig("orc_basis32_scalars_residue", u32p, vp)
    sig("orc_basis32_init_value_carry_slice_inplace", None, vp, u32p, _u8p, sz)
    sig("orc_basis32_init_value_carry_slice_to", None, vp, u32p, u32p, _u8p, sz)
    sig("orc_basis32_unsigned_decompose_slice_to", None, vp, sz, u32p, u32p, _u8p, sz)
    sig("orc_basis32_decompose_slice_to", None, vp, sz, u32p, u32p, _u8p, sz)
    sig("orc_add_dcrt32_glev_mul_crt_poly_assign", None, vp, vp, vp, sz, u32p, u32p, u32p)
    sig("orc_add_dcrt32_glev_mul_big_uint_poly_assign", None, vp, vp, vp, sz, u32p, u32p, u32p)
    sig("orc_mul_dcrt32_ggsw_to", None, vp, vp, vp, sz, u32p, u32p, u32p)
    return lib


_lib: C.CDLL | None = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def use_library(path: str) -> None:
    """Switch to another build (e.g. the -march=native copy for the CPU baseline)."""
    global _lib
    _lib = _load(path)


class OracleError(Exception):
    def __init__(self, code: int):
        super().__init__(f"oracle status {code}")
        self.code = code


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    return a.ctypes.data_as(_u64p)


def _arr(ptr, n) -> np.ndarray:
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class Barrett:
    """BarrettModulus<u64> (primus_modulus/src/barrett/mod.rs)."""

    def __init__(self, q: int):
        self._buf = (C.c_uint64 * 3)()
        rc = lib().orc_barrett_new(q, C.addressof(self._buf))
        if rc:
            raise OracleError(rc)
        self.value = q
        self.ratio = (int(self._buf[1]), int(self._buf[2]))

    def _h(self):
        return C.addressof(self._buf)

    def lazy_reduce_wide(self, lo, hi): return int(lib().orc_barrett_lazy_reduce_wide(self._h(), lo, hi))
    def reduce_wide(self, lo, hi): return int(lib().orc_barrett_reduce_wide(self._h(), lo, hi))
    def reduce(self, v): return int(lib().orc_barrett_reduce(self._h(), v))
    def reduce_mul(self, a, b): return int(lib().orc_barrett_mul(self._h(), a, b))
    def reduce_mul_add(self, a, b, c): return int(lib().orc_barrett_mul_add(self._h(), a, b, c))


def minimal_primitive_root(log_degree: int, q: int) -> int:
    r = C.c_uint64(0)
    rc = lib().orc_minimal_primitive_root(log_degree, q, C.byref(r))
    if rc:
        raise OracleError(rc)
    return int(r.value)


class U64NttTable:
    """primus_ntt::U64NttTable, scalar backend (prime64/table.rs, scalar/transform.rs)."""

    def __init__(self, log_n: int, q: int, _handle=None):
        self._own = _handle is None
        if _handle is None:
            h = C.c_void_p()
            rc = lib().orc_u64_ntt_new(log_n, q, C.byref(h))
            if rc:
                raise OracleError(rc)
            _handle = h
        self._h = _handle
        self.n = int(lib().orc_u64_ntt_n(self._h))
        self.log_n = log_n
        self.q = int(lib().orc_u64_ntt_modulus(self._h))

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "_h", None):
            lib().orc_u64_ntt_free(self._h)
            self._h = None

    root = property(lambda s: int(lib().orc_u64_ntt_root(s._h)))
    inv_root = property(lambda s: int(lib().orc_u64_ntt_inv_root(s._h)))
    inv_n = property(lambda s: int(lib().orc_u64_ntt_inv_n(s._h)))
    inv_n_w = property(lambda s: int(lib().orc_u64_ntt_inv_n_w(s._h)))
    roots = property(lambda s: _arr(lib().orc_u64_ntt_roots(s._h), s.n))
    roots_precon64 = property(lambda s: _arr(lib().orc_u64_ntt_roots_precon64(s._h), s.n))
    inv_roots = property(lambda s: _arr(lib().orc_u64_ntt_inv_roots(s._h), s.n))
    inv_roots_precon64 = property(lambda s: _arr(lib().orc_u64_ntt_inv_roots_precon64(s._h), s.n))
    ordinal_roots = property(lambda s: _arr(lib().orc_u64_ntt_ordinal_roots(s._h), 2 * s.n))

    def _each(self, fn, a: np.ndarray):
        assert a.size % self.n == 0
        flat = a.reshape(-1)
        base = flat.ctypes.data
        for i in range(flat.size // self.n):
            fn(self._h, C.cast(base + 8 * self.n * i, _u64p))

    def transform_slice(self, a): self._each(lib().orc_u64_ntt_transform_slice, a)
    def inverse_transform_slice(self, a): self._each(lib().orc_u64_ntt_inverse_transform_slice, a)
    def lazy_transform_slice(self, a): self._each(lib().orc_u64_ntt_lazy_transform_slice, a)
    def lazy_inverse_transform_slice(self, a): self._each(lib().orc_u64_ntt_lazy_inverse_transform_slice, a)

    def transform_slice_avx512(self, a, lazy: bool = False, shift: int = 0):
        """Forward transform through the AVX-512 backend (prime64/avx512/): shift 64 = DQ rung, 52 = IFMA rung
        (q < 2^50), 0 = the reference's dispatch (table.rs:166-232); raises when unavailable."""
        assert a.size % self.n == 0
        flat = a.reshape(-1)
        base = flat.ctypes.data
        for i in range(flat.size // self.n):
            rc = lib().orc_u64_ntt_forward_avx512_shift(self._h, C.cast(base + 8 * self.n * i, _u64p), int(lazy), shift)
            if rc:
                raise OracleError(rc)

    def transform_batch_avx512(self, a, lazy: bool = False, shift: int = 0):
        """Every polynomial of `a` forward-transformed inside ONE foreign call (no per-polynomial Python dispatch)."""
        assert a.size % self.n == 0
        rc = lib().orc_u64_ntt_forward_avx512_batch(self._h, _p(a.reshape(-1)), a.size // self.n, int(lazy), shift)
        if rc:
            raise OracleError(rc)

    def inverse_transform_slice_avx512(self, a, lazy: bool = False, shift: int = 0):
        """Inverse transform through the AVX-512 backend (prime64/avx512/transform.rs:205); shift as above."""
        assert a.size % self.n == 0
        flat = a.reshape(-1)
        base = flat.ctypes.data
        for i in range(flat.size // self.n):
            rc = lib().orc_u64_ntt_inverse_avx512_shift(self._h, C.cast(base + 8 * self.n * i, _u64p), int(lazy), shift)
            if rc:
                raise OracleError(rc)

    def scalar_forward(self, a, bit_shift, output_mod_factor):
        lib().orc_u64_ntt_scalar_forward(self._h, _p(a), bit_shift, output_mod_factor)

    def scalar_inverse(self, a, bit_shift, output_mod_factor):
        lib().orc_u64_ntt_scalar_inverse(self._h, _p(a), bit_shift, output_mod_factor)

    def transform_monomial(self, coeff, degree):
        out = np.empty(self.n, np.uint64)
        lib().orc_u64_ntt_transform_monomial(self._h, coeff, degree, _p(out))
        return out

    def transform_coeff_one_monomial(self, degree):
        out = np.empty(self.n, np.uint64)
        lib().orc_u64_ntt_transform_coeff_one_monomial(self._h, degree, _p(out))
        return out

    def transform_coeff_minus_one_monomial(self, degree):
        out = np.empty(self.n, np.uint64)
        lib().orc_u64_ntt_transform_coeff_minus_one_monomial(self._h, degree, _p(out))
        return out


def _p32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags.c_contiguous
    return a.ctypes.data_as(_u32p)


class U32NttTable:
    """primus_ntt::U32NttTable, scalar backend (prime32/table.rs, prime32/scalar/transform.rs)."""

    def __init__(self, log_n: int, q: int):
        h = C.c_void_p()
        rc = lib().orc_u32_ntt_new(log_n, q, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.n = 1 << log_n
        self.log_n = log_n
        self.q = q

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_u32_ntt_free(self._h)
            self._h = None

    root = property(lambda s: int(lib().orc_u32_ntt_root(s._h)))
    inv_root = property(lambda s: int(lib().orc_u32_ntt_inv_root(s._h)))
    inv_n = property(lambda s: int(lib().orc_u32_ntt_inv_n(s._h)))
    inv_n_w = property(lambda s: int(lib().orc_u32_ntt_inv_n_w(s._h)))
    roots = property(lambda s: np.ctypeslib.as_array(lib().orc_u32_ntt_roots(s._h), (s.n,)).copy())
    inv_roots = property(lambda s: np.ctypeslib.as_array(lib().orc_u32_ntt_inv_roots(s._h), (s.n,)).copy())

    def _each(self, fn, a: np.ndarray):
        assert a.dtype == np.uint32 and a.flags.c_contiguous and a.size % self.n == 0
        flat = a.reshape(-1)
        base = flat.ctypes.data
        for i in range(flat.size // self.n):
            fn(self._h, C.cast(base + 4 * self.n * i, _u32p))

    def transform_slice(self, a): self._each(lib().orc_u32_ntt_transform_slice, a)
    def inverse_transform_slice(self, a): self._each(lib().orc_u32_ntt_inverse_transform_slice, a)
    def lazy_transform_slice(self, a): self._each(lib().orc_u32_ntt_lazy_transform_slice, a)
    def lazy_inverse_transform_slice(self, a): self._each(lib().orc_u32_ntt_lazy_inverse_transform_slice, a)

    def transform_monomial(self, coeff, degree):
        out = np.empty(self.n, np.uint32)
        lib().orc_u32_ntt_transform_monomial(self._h, coeff, degree, _p32(out))
        return out

    def transform_coeff_one_monomial(self, degree):
        out = np.empty(self.n, np.uint32)
        lib().orc_u32_ntt_transform_coeff_one_monomial(self._h, degree, _p32(out))
        return out

    def transform_coeff_minus_one_monomial(self, degree):
        out = np.empty(self.n, np.uint32)
        lib().orc_u32_ntt_transform_coeff_minus_one_monomial(self._h, degree, _p32(out))
        return out

    def mul_assign(self, a, b):
        lib().orc_u32_reduce_mul_slice_assign(self.q, _p32(a), _p32(b), a.size)

    def add_mul_assign(self, acc, a, b):
        lib().orc_u32_reduce_add_mul_slice_assign(self.q, _p32(acc), _p32(a), _p32(b), a.size)


class U32DcrtTable:
    """primus_ntt::U32DcrtTable (dcrt/prime32.rs): one U32NttTable per limb, modulus-major data."""

    def __init__(self, log_n: int, moduli):
        self.moduli = [int(m) for m in moduli]
        self.tables = [U32NttTable(log_n, q) for q in self.moduli]
        self.log_n, self.n, self.count = log_n, 1 << log_n, len(self.moduli)
        self.crt_poly_length = self.n * self.count

    def _each(self, name, a):
        assert a.size % self.crt_poly_length == 0
        v = a.reshape(-1, self.count, self.n)
        for e in range(v.shape[0]):
            for r, t in enumerate(self.tables):
                getattr(t, name)(v[e, r])

    def transform_slice(self, a): self._each("transform_slice", a)
    def inverse_transform_slice(self, a): self._each("inverse_transform_slice", a)
    def lazy_transform_slice(self, a): self._each("lazy_transform_slice", a)
    def lazy_inverse_transform_slice(self, a): self._each("lazy_inverse_transform_slice", a)

    def mul_assign(self, a, b):
        va, vb = a.reshape(-1, self.count, self.n), b.reshape(-1, self.count, self.n)
        for e in range(va.shape[0]):
            for r, t in enumerate(self.tables):
                t.mul_assign(va[e, r], vb[e if vb.shape[0] > 1 else 0, r])

    def add_mul_assign(self, acc, a, b):
        vc, va, vb = (x.reshape(-1, self.count, self.n) for x in (acc, a, b))
        for e in range(va.shape[0]):
            for r, t in enumerate(self.tables):
                t.add_mul_assign(vc[e, r], va[e, r], vb[e if vb.shape[0] > 1 else 0, r])

    def transform_monomial(self, coeff, degree):
        return np.concatenate([t.transform_monomial(coeff, degree) for t in self.tables])

    def transform_coeff_one_monomial(self, degree):
        return np.concatenate([t.transform_coeff_one_monomial(degree) for t in self.tables])

    def transform_coeff_minus_one_monomial(self, degree):
        return np.concatenate([t.transform_coeff_minus_one_monomial(degree) for t in self.tables])


class UintNttTable:
    """primus_ntt::UintNttTable<u64> (ntt/primitive.rs) — the reference's own cross-check."""

    def __init__(self, log_n: int, q: int):
        h = C.c_void_p()
        rc = lib().orc_uint_ntt_new(log_n, q, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.n = 1 << log_n
        self.q = q

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_uint_ntt_free(self._h)
            self._h = None

    def transform_slice(self, a): lib().orc_uint_ntt_transform_slice(self._h, _p(a))
    def inverse_transform_slice(self, a): lib().orc_uint_ntt_inverse_transform_slice(self._h, _p(a))
    def lazy_transform_slice(self, a): lib().orc_uint_ntt_lazy_transform_slice(self._h, _p(a))
    def lazy_inverse_transform_slice(self, a): lib().orc_uint_ntt_lazy_inverse_transform_slice(self._h, _p(a))

    def transform_monomial(self, coeff, degree):
        out = np.empty(self.n, np.uint64)
        lib().orc_uint_ntt_transform_monomial(self._h, coeff, degree, _p(out))
        return out


class U64DcrtTable:
    """primus_ntt::U64DcrtTable (dcrt/prime64.rs): modulus-major L x N words per polynomial."""

    def __init__(self, log_n: int, moduli):
        self.moduli = [int(m) for m in moduli]
        arr = np.array(self.moduli, np.uint64)
        h = C.c_void_p()
        rc = lib().orc_dcrt_new(log_n, _p(arr), len(self.moduli), C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.log_n = log_n
        self.n = 1 << log_n
        self.count = len(self.moduli)
        self.crt_poly_length = self.n * self.count

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_dcrt_free(self._h)
            self._h = None

    def table(self, i) -> U64NttTable:
        return U64NttTable(self.log_n, self.moduli[i], _handle=C.c_void_p(lib().orc_dcrt_table(self._h, i)))

    def _each(self, fn, a):
        W = self.crt_poly_length
        assert a.size % W == 0
        flat = a.reshape(-1)
        base = flat.ctypes.data
        for i in range(flat.size // W):
            fn(self._h, C.cast(base + 8 * W * i, _u64p))

    def transform_slice(self, a): self._each(lib().orc_dcrt_transform_slice, a)
    def inverse_transform_slice(self, a): self._each(lib().orc_dcrt_inverse_transform_slice, a)

    def mul_assign(self, a, b):
        lib().orc_dcrt_poly_mul_assign(self._h, _p(a), _p(b))

    def add_mul_assign(self, acc, a, b):
        lib().orc_dcrt_poly_add_mul_assign(self._h, _p(acc), _p(a), _p(b))

    def butterfly_mul_factor_to(self, a, s, w_pairs):
        """(a, b) = (a + s, (a - s) * w) with ShoupFactor pairs; a is updated in place, b returned."""
        b = np.empty_like(a)
        lib().orc_dcrt_poly_butterfly_mul_factor_to(self._h, _p(a), _p(s), _p(w_pairs), _p(b))
        return b

    def butterfly_mul_to(self, a, s, w):
        b = np.empty_like(a)
        lib().orc_dcrt_poly_butterfly_mul_to(self._h, _p(a), _p(s), _p(w), _p(b))
        return b


class CrtPolyOps:
    """The element-wise family of primus_poly's CrtPolynomial / DcrtPolynomial (crt/{add,sub,neg,mul}.rs,
    dcrt/inv.rs) over a batch of RNS polynomials (L limbs x n words each, modulus-major); CrtGlwe's
    add_element_wise* / mul_scalar_* / mul_factor_to / mul_monic_monomial_assign are the same loops."""

    def __init__(self, moduli, n: int):
        self.moduli = np.ascontiguousarray(np.array([int(m) for m in moduli], dtype=np.uint64))
        self.L, self.n = self.moduli.size, int(n)
        self.unit = self.L * self.n

    def _each(self, *arrays):
        count = arrays[0].size // self.unit
        assert all(a.size == count * self.unit for a in arrays)
        for e in range(count):
            yield [a[e * self.unit:(e + 1) * self.unit] for a in arrays]

    def _binary(self, fn, a, b):
        out = np.empty_like(a)
        for x, y, o in self._each(a, b, out):
            fn(_p(self.moduli), self.L, self.n, _p(x), _p(y), _p(o))
        return out

    def add_to(self, a, b): return self._binary(lib().orc_crt_poly_add_to, a, b)
    def sub_to(self, a, b): return self._binary(lib().orc_crt_poly_sub_to, a, b)

    def neg_to(self, a):
        out = np.empty_like(a)
        for x, o in self._each(a, out):
            lib().orc_crt_poly_neg_to(_p(self.moduli), self.L, self.n, _p(x), _p(o))
        return out

    def _words(self, values, per_limb):
        arr = np.ascontiguousarray(np.array(values, dtype=np.uint64).reshape(-1))
        assert arr.size == per_limb * self.L
        return arr

    def mul_scalar_to(self, a, scalars):
        sc, out = self._words(scalars, 1), np.empty_like(a)
        for x, o in self._each(a, out):
            assert lib().orc_crt_poly_mul_scalar_to(_p(self.moduli), self.L, self.n, _p(x), _p(sc), _p(o)) == 0
        return out

    def add_mul_scalar_assign(self, acc, rhs, scalars):
        sc = self._words(scalars, 1)
        for c, r in self._each(acc, rhs):
            assert lib().orc_crt_poly_add_mul_scalar_assign(_p(self.moduli), self.L, self.n, _p(c), _p(r), _p(sc)) == 0

    def mul_factor_to(self, a, factors):
        f, out = self._words(factors, 2), np.empty_like(a)
        for x, o in self._each(a, out):
            lib().orc_crt_poly_mul_factor_to(_p(self.moduli), self.L, self.n, _p(x), _p(f), _p(o))
        return out

    def add_mul_factor_assign(self, acc, rhs, factors):
        f = self._words(factors, 2)
        for c, r in self._each(acc, rhs):
            lib().orc_crt_poly_add_mul_factor_assign(_p(self.moduli), self.L, self.n, _p(c), _p(r), _p(f))

    def mul_monomial_assign(self, data, r: int):
        for (x,) in self._each(data):
            if lib().orc_crt_poly_mul_monomial_assign(_p(self.moduli), self.L, self.n, _p(x), r):
                raise ValueError("monomial degree must be below 2N")

    def inv_to(self, a):
        out = np.empty_like(a)
        for x, o in self._each(a, out):
            if lib().orc_dcrt_poly_inv_to(_p(self.moduli), self.L, self.n, _p(x), _p(o)):
                raise ZeroDivisionError("element has no inverse")  # the reference panics
        return out


def naive_negacyclic_mul(q, a, b):
    out = np.empty_like(a)
    lib().orc_naive_negacyclic_mul(q, _p(a), _p(b), _p(out), a.size)
    return out


class RNSBase:
    """primus_rns::RNSBase<u64, BarrettModulus<u64>> (base.rs)."""

    def __init__(self, moduli):
        self.moduli = [int(m) for m in moduli]
        arr = np.array(self.moduli, np.uint64)
        h = C.c_void_p()
        rc = lib().orc_rns_new(_p(arr) if len(arr) else None, len(self.moduli), C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.count = len(self.moduli)
        self.value_len = int(lib().orc_rns_value_len(h))
        self.moduli_product = _arr(lib().orc_rns_moduli_product(h), self.value_len)
        self.punctured_product = _arr(lib().orc_rns_punctured_product(h), self.value_len * self.count)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rns_free(self._h)
            self._h = None

    def compose(self, residues):
        r = np.array(residues, np.uint64)
        out = np.empty(self.value_len, np.uint64)
        lib().orc_rns_compose_to(self._h, _p(r), _p(out))
        return out

    def compose_multiple_values_to(self, multi_residues, value_count):
        out = np.empty(value_count * self.value_len, np.uint64)
        lib().orc_rns_compose_multiple_values_to(self._h, _p(multi_residues), _p(out), value_count)
        return out

    def decompose(self, value):
        out = np.empty(self.count, np.uint64)
        lib().orc_rns_decompose_to(self._h, _p(np.ascontiguousarray(value, np.uint64)), _p(out))
        return out

    def decompose_big_uint_values_to(self, values, value_count):
        out = np.empty(self.count * value_count, np.uint64)
        lib().orc_rns_decompose_big_uint_values_to(self._h, _p(values), _p(out), value_count)
        return out

    def wrapping_decompose_small_values_to(self, small_values, small_value_modulus):
        sv = np.ascontiguousarray(small_values, np.uint64)
        out = np.empty(self.count * sv.size, np.uint64)
        lib().orc_rns_wrapping_decompose_small_values_to(self._h, _p(sv), _p(out), sv.size, small_value_modulus)
        return out


    def add_wrapping_decompose_small_values_scaled(self, small_values, acc, small_value_modulus, factors):
        """base.rs:326-384; acc (modulus-major, count * len(small_values) words) is updated in place."""
        sv = np.ascontiguousarray(small_values, np.uint64)
        f = np.ascontiguousarray(np.array(factors, np.uint64).reshape(-1))
        assert acc.size == self.count * sv.size and f.size == 2 * self.count
        lib().orc_rns_add_wrapping_decompose_small_values_scaled(self._h, _p(sv), _p(acc), sv.size, small_value_modulus, _p(f))

    def add_decompose_small_values_scaled(self, small_values, acc, factors):
        """base.rs:398-416."""
        sv = np.ascontiguousarray(small_values, np.uint64)
        f = np.ascontiguousarray(np.array(factors, np.uint64).reshape(-1))
        assert acc.size == self.count * sv.size and f.size == 2 * self.count
        lib().orc_rns_add_decompose_small_values_scaled(self._h, _p(sv), _p(acc), sv.size, _p(f))


class BaseConverter:
    """primus_rns::BaseConverter<u64, BarrettModulus<u64>> (converter.rs)."""

    def __init__(self, input_base: RNSBase, output_base: RNSBase):
        h = C.c_void_p()
        rc = lib().orc_conv_new(input_base._h, output_base._h, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.input_base, self.output_base = input_base, output_base  # keep the borrowed bases alive
        self.base_change_matrix = _arr(lib().orc_conv_matrix(h), input_base.count * output_base.count)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_conv_free(self._h)
            self._h = None

    def fast_convert(self, residues_in):
        r = np.ascontiguousarray(residues_in, np.uint64)
        out = np.empty(self.output_base.count, np.uint64)
        scratch = np.empty(self.input_base.count, np.uint64)
        lib().orc_conv_fast_convert(self._h, _p(r), _p(out), _p(scratch))
        return out

    def fast_convert_array(self, crt_poly_in, poly_length):
        out = np.empty(self.output_base.count * poly_length, np.uint64)
        scratch = np.empty(self.input_base.count * poly_length, np.uint64)
        lib().orc_conv_fast_convert_array(self._h, _p(crt_poly_in), _p(out), poly_length, _p(scratch))
        return out

    def exact_convert_array(self, crt_poly_in, poly_length):
        out = np.empty(poly_length, np.uint64)
        rc = lib().orc_conv_exact_convert_array(self._h, _p(crt_poly_in), _p(out), poly_length)
        if rc:
            raise OracleError(rc)
        return out


class BigUintApproxSignedBasis:
    """primus_decompose::big_integer::BigUintApproxSignedBasis<u64> (basis.rs, common.rs)."""

    def __init__(self, rns: RNSBase, log_basis: int, reverse_length: int | None = None):
        h = C.c_void_p()
        rc = lib().orc_basis_new(rns._h, log_basis, reverse_length or 0, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.rns = rns
        self.decompose_length = int(lib().orc_basis_decompose_length(h))
        self.log_basis = int(lib().orc_basis_log_basis(h))
        self.drop_bits = int(lib().orc_basis_drop_bits(h))
        self.basis_value = int(lib().orc_basis_basis_value(h))
        self.init_mode = int(lib().orc_basis_init_mode(h))
        L = rns.value_len
        self.threshold = _arr(lib().orc_basis_threshold(h), L)
        self.adjust_add = _arr(lib().orc_basis_adjust_add(h), L)
        self.scalars = _arr(lib().orc_basis_scalars(h), L * self.decompose_length)
        self.scalars_residue = _arr(lib().orc_basis_scalars_residue(h), rns.count * self.decompose_length)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_basis_free(self._h)
            self._h = None

    def init_value_carry_slice_inplace(self, values, count):
        carries = np.zeros(count, np.uint8)
        lib().orc_basis_init_value_carry_slice_inplace(self._h, _p(values), carries.ctypes.data_as(_u8p), count)
        return carries

    def unsigned_decompose_slice_to(self, level, values, carries, count):
        digits = np.empty(count, np.uint64)
        lib().orc_basis_unsigned_decompose_slice_to(self._h, level, _p(values), _p(digits),
                                                    carries.ctypes.data_as(_u8p), count)
        return digits

    def init_value_carry_slice_to(self, values, count):
        """basis.rs:371-420: (adjusted values, carries), input untouched."""
        adjusted, carries = np.empty_like(values), np.zeros(count, np.uint8)
        lib().orc_basis_init_value_carry_slice_to(self._h, _p(values), _p(adjusted), carries.ctypes.data_as(_u8p), count)
        return adjusted, carries

    def decompose_slice_to(self, level, values, carries, count):
        """common.rs:289-306: signed digits as residues modulo Q (value_len limbs each); carries updated."""
        out = np.empty(values.size, np.uint64)
        lib().orc_basis_decompose_slice_to(self._h, level, _p(values), _p(out), carries.ctypes.data_as(_u8p), count)
        return out


def add_dcrt_glev_mul_crt_poly_assign(table: U64DcrtTable, rns: RNSBase, basis, k, acc, glev, crt_poly):
    lib().orc_add_dcrt_glev_mul_crt_poly_assign(table._h, rns._h, basis._h, k, _p(acc), _p(glev), _p(crt_poly))


def add_dcrt_glev_mul_big_uint_poly_assign(table: U64DcrtTable, rns: RNSBase, basis, k, acc, glev, big_uint_poly):
    """glwe/dcrt.rs:258-338: the polynomial as value_len-limb big integers modulo Q."""
    lib().orc_add_dcrt_glev_mul_big_uint_poly_assign(table._h, rns._h, basis._h, k, _p(acc), _p(glev), _p(big_uint_poly))


def mul_dcrt_ggsw_to(table: U64DcrtTable, rns: RNSBase, basis, k, crt_glwe, dcrt_ggsw):
    """CrtGlwe::mul_dcrt_ggsw_to; returns the DcrtGlwe result ((k+1)*L*N words, NTT form)."""
    out = np.empty((k + 1) * table.crt_poly_length, np.uint64)
    lib().orc_mul_dcrt_ggsw_to(table._h, rns._h, basis._h, k, _p(crt_glwe), _p(dcrt_ggsw), _p(out))
    return out


# ---------------------------------------------------------------------------------------------
# The <u32> instantiations (pfhe_oracle_rns32.c): RNSBase<u32>, BigUintApproxSignedBasis<u32>, external product over
# U32DcrtTable.  numpy uint32 arrays throughout.
# ---------------------------------------------------------------------------------------------
def _arr32(ptr, n) -> np.ndarray:
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.empty(0, np.uint32)


class RNSBase32:
    """primus_rns::RNSBase<u32, BarrettModulus<u32>> (base.rs:26-117 with T = u32)."""

    def __init__(self, moduli):
        self.moduli = [int(m) for m in moduli]
        arr = np.array(self.moduli, np.uint32)
        h = C.c_void_p()
        rc = lib().orc_rns32_new(_p32(arr) if len(arr) else None, len(self.moduli), C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.count = len(self.moduli)
        self.value_len = int(lib().orc_rns32_value_len(h))
        self.moduli_product = _arr32(lib().orc_rns32_moduli_product(h), self.value_len)
        self.punctured_product = _arr32(lib().orc_rns32_punctured_product(h), self.value_len * self.count)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rns32_free(self._h)
            self._h = None

    def compose_multiple_values_to(self, multi_residues, value_count):
        out = np.empty(value_count * self.value_len, np.uint32)
        lib().orc_rns32_compose_multiple_values_to(self._h, _p32(multi_residues), _p32(out), value_count)
        return out

    def decompose_big_uint_values_to(self, values, value_count):
        out = np.empty(self.count * value_count, np.uint32)
        lib().orc_rns32_decompose_big_uint_values_to(self._h, _p32(values), _p32(out), value_count)
        return out

    def wrapping_decompose_small_values_to(self, small_values, small_value_modulus):
        sv = np.ascontiguousarray(small_values, np.uint32)
        out = np.empty(self.count * sv.size, np.uint32)
        lib().orc_rns32_wrapping_decompose_small_values_to(self._h, _p32(sv), _p32(out), sv.size, small_value_modulus)
        return out

    def add_wrapping_decompose_small_values_scaled(self, small_values, acc, small_value_modulus, factors):
        sv = np.ascontiguousarray(small_values, np.uint32)
        f = np.ascontiguousarray(np.array(factors, np.uint32).reshape(-1))
        assert acc.size == self.count * sv.size and f.size == 2 * self.count
        lib().orc_rns32_add_wrapping_decompose_small_values_scaled(self._h, _p32(sv), _p32(acc), sv.size, small_value_modulus,
                                                                   _p32(f))

    def add_decompose_small_values_scaled(self, small_values, acc, factors):
        sv = np.ascontiguousarray(small_values, np.uint32)
        f = np.ascontiguousarray(np.array(factors, np.uint32).reshape(-1))
        assert acc.size == self.count * sv.size and f.size == 2 * self.count
        lib().orc_rns32_add_decompose_small_values_scaled(self._h, _p32(sv), _p32(acc), sv.size, _p32(f))


class BaseConverter32:
    """primus_rns::BaseConverter<u32, BarrettModulus<u32>> (converter.rs with T = u32)."""

    def __init__(self, input_base: "RNSBase32", output_base: "RNSBase32"):
        h = C.c_void_p()
        rc = lib().orc_conv32_new(input_base._h, output_base._h, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.input_base, self.output_base = input_base, output_base  # keep the borrowed bases alive
        self.base_change_matrix = _arr32(lib().orc_conv32_matrix(h), input_base.count * output_base.count)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_conv32_free(self._h)
            self._h = None

    def fast_convert_array(self, crt_poly_in, poly_length):
        out = np.empty(self.output_base.count * poly_length, np.uint32)
        scratch = np.empty(self.input_base.count * poly_length, np.uint32)
        lib().orc_conv32_fast_convert_array(self._h, _p32(crt_poly_in), _p32(out), poly_length, _p32(scratch))
        return out

    def exact_convert_array(self, crt_poly_in, poly_length):
        out = np.empty(poly_length, np.uint32)
        rc = lib().orc_conv32_exact_convert_array(self._h, _p32(crt_poly_in), _p32(out), poly_length)
        if rc:
            raise OracleError(rc)
        return out


class BigUintApproxSignedBasis32:
    """primus_decompose::big_integer::BigUintApproxSignedBasis<u32> (basis.rs:33; tests/big_uint.rs:13)."""

    def __init__(self, rns: RNSBase32, log_basis: int, reverse_length: int | None = None):
        h = C.c_void_p()
        rc = lib().orc_basis32_new(rns._h, log_basis, reverse_length or 0, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self.rns = rns
        self.decompose_length = int(lib().orc_basis32_decompose_length(h))
        self.log_basis = int(lib().orc_basis32_log_basis(h))
        self.drop_bits = int(lib().orc_basis32_drop_bits(h))
        self.basis_value = int(lib().orc_basis32_basis_value(h))
        self.init_mode = int(lib().orc_basis32_init_mode(h))
        self.scalars = _arr32(lib().orc_basis32_scalars(h), rns.value_len * self.decompose_length)
        self.scalars_residue = _arr32(lib().orc_basis32_scalars_residue(h), rns.count * self.decompose_length)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_basis32_free(self._h)
            self._h = None

    def init_value_carry_slice_inplace(self, values, count):
        carries = np.zeros(count, np.uint8)
        lib().orc_basis32_init_value_carry_slice_inplace(self._h, _p32(values), carries.ctypes.data_as(_u8p), count)
        return carries

    def init_value_carry_slice_to(self, values, count):
        adjusted, carries = np.empty_like(values), np.zeros(count, np.uint8)
        lib().orc_basis32_init_value_carry_slice_to(self._h, _p32(values), _p32(adjusted), carries.ctypes.data_as(_u8p), count)
        return adjusted, carries

    def unsigned_decompose_slice_to(self, level, values, carries, count):
        digits = np.empty(count, np.uint32)
        lib().orc_basis32_unsigned_decompose_slice_to(self._h, level, _p32(values), _p32(digits),
                                                      carries.ctypes.data_as(_u8p), count)
        return digits

    def decompose_slice_to(self, level, values, carries, count):
        out = np.empty(values.size, np.uint32)
        lib().orc_basis32_decompose_slice_to(self._h, level, _p32(values), _p32(out), carries.ctypes.data_as(_u8p), count)
        return out


def _tables32(table: U32DcrtTable):
    return (C.c_void_p * table.count)(*[t._h for t in table.tables])


def mul_dcrt32_ggsw_to(table: U32DcrtTable, rns: RNSBase32, basis: BigUintApproxSignedBasis32, k, crt_glwe, dcrt_ggsw):
    """CrtGlwe<u32>::mul_dcrt_ggsw_to; returns the DcrtGlwe result ((k+1)*L*N u32 words, NTT form)."""
    out = np.empty((k + 1) * table.crt_poly_length, np.uint32)
    lib().orc_mul_dcrt32_ggsw_to(_tables32(table), rns._h, basis._h, k, _p32(crt_glwe), _p32(dcrt_ggsw), _p32(out))
    return out


def add_dcrt32_glev_mul_crt_poly_assign(table: U32DcrtTable, rns: RNSBase32, basis, k, acc, glev, crt_poly):
    lib().orc_add_dcrt32_glev_mul_crt_poly_assign(_tables32(table), rns._h, basis._h, k, _p32(acc), _p32(glev), _p32(crt_poly))


def add_dcrt32_glev_mul_big_uint_poly_assign(table: U32DcrtTable, rns: RNSBase32, basis, k, acc, glev, big_uint_poly):
    lib().orc_add_dcrt32_glev_mul_big_uint_poly_assign(_tables32(table), rns._h, basis._h, k, _p32(acc), _p32(glev),
                                                       _p32(big_uint_poly))
