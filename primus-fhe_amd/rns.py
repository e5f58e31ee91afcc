"""RNSBase and BigUintApproxSignedBasis — host-side mirror of primus_rns / primus_decompose.

Reference: primus_rns/src/base.rs:26 (RNSBase<u64, BarrettModulus<u64>>),
primus_decompose/src/big_integer/basis.rs:17 (BigUintApproxSignedBasis<u64>),
primus_decompose/src/big_integer/common.rs:242 (OnceBigUintSignedDecomposer).
Every method runs a HIP kernel through the C ABI; `*_to` names keep the reference's meaning.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import PfheError, check, lib, u64p
from .ntt import _dev, _dev32, _host, _host32, _stream

RNSError = PfheError


class RNSBase:
    """primus_rns::RNSBase<u64, BarrettModulus<u64>> — pairwise-coprime basis with CRT precomputations (base.rs:26-117).
    RNSBase32 below is the <u32> instantiation: the same methods on numpy uint32 / 32-bit CUDA tensors."""

    _pre, _dtype, _wp = "pfhe_rns_", np.uint64, u64p
    _host, _dev = staticmethod(_host), staticmethod(_dev)

    def _f(self, name):
        return getattr(lib(), self._pre + name)

    def __init__(self, moduli, device: int = 0):
        arr = np.ascontiguousarray(np.array([int(m) for m in moduli], dtype=self._dtype))
        h = C.c_void_p()
        check(self._f("create")(arr.ctypes.data_as(self._wp) if arr.size else None, arr.size, device, C.byref(h)))
        self._h = h
        self._moduli = [int(m) for m in moduli]

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._f("destroy")(h)
            self._h = None

    def moduli(self): return list(self._moduli)
    def moduli_count(self) -> int: return int(self._f("moduli_count")(self._h))
    def big_uint_value_len(self) -> int: return int(self._f("big_uint_value_len")(self._h))

    def moduli_product(self) -> np.ndarray:
        out = np.empty(self.big_uint_value_len(), self._dtype)
        check(self._f("moduli_product")(self._h, *self._host(out)))
        return out

    def compose_multiple_values_to(self, multi_residues, big_uint_values, value_count: int):
        """base.rs:648-675; `scratch` of the reference is owned by the kernel (registers)."""
        pi, ni = self._host(multi_residues)
        po, no = self._host(big_uint_values)
        check(self._f("compose_multiple_values_to")(self._h, pi, ni, po, no, value_count))

    compose_polynomial_to = compose_multiple_values_to  # base.rs:690-706

    def wrapping_decompose_small_values_to(self, small_values, multi_residues, value_count: int,
                                           small_value_modulus: int):
        """base.rs:279-312."""
        pi, ni = self._host(small_values)
        po, no = self._host(multi_residues)
        if ni != value_count:
            raise PfheError(32, "small_values.len() must equal value_count")
        check(self._f("wrapping_decompose_small_values_to")(self._h, pi, value_count, po, no, small_value_modulus))

    def compose_multiple_values_to_dev(self, multi_residues, big_uint_values, value_count: int, stream=None):
        (pi, ni), (po, no) = self._dev(multi_residues), self._dev(big_uint_values)
        check(self._f("compose_multiple_values_to_dev")(self._h, pi, ni, po, no, value_count, _stream(stream)))

    def wrapping_decompose_small_values_to_dev(self, small_values, multi_residues, value_count: int,
                                               small_value_modulus: int, stream=None):
        (pi, ni), (po, no) = self._dev(small_values), self._dev(multi_residues)
        if ni != value_count:
            raise PfheError(32, "small_values.len() must equal value_count")
        check(self._f("wrapping_decompose_small_values_to_dev")(self._h, pi, value_count, po, no,
                                                                    small_value_modulus, _stream(stream)))


    def _factor_words(self, factors, count):
        f = np.ascontiguousarray(np.array(factors, dtype=self._dtype).reshape(-1))
        if f.size != 2 * count:
            raise PfheError(32, "expected one (value, quotient) pair per modulus")
        return f

    def add_wrapping_decompose_small_values_scaled(self, small_values, acc, value_count: int, small_value_modulus: int,
                                                   factors):
        """base.rs:326-384: acc[i][c] += factor_i * centred_lift_i(small[c]) (in place; factors = ShoupFactor pairs)."""
        (pi, ni), (pa, na) = self._host(small_values), self._host(acc)
        if ni != value_count:
            raise PfheError(32, "small_values.len() must equal value_count")
        f = self._factor_words(factors, self.moduli_count())
        check(self._f("add_wrapping_decompose_small_values_scaled")(self._h, pi, value_count, pa, na,
                                                                        small_value_modulus, f.ctypes.data_as(self._wp)))

    def add_decompose_small_values_scaled(self, small_values, acc, value_count: int, factors):
        """base.rs:398-416 (= add_decompose_small_polynomial_scaled, :429-443): acc[i][c] += factor_i * small[c]."""
        (pi, ni), (pa, na) = self._host(small_values), self._host(acc)
        if ni != value_count:
            raise PfheError(32, "small_values.len() must equal value_count")
        f = self._factor_words(factors, self.moduli_count())
        check(self._f("add_decompose_small_values_scaled")(self._h, pi, value_count, pa, na, f.ctypes.data_as(self._wp)))

    add_decompose_small_polynomial_scaled = add_decompose_small_values_scaled

    def add_wrapping_decompose_small_values_scaled_dev(self, small_values, acc, value_count: int,
                                                       small_value_modulus: int, factors, stream=None):
        (pi, ni), (pa, na) = self._dev(small_values), self._dev(acc)
        if ni != value_count:
            raise PfheError(32, "small_values.len() must equal value_count")
        f = self._factor_words(factors, self.moduli_count())
        check(self._f("add_wrapping_decompose_small_values_scaled_dev")(self._h, pi, value_count, pa, na,
                                                                            small_value_modulus, f.ctypes.data_as(self._wp),
                                                                            _stream(stream)))

    def add_decompose_small_values_scaled_dev(self, small_values, acc, value_count: int, factors, stream=None):
        (pi, ni), (pa, na) = self._dev(small_values), self._dev(acc)
        if ni != value_count:
            raise PfheError(32, "small_values.len() must equal value_count")
        f = self._factor_words(factors, self.moduli_count())
        check(self._f("add_decompose_small_values_scaled_dev")(self._h, pi, value_count, pa, na,
                                                                   f.ctypes.data_as(self._wp), _stream(stream)))

    def decompose_big_uint_values_to(self, big_uint_values, multi_residues, value_count: int):
        """base.rs:457-481: value_count little-endian big integers -> modulus-major residues."""
        pi, ni = self._host(big_uint_values)
        po, no = self._host(multi_residues)
        check(self._f("decompose_big_uint_values_to")(self._h, pi, ni, po, no, value_count))

    def decompose_big_uint_values_to_dev(self, big_uint_values, multi_residues, value_count: int, stream=None):
        (pi, ni), (po, no) = self._dev(big_uint_values), self._dev(multi_residues)
        check(self._f("decompose_big_uint_values_to_dev")(self._h, pi, ni, po, no, value_count, _stream(stream)))


class RNSBase32(RNSBase):
    """primus_rns::RNSBase<u32, BarrettModulus<u32>> (base.rs:26-37): moduli below 2^30; residues and the limbs of
    big integers are uint32 (big_uint_value_len counts u32 limbs).  ShoupFactor pairs are (value, floor(value*2^32/q))."""

    _pre, _dtype, _wp = "pfhe_rns32_", np.uint32, C.POINTER(C.c_uint32)
    _host, _dev = staticmethod(_host32), staticmethod(_dev32)


class BaseConverter:
    """primus_rns::BaseConverter — precomputed converter between two RNS bases (converter.rs:21-69).

    Arrays are modulus-major like the reference's; its `scratch` argument has no counterpart.  BaseConverter32 below is
    the <u32> instantiation (uint32 arrays, RNSBase32 bases)."""

    _pre, _dtype, _base = "pfhe_conv_", np.uint64, RNSBase
    _host, _dev = staticmethod(_host), staticmethod(_dev)

    def _f(self, name):
        return getattr(lib(), self._pre + name)

    def __init__(self, input_base, output_base):
        if not (type(input_base) is self._base and type(output_base) is self._base):
            raise TypeError(f"{type(self).__name__} converts between two {self._base.__name__} bases")
        h = C.c_void_p()
        check(self._f("create")(input_base._h, output_base._h, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._f("destroy")(h)
            self._h = None

    def input_moduli_count(self) -> int: return int(self._f("input_moduli_count")(self._h))
    def output_moduli_count(self) -> int: return int(self._f("output_moduli_count")(self._h))

    def base_change_matrix(self) -> np.ndarray:
        out = np.empty(self.input_moduli_count() * self.output_moduli_count(), self._dtype)
        check(self._f("base_change_matrix")(self._h, *self._host(out)))
        return out

    def fast_convert_array(self, crt_poly_in, crt_poly_out, poly_length: int):
        """converter.rs:192-218."""
        (pi, ni), (po, no) = self._host(crt_poly_in), self._host(crt_poly_out)
        check(self._f("fast_convert_array")(self._h, pi, ni, po, no, poly_length))

    def exact_convert_array(self, crt_poly_in, crt_poly_out, poly_length: int):
        """converter.rs:274-364 (single output modulus)."""
        (pi, ni), (po, no) = self._host(crt_poly_in), self._host(crt_poly_out)
        check(self._f("exact_convert_array")(self._h, pi, ni, po, no, poly_length))

    def fast_convert_array_dev(self, crt_poly_in, crt_poly_out, poly_length: int, stream=None):
        (pi, ni), (po, no) = self._dev(crt_poly_in), self._dev(crt_poly_out)
        check(self._f("fast_convert_array_dev")(self._h, pi, ni, po, no, poly_length, _stream(stream)))

    def fast_convert_array_to_pairs_dev(self, crt_poly_in, pairs_out, poly_length: int, stream=None):
        """converter.rs:233-272 (fast_convert_array_to_pair_iter): interleaved (mod p_0, mod p_1) pairs."""
        (pi, ni), (po, no) = self._dev(crt_poly_in), self._dev(pairs_out)
        check(self._f("fast_convert_array_to_pairs_dev")(self._h, pi, ni, po, no, poly_length, _stream(stream)))

    def exact_convert_array_dev(self, crt_poly_in, crt_poly_out, poly_length: int, stream=None):
        (pi, ni), (po, no) = self._dev(crt_poly_in), self._dev(crt_poly_out)
        check(self._f("exact_convert_array_dev")(self._h, pi, ni, po, no, poly_length, _stream(stream)))


class BaseConverter32(BaseConverter):
    """primus_rns::BaseConverter<u32, BarrettModulus<u32>> (converter.rs:21, generic over T: FheUint)."""

    _pre, _dtype, _base = "pfhe_conv32_", np.uint32, RNSBase32
    _host, _dev = staticmethod(_host32), staticmethod(_dev32)


class BigUintApproxSignedBasis:
    """primus_decompose::big_integer::BigUintApproxSignedBasis<u64> (basis.rs:17-211); BigUintApproxSignedBasis32 is
    the <u32> instantiation (the type the reference's own tests/big_uint.rs runs)."""

    _pre, _dtype = "pfhe_basis_", np.uint64
    _host, _dev = staticmethod(_host), staticmethod(_dev)

    def _f(self, name):
        return getattr(lib(), self._pre + name)

    def __init__(self, rns_base: RNSBase, log_basis: int, reverse_length: int | None = None):
        h = C.c_void_p()
        check(self._f("create")(rns_base._h, log_basis, reverse_length or 0, C.byref(h)))
        self._h = h
        self.rns_base = rns_base

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._f("destroy")(h)
            self._h = None

    def decompose_length(self) -> int: return int(self._f("decompose_length")(self._h))
    def log_basis(self) -> int: return int(self._f("log_basis")(self._h))
    def drop_bits(self) -> int: return int(self._f("drop_bits")(self._h))
    def basis_value(self) -> int: return int(self._f("basis_value")(self._h))

    def scalars(self) -> np.ndarray:
        out = np.empty(self.decompose_length() * self.rns_base.big_uint_value_len(), self._dtype)
        check(self._f("scalars")(self._h, *self._host(out)))
        return out

    def scalars_residue(self) -> np.ndarray:
        out = np.empty(self.decompose_length() * self.rns_base.moduli_count(), self._dtype)
        check(self._f("scalars_residue")(self._h, *self._host(out)))
        return out

    def init_value_carry_slice_inplace(self, values, carries, big_uint_value_len: int | None = None):
        """basis.rs:326-367; carries is a numpy uint8 (bool) array, one entry per value."""
        pv, nv = self._host(values)
        assert carries.dtype in (np.uint8, np.bool_) and carries.flags.c_contiguous
        check(self._f("init_value_carry_slice_inplace")(self._h, pv, nv, carries.ctypes.data_as(C.c_void_p),
                                                              carries.size))

    def init_value_carry_slice_to(self, big_uint_values, adjust_big_uint_values, carries):
        """basis.rs:371-420: out-of-place form of init_value_carry_slice_inplace."""
        (pv, nv), (pa, na) = self._host(big_uint_values), self._host(adjust_big_uint_values)
        assert carries.dtype in (np.uint8, np.bool_) and carries.flags.c_contiguous
        if na != nv:
            raise PfheError(32, "values and adjusted values differ in length")
        check(self._f("init_value_carry_slice_to")(self._h, pv, nv, pa, carries.ctypes.data_as(C.c_void_p),
                                                         carries.size))

    def decompose_slice_to(self, level: int, big_uint_values, decomposed_big_uint_values, carries):
        """decomposer_iter().nth(level).decompose_slice_to(...) — common.rs:289-306: signed digits modulo Q."""
        (pv, nv), (pd, nd) = self._host(big_uint_values), self._host(decomposed_big_uint_values)
        check(self._f("decompose_slice_to")(self._h, level, pv, nv, pd, nd, carries.ctypes.data_as(C.c_void_p),
                                                  carries.size))

    def init_value_carry_slice_to_dev(self, values, adjusted, carries, stream=None):
        (pv, nv), (pa, na) = self._dev(values), self._dev(adjusted)
        if na != nv:
            raise PfheError(32, "values and adjusted values differ in length")
        check(self._f("init_value_carry_slice_to_dev")(self._h, pv, nv, pa, C.c_void_p(carries.data_ptr()),
                                                             carries.numel(), _stream(stream)))

    def decompose_slice_to_dev(self, level: int, values, decomposed, carries, stream=None):
        (pv, nv), (pd, nd) = self._dev(values), self._dev(decomposed)
        check(self._f("decompose_slice_to_dev")(self._h, level, pv, nv, pd, nd, C.c_void_p(carries.data_ptr()),
                                                      carries.numel(), _stream(stream)))

    def unsigned_decompose_slice_to(self, level: int, big_uint_values, decomposed_unsigned_values, carries):
        """decomposer_iter().nth(level).unsigned_decompose_slice_to(...) — common.rs:309-325."""
        pv, nv = self._host(big_uint_values)
        pd, nd = self._host(decomposed_unsigned_values)
        if nd != carries.size:
            raise PfheError(32, "carries and digits differ in length")
        check(self._f("unsigned_decompose_slice_to")(self._h, level, pv, nv, pd,
                                                           carries.ctypes.data_as(C.c_void_p), carries.size))


class BigUintApproxSignedBasis32(BigUintApproxSignedBasis):
    """BigUintApproxSignedBasis<u32> over an RNSBase32: 0 < log_basis < 32, digits and limbs uint32."""

    _pre, _dtype = "pfhe_basis32_", np.uint32
    _host, _dev = staticmethod(_host32), staticmethod(_dev32)
