"""U64NttTable / U64DcrtTable — the reference's `NttTable` / `DcrtTable` operator surface.

Reference: primus_ntt/src/ntt/mod.rs:16-113 (trait NttTable), ntt/prime64/table.rs:41 (U64NttTable),
primus_ntt/src/dcrt/mod.rs:19-135 (trait DcrtTable), dcrt/prime64.rs:11 (U64DcrtTable).

`*_slice` methods take numpy uint64 arrays on the host and transform them in place, like the
reference's `&mut [u64]`.  `*_dev` methods take a device pointer (int) or a torch CUDA tensor and
an optional stream and are asynchronous; the batch is len / unit polynomials.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import PfheError, check, lib, u64p

NttError = PfheError  # primus_ntt::NttError variants are carried in PfheError.kind


def _host(a: np.ndarray):
    if not isinstance(a, np.ndarray) or a.dtype != np.uint64 or not a.flags.c_contiguous:
        raise TypeError("expected a C-contiguous numpy uint64 array")
    return a.ctypes.data_as(C.c_void_p), a.size


def _dev(x):
    """(device pointer, number of 64-bit words) of a torch CUDA tensor or a (ptr, words) pair."""
    if isinstance(x, tuple):
        return C.c_void_p(int(x[0])), int(x[1])
    if hasattr(x, "data_ptr"):
        if x.element_size() != 8 or not x.is_contiguous() or not x.is_cuda:
            raise TypeError("expected a contiguous 64-bit CUDA tensor")
        return C.c_void_p(x.data_ptr()), x.numel()
    raise TypeError("expected a torch CUDA tensor or a (device_ptr, words) tuple")


def _stream(stream):
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except Exception:
            pass
        return C.c_void_p(0)
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))


class U64NttTable:
    """primus_ntt::U64NttTable — negacyclic NTT over one prime q < 2^62 (table.rs:41-516)."""

    def __init__(self, log_n: int, modulus: int, device: int = 0):
        h = C.c_void_p()
        check(lib().pfhe_ntt_create(log_n, modulus, device, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().pfhe_ntt_destroy(h)
            self._h = None

    # getters (table.rs:127-161)
    def poly_length(self) -> int: return int(lib().pfhe_ntt_poly_length(self._h))
    def n(self) -> int: return self.poly_length()
    def log_n(self) -> int: return int(lib().pfhe_ntt_log_n(self._h))
    def modulus(self) -> int: return int(lib().pfhe_ntt_modulus(self._h))
    def root(self) -> int: return int(lib().pfhe_ntt_root(self._h))
    def inv_root(self) -> int: return int(lib().pfhe_ntt_inv_root(self._h))
    def inv_n(self) -> int: return int(lib().pfhe_ntt_inv_n(self._h))
    def device(self) -> int: return int(lib().pfhe_ntt_device(self._h))

    # host slices, in place (table.rs:541-563)
    def transform_slice(self, poly): check(lib().pfhe_ntt_transform_slice(self._h, *_host(poly)))
    def inverse_transform_slice(self, values): check(lib().pfhe_ntt_inverse_transform_slice(self._h, *_host(values)))
    def lazy_transform_slice(self, poly): check(lib().pfhe_ntt_lazy_transform_slice(self._h, *_host(poly)))
    def lazy_inverse_transform_slice(self, values): check(lib().pfhe_ntt_lazy_inverse_transform_slice(self._h, *_host(values)))
    transform_inplace = transform_slice                  # table.rs:523-530
    inverse_transform_inplace = inverse_transform_slice  # table.rs:532-539

    # monomial shortcuts (table.rs:565-651)
    def transform_monomial(self, coeff: int, degree: int, values):
        check(lib().pfhe_ntt_transform_monomial(self._h, coeff, degree, *_host(values)))

    def transform_coeff_one_monomial(self, degree: int, values):
        check(lib().pfhe_ntt_transform_coeff_one_monomial(self._h, degree, *_host(values)))

    def transform_coeff_minus_one_monomial(self, degree: int, values):
        check(lib().pfhe_ntt_transform_coeff_minus_one_monomial(self._h, degree, *_host(values)))

    # device path
    def transform_dev(self, poly, lazy: bool = False, stream=None):
        p, n = _dev(poly)
        check(lib().pfhe_ntt_transform_dev(self._h, p, n, int(lazy), _stream(stream)))

    def inverse_transform_dev(self, values, lazy: bool = False, stream=None):
        p, n = _dev(values)
        check(lib().pfhe_ntt_inverse_transform_dev(self._h, p, n, int(lazy), _stream(stream)))

    def transform_monomial_dev(self, coeff: int, degree: int, values, stream=None):
        p, n = _dev(values)
        check(lib().pfhe_ntt_transform_monomial_dev(self._h, coeff, degree, p, n, _stream(stream)))

    def mul_assign_dev(self, a, b, stream=None):
        """NttPolynomial::mul_assign (primus_poly/src/ntt/mul.rs:84-90)."""
        (pa, na), (pb, nb) = _dev(a), _dev(b)
        check(lib().pfhe_ntt_mul_assign_dev(self._h, pa, na, pb, nb, _stream(stream)))

    def add_mul_assign_dev(self, acc, a, b, stream=None):
        """NttPolynomial::add_mul_assign (primus_poly/src/ntt/mod.rs:101-112)."""
        (pc, nc), (pa, na), (pb, nb) = _dev(acc), _dev(a), _dev(b)
        if nc != na:
            raise PfheError(32, "acc and a differ in length")
        check(lib().pfhe_ntt_add_mul_assign_dev(self._h, pc, pa, na, pb, nb, _stream(stream)))

    def mul_to_dev(self, a, b, out, stream=None):
        """NttPolynomial::mul_to (primus_poly/src/ntt/mul.rs:100-107): out = a*b."""
        (pa, na), (pb, nb), (po, no) = _dev(a), _dev(b), _dev(out)
        if no != na:
            raise PfheError(32, "output and multiplicand differ in length")
        check(lib().pfhe_ntt_mul_to_dev(self._h, pa, na, pb, nb, po, _stream(stream)))

    def mul_add_to_dev(self, a, b, c, out, stream=None):
        """NttPolynomial::mul_add_to (primus_poly/src/ntt/mod.rs:169-187): out = a*b + c."""
        (pa, na), (pb, nb), (pc, nc), (po, no) = _dev(a), _dev(b), _dev(c), _dev(out)
        if no != na or nc != na:
            raise PfheError(32, "operands differ in length")
        check(lib().pfhe_ntt_mul_add_to_dev(self._h, pa, na, pb, nb, pc, po, _stream(stream)))


class U64DcrtTable:
    """primus_ntt::U64DcrtTable — one U64NttTable per RNS limb, modulus-major data (dcrt/prime64.rs)."""

    def __init__(self, log_n: int, moduli, device: int = 0):
        arr = np.ascontiguousarray(np.array([int(m) for m in moduli], dtype=np.uint64))
        h = C.c_void_p()
        check(lib().pfhe_dcrt_create(log_n, arr.ctypes.data_as(u64p), arr.size, device, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().pfhe_dcrt_destroy(h)
            self._h = None

    def poly_length(self) -> int: return int(lib().pfhe_dcrt_poly_length(self._h))
    def moduli_count(self) -> int: return int(lib().pfhe_dcrt_moduli_count(self._h))
    def crt_poly_length(self) -> int: return int(lib().pfhe_dcrt_crt_poly_length(self._h))
    def device(self) -> int: return int(lib().pfhe_dcrt_device(self._h))
    def moduli(self): return [int(lib().pfhe_dcrt_modulus(self._h, i)) for i in range(self.moduli_count())]
    def roots(self): return [int(lib().pfhe_dcrt_root(self._h, i)) for i in range(self.moduli_count())]

    def transform_slice(self, poly): check(lib().pfhe_dcrt_transform_slice(self._h, *_host(poly)))
    def inverse_transform_slice(self, poly): check(lib().pfhe_dcrt_inverse_transform_slice(self._h, *_host(poly)))
    def lazy_transform_slice(self, poly): check(lib().pfhe_dcrt_lazy_transform_slice(self._h, *_host(poly)))
    def lazy_inverse_transform_slice(self, poly): check(lib().pfhe_dcrt_lazy_inverse_transform_slice(self._h, *_host(poly)))
    transform_inplace = transform_slice
    inverse_transform_inplace = inverse_transform_slice

    def transform_monomial(self, coeff: int, degree: int, values):
        check(lib().pfhe_dcrt_transform_monomial(self._h, coeff, degree, *_host(values)))

    def transform_coeff_one_monomial(self, degree: int, values):
        check(lib().pfhe_dcrt_transform_coeff_one_monomial(self._h, degree, *_host(values)))

    def transform_coeff_minus_one_monomial(self, degree: int, values):
        """-X^degree: q_i - 1 in limb i (primus_ntt/src/dcrt/mod.rs:124-134)."""
        check(lib().pfhe_dcrt_transform_coeff_minus_one_monomial(self._h, degree, *_host(values)))

    def transform_monomial_dev(self, coeff: int, degree: int, values, stream=None):
        """DcrtTable::transform_monomial into device memory: launches only, capturable."""
        p, n = _dev(values)
        check(lib().pfhe_dcrt_transform_monomial_dev(self._h, coeff, degree, p, n, 0, _stream(stream)))

    def transform_coeff_one_monomial_dev(self, degree: int, values, stream=None):
        self.transform_monomial_dev(1, degree, values, stream)

    def transform_coeff_minus_one_monomial_dev(self, degree: int, values, stream=None):
        p, n = _dev(values)
        check(lib().pfhe_dcrt_transform_monomial_dev(self._h, 0, degree, p, n, 1, _stream(stream)))

    def transform_dev(self, poly, lazy: bool = False, stream=None):
        p, n = _dev(poly)
        check(lib().pfhe_dcrt_transform_dev(self._h, p, n, int(lazy), _stream(stream)))

    def inverse_transform_dev(self, poly, lazy: bool = False, stream=None):
        p, n = _dev(poly)
        check(lib().pfhe_dcrt_inverse_transform_dev(self._h, p, n, int(lazy), _stream(stream)))

    def mul_assign_dev(self, a, b, stream=None):
        """DcrtPolynomial::mul_assign (primus_poly/src/dcrt/mul.rs:176-187); b may be one shared polynomial."""
        (pa, na), (pb, nb) = _dev(a), _dev(b)
        check(lib().pfhe_dcrt_mul_assign_dev(self._h, pa, na, pb, nb, _stream(stream)))

    def add_mul_assign_dev(self, acc, a, b, stream=None):
        """DcrtPolynomial::add_mul_assign (primus_poly/src/dcrt/mod.rs:105-123)."""
        (pc, nc), (pa, na), (pb, nb) = _dev(acc), _dev(a), _dev(b)
        if nc != na:
            raise PfheError(32, "acc and a differ in length")
        check(lib().pfhe_dcrt_add_mul_assign_dev(self._h, pc, pa, na, pb, nb, _stream(stream)))

    def mul_to_dev(self, a, b, out, stream=None):
        """NttPolynomial::mul_to (primus_poly/src/ntt/mul.rs:100-107): out = a*b."""
        (pa, na), (pb, nb), (po, no) = _dev(a), _dev(b), _dev(out)
        if no != na:
            raise PfheError(32, "output and multiplicand differ in length")
        check(lib().pfhe_dcrt_mul_to_dev(self._h, pa, na, pb, nb, po, _stream(stream)))

    def mul_add_to_dev(self, a, b, c, out, stream=None):
        """NttPolynomial::mul_add_to (primus_poly/src/ntt/mod.rs:169-187): out = a*b + c."""
        (pa, na), (pb, nb), (pc, nc), (po, no) = _dev(a), _dev(b), _dev(c), _dev(out)
        if no != na or nc != na:
            raise PfheError(32, "operands differ in length")
        check(lib().pfhe_dcrt_mul_add_to_dev(self._h, pa, na, pb, nb, pc, po, _stream(stream)))

    def add_dcrt_glwe_mul_dcrt_polynomial_assign_dev(self, acc, dcrt_glwe, dcrt_poly, glwe_polys: int, stream=None):
        """DcrtGlwe::add_dcrt_glwe_mul_dcrt_polynomial_assign (primus_lattice/src/glwe/dcrt.rs:107-126) over a
        batch: acc[e][c] += dcrt_glwe[e][c] * dcrt_poly[e], c < glwe_polys = k + 1."""
        (pc, nc), (pa, na), (pb, nb) = _dev(acc), _dev(dcrt_glwe), _dev(dcrt_poly)
        if nc != na:
            raise PfheError(32, "accumulator and ciphertext differ in length")
        check(lib().pfhe_dcrt_add_dcrt_glwe_mul_dcrt_polynomial_assign_dev(self._h, pc, pa, na, pb, nb, glwe_polys,
                                                                           _stream(stream)))

    def transform_form(self, words: int, inverse: bool = False):
        """(name, launches): how transform_dev / inverse_transform_dev will run `words` words of data."""
        buf, k = C.create_string_buffer(96), C.c_int(0)
        check(lib().pfhe_dcrt_transform_form(self._h, words, int(inverse), buf, len(buf), C.byref(k)))
        return buf.value.decode(), int(k.value)

    def fill_uniform_dev(self, dst, seed: int, stream=None):
        """Synthetic residues (bench / test input): uniform in [0, q_limb) from SplitMix64(seed)."""
        p, n = _dev(dst)
        mods = np.array(self.moduli(), np.uint64)
        check(lib().pfhe_fill_uniform_dev(self.device(), p, n, mods.ctypes.data_as(u64p), mods.size, self.poly_length(),
                                          seed, _stream(stream)))

    # ---- element-wise family on canonical residues (CrtPolynomial / DcrtPolynomial / CrtGlwe) ----
    def _same_len(self, *bufs):
        ptrs = [_dev(b) for b in bufs]
        if any(n != ptrs[0][1] for _, n in ptrs):
            raise PfheError(32, "operands differ in length")
        return [p for p, _ in ptrs], ptrs[0][1]

    def _host_words(self, values, per_limb: int):
        arr = np.ascontiguousarray(np.array(values, dtype=np.uint64).reshape(-1))
        if arr.size != per_limb * self.moduli_count():
            raise PfheError(32, "expected one entry per modulus")
        return arr

    def add_to_dev(self, a, b, out, stream=None):
        """CrtPolynomial::add_to / add_assign (primus_poly/src/crt/add.rs:28-70); CrtGlwe::add_element_wise_to
        (primus_lattice/src/macros/mod.rs:472-500).  out may alias a."""
        (pa, pb, po), n = self._same_len(a, b, out)
        check(lib().pfhe_dcrt_add_to_dev(self._h, pa, pb, po, n, _stream(stream)))

    def sub_to_dev(self, a, b, out, stream=None):
        """CrtPolynomial::sub_to / sub_assign / sub_rev_assign (crt/sub.rs:26-82): out = a - b; out may alias
        a (sub_assign) or b (sub_rev_assign)."""
        (pa, pb, po), n = self._same_len(a, b, out)
        check(lib().pfhe_dcrt_sub_to_dev(self._h, pa, pb, po, n, _stream(stream)))

    def neg_to_dev(self, a, out, stream=None):
        """CrtPolynomial::neg_to / neg_assign (crt/neg.rs:25-53)."""
        (pa, po), n = self._same_len(a, out)
        check(lib().pfhe_dcrt_neg_to_dev(self._h, pa, po, n, _stream(stream)))

    def mul_scalar_to_dev(self, a, scalars, out, stream=None):
        """CrtPolynomial::mul_scalar_to / mul_scalar_assign (crt/mul.rs:26-34,138-158); CrtGlwe::mul_scalar_to
        (glwe/crt.rs:132-150): per-limb scalar residues."""
        (pa, po), n = self._same_len(a, out)
        sc = self._host_words(scalars, 1)
        check(lib().pfhe_dcrt_mul_scalar_to_dev(self._h, pa, sc.ctypes.data_as(u64p), po, n, _stream(stream)))

    def add_mul_scalar_assign_dev(self, acc, rhs, scalars, stream=None):
        """CrtPolynomial::add_mul_scalar_assign (crt/mul.rs:57-77): acc += scalar * rhs."""
        (pc, pr), n = self._same_len(acc, rhs)
        sc = self._host_words(scalars, 1)
        check(lib().pfhe_dcrt_add_mul_scalar_assign_dev(self._h, pc, pr, sc.ctypes.data_as(u64p), n, _stream(stream)))

    def mul_factor_to_dev(self, a, factors, out, stream=None):
        """CrtPolynomial::mul_factor_to / mul_factor_assign (crt/mul.rs:47-54,161-180); CrtGlwe::mul_factor_to
        (glwe/crt.rs:153-171): per-limb ShoupFactor (value, quotient) pairs."""
        (pa, po), n = self._same_len(a, out)
        f = self._host_words(factors, 2)
        check(lib().pfhe_dcrt_mul_factor_to_dev(self._h, pa, f.ctypes.data_as(u64p), po, n, _stream(stream)))

    def add_mul_factor_assign_dev(self, acc, rhs, factors, stream=None):
        """CrtPolynomial::add_mul_factor_assign (crt/mul.rs:80-99): acc += factor * rhs."""
        (pc, pr), n = self._same_len(acc, rhs)
        f = self._host_words(factors, 2)
        check(lib().pfhe_dcrt_add_mul_factor_assign_dev(self._h, pc, pr, f.ctypes.data_as(u64p), n, _stream(stream)))

    def mul_monomial_to_dev(self, a, r: int, out, stream=None):
        """out = a * X^r (0 <= r < 2N) per polynomial; out-of-place form of CrtPolynomial::mul_monomial_assign."""
        (pa, po), n = self._same_len(a, out)
        check(lib().pfhe_dcrt_mul_monomial_to_dev(self._h, pa, r, po, n, _stream(stream)))

    def mul_monomial_assign_dev(self, data, r: int, stream=None):
        """CrtPolynomial::mul_monomial_assign (crt/mul.rs:102-127); CrtGlwe::mul_monic_monomial_assign
        (glwe/crt.rs:76-113)."""
        p, n = _dev(data)
        check(lib().pfhe_dcrt_mul_monomial_assign_dev(self._h, p, r, n, _stream(stream)))

    def inv_to_dev(self, a, out, stream=None):
        """DcrtPolynomial::inv_to / inv_assign (dcrt/inv.rs:33-68): point-wise inverse; raises NoInverse where the
        reference panics."""
        (pa, po), n = self._same_len(a, out)
        check(lib().pfhe_dcrt_inv_to_dev(self._h, pa, po, n, _stream(stream)))

    def glwe_mul_dcrt_polynomial_to_dev(self, dcrt_glwe, dcrt_poly, result, glwe_polys: int, stream=None):
        """DcrtGlwe::mul_dcrt_polynomial_to (primus_lattice/src/glwe/dcrt.rs:377-395) over a batch:
        result[e][c] = dcrt_glwe[e][c] * dcrt_poly[e], c < glwe_polys = k + 1."""
        (pa, na), (pb, nb), (pr, nr) = _dev(dcrt_glwe), _dev(dcrt_poly), _dev(result)
        if nr != na:
            raise PfheError(32, "result and ciphertext differ in length")
        check(lib().pfhe_dcrt_glwe_mul_dcrt_polynomial_to_dev(self._h, pa, na, pb, nb, glwe_polys, pr, _stream(stream)))

    def butterfly_mul_dcrt_polynomial_to_dev(self, a, rhs, dcrt_poly, result, stream=None):
        """DcrtGlwe::butterfly_mul_dcrt_polynomial_to (primus_lattice/src/glwe/dcrt.rs:128-155):
        (a, result) = (a + rhs, (a_orig - rhs) * dcrt_poly)."""
        (pa, na), (ps, ns), (pw, nw), (pr, nr) = _dev(a), _dev(rhs), _dev(dcrt_poly), _dev(result)
        if not (na == ns == nr):
            raise PfheError(32, "a, rhs and result differ in length")
        check(lib().pfhe_dcrt_butterfly_mul_dcrt_polynomial_to_dev(self._h, pa, ps, na, pw, nw, pr, _stream(stream)))

    def butterfly_mul_factor_to_dev(self, a, rhs, factor_poly, result, stream=None):
        """DcrtGlwe::butterfly_mul_factor_to (glwe/dcrt.rs:157-175): factor_poly holds
        ShoupFactor<u64> (value, quotient) pairs, two words per coefficient."""
        (pa, na), (ps, ns), (pw, nw), (pr, nr) = _dev(a), _dev(rhs), _dev(factor_poly), _dev(result)
        if not (na == ns == nr):
            raise PfheError(32, "a, rhs and result differ in length")
        check(lib().pfhe_dcrt_butterfly_mul_factor_to_dev(self._h, pa, ps, na, pw, nw, pr, _stream(stream)))

    def mul_dcrt_polynomial_dev(self, crt_poly, dcrt_poly, stream=None):
        """CrtRlwe::mul_dcrt_polynomial_to + into_coeff_form (primus_lattice/src/rlwe/crt.rs:42-65,
        macros/mod.rs:901-911): NTT -> pointwise multiply -> INTT, in place."""
        (pa, na), (pb, nb) = _dev(crt_poly), _dev(dcrt_poly)
        check(lib().pfhe_dcrt_mul_dcrt_polynomial_dev(self._h, pa, na, pb, nb, _stream(stream)))


# ---------------------------------------------------------------------------------------------
# u32 tables — primus_ntt::U32NttTable (ntt/prime32/table.rs:37) / U32DcrtTable (dcrt/prime32.rs:11)
# ---------------------------------------------------------------------------------------------

def _host32(a: np.ndarray):
    if not isinstance(a, np.ndarray) or a.dtype != np.uint32 or not a.flags.c_contiguous:
        raise TypeError("expected a C-contiguous numpy uint32 array")
    return a.ctypes.data_as(C.c_void_p), a.size


def _dev32(x):
    """(device pointer, number of 32-bit words) of a torch CUDA tensor or a (ptr, words) pair."""
    if isinstance(x, tuple):
        return C.c_void_p(int(x[0])), int(x[1])
    if hasattr(x, "data_ptr"):
        if x.element_size() != 4 or not x.is_contiguous() or not x.is_cuda:
            raise TypeError("expected a contiguous 32-bit CUDA tensor")
        return C.c_void_p(x.data_ptr()), x.numel()
    raise TypeError("expected a torch CUDA tensor or a (device_ptr, words) tuple")


class _U32Common:
    """Methods shared by the two u32 tables; `_pre` is the C symbol prefix."""

    _pre = ""

    def _f(self, name):
        return getattr(lib(), self._pre + name)

    def transform_slice(self, poly): check(self._f("transform_slice")(self._h, *_host32(poly)))
    def inverse_transform_slice(self, values): check(self._f("inverse_transform_slice")(self._h, *_host32(values)))
    def lazy_transform_slice(self, poly): check(self._f("lazy_transform_slice")(self._h, *_host32(poly)))
    def lazy_inverse_transform_slice(self, values): check(self._f("lazy_inverse_transform_slice")(self._h, *_host32(values)))

    def transform_monomial(self, coeff: int, degree: int, values):
        check(self._f("transform_monomial")(self._h, coeff, degree, *_host32(values)))

    def transform_coeff_one_monomial(self, degree: int, values):
        check(self._f("transform_coeff_one_monomial")(self._h, degree, *_host32(values)))

    def transform_coeff_minus_one_monomial(self, degree: int, values):
        check(self._f("transform_coeff_minus_one_monomial")(self._h, degree, *_host32(values)))

    def transform_dev(self, poly, lazy: bool = False, stream=None):
        p, n = _dev32(poly)
        check(self._f("transform_dev")(self._h, p, n, int(lazy), _stream(stream)))

    def inverse_transform_dev(self, values, lazy: bool = False, stream=None):
        p, n = _dev32(values)
        check(self._f("inverse_transform_dev")(self._h, p, n, int(lazy), _stream(stream)))

    def mul_assign_dev(self, a, b, stream=None):
        """a *= b pointwise (b: same length, or one polynomial shared by the batch)."""
        pa, na = _dev32(a)
        pb, nb = _dev32(b)
        check(self._f("mul_assign_dev")(self._h, pa, na, pb, nb, _stream(stream)))

    def add_mul_assign_dev(self, acc, a, b, stream=None):
        pc, nc = _dev32(acc)
        pa, na = _dev32(a)
        pb, nb = _dev32(b)
        if nc != na:
            raise PfheError(32, "accumulator and multiplicand differ in length")
        check(self._f("add_mul_assign_dev")(self._h, pc, pa, na, pb, nb, _stream(stream)))


class U32NttTable(_U32Common):
    """primus_ntt::U32NttTable — negacyclic NTT over one prime q < 2^30, u32 data (table.rs:37-470)."""

    _pre = "pfhe_ntt32_"

    def __init__(self, log_n: int, modulus: int, device: int = 0):
        h = C.c_void_p()
        check(lib().pfhe_ntt32_create(log_n, modulus, device, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().pfhe_ntt32_destroy(h)
            except Exception:
                pass
            self._h = None

    def poly_length(self) -> int: return int(lib().pfhe_ntt32_poly_length(self._h))
    def n(self) -> int: return self.poly_length()
    def log_n(self) -> int: return int(lib().pfhe_ntt32_log_n(self._h))
    def modulus(self) -> int: return int(lib().pfhe_ntt32_modulus(self._h))
    def root(self) -> int: return int(lib().pfhe_ntt32_root(self._h))
    def inv_root(self) -> int: return int(lib().pfhe_ntt32_inv_root(self._h))
    def inv_n(self) -> int: return int(lib().pfhe_ntt32_inv_n(self._h))
    def device(self) -> int: return int(lib().pfhe_ntt32_device(self._h))

    def transform_monomial_dev(self, coeff: int, degree: int, values, stream=None):
        p, n = _dev32(values)
        check(lib().pfhe_ntt32_transform_monomial_dev(self._h, coeff, degree, p, n, _stream(stream)))


class U32DcrtTable(_U32Common):
    """primus_ntt::U32DcrtTable — one U32NttTable per RNS limb; unit = L*N words, modulus-major."""

    _pre = "pfhe_dcrt32_"

    def __init__(self, log_n: int, moduli, device: int = 0):
        arr = (C.c_uint32 * len(moduli))(*[int(m) for m in moduli])
        h = C.c_void_p()
        check(lib().pfhe_dcrt32_create(log_n, arr, len(moduli), device, C.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().pfhe_dcrt32_destroy(h)
            except Exception:
                pass
            self._h = None

    def poly_length(self) -> int: return int(lib().pfhe_dcrt32_poly_length(self._h))
    def moduli_count(self) -> int: return int(lib().pfhe_dcrt32_moduli_count(self._h))
    def transform_form(self, words: int, inverse: bool = False):
        """(name, launches): how transform_dev / inverse_transform_dev will run `words` u32 words of data."""
        buf, k = C.create_string_buffer(112), C.c_int(0)
        check(lib().pfhe_dcrt32_transform_form(self._h, words, int(inverse), buf, len(buf), C.byref(k)))
        return buf.value.decode(), int(k.value)

    def crt_poly_length(self) -> int: return int(lib().pfhe_dcrt32_crt_poly_length(self._h))
    def device(self) -> int: return int(lib().pfhe_dcrt32_device(self._h))
    def moduli(self): return [int(lib().pfhe_dcrt32_modulus(self._h, i)) for i in range(self.moduli_count())]
    def roots(self): return [int(lib().pfhe_dcrt32_root(self._h, i)) for i in range(self.moduli_count())]

    def fill_uniform_dev(self, dst, seed: int, stream=None):
        """Synthetic residues (bench input): uniform in [0, q_limb) from SplitMix64(seed)."""
        p, n = _dev32(dst)
        check(lib().pfhe_dcrt32_fill_uniform_dev(self._h, p, n, seed, _stream(stream)))

    def transform_num_passes(self) -> int:
        return int(lib().pfhe_dcrt32_transform_num_passes(self._h))

    def transform_pass_name(self, inverse: bool, index: int) -> str:
        return lib().pfhe_dcrt32_transform_pass_name(self._h, int(inverse), index).decode()

    def transform_pass_dev(self, poly, inverse: bool, index: int, lazy: bool = False, stream=None):
        p, n = _dev32(poly)
        check(lib().pfhe_dcrt32_transform_pass_dev(self._h, p, n, int(inverse), index, int(lazy), _stream(stream)))
