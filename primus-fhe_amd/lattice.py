"""RNS gadget external product — host-side mirror of the primus_lattice entry points.

Reference: CrtGlwe::mul_dcrt_ggsw_to (primus_lattice/src/glwe/crt.rs:200-227),
DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign (glwe/dcrt.rs:178-255),
DcrtGlevContext (context/glev.rs:4-68), DcrtGlwe::into_coeff_form (macros/mod.rs:901-911).

Layouts (flat uint64, exactly the reference's nesting): CrtGlwe / DcrtGlwe = (k+1) x L x N;
DcrtGlev = ell x DcrtGlwe; DcrtGgsw = (k+1) x DcrtGlev.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib
from .ntt import U64DcrtTable, _dev, _dev32, _host, _host32, _stream
from .rns import BigUintApproxSignedBasis, RNSBase


class DcrtGlevContext:
    """Working context of the external product (context/glev.rs:4-68): bundles basis, table and
    RNS base and owns the device scratch.  It is `&mut` in the reference; here it has one holder at a time: a call from a
    second thread while one is inside raises PfheError (Busy, "plan in use"); successive calls on different streams
    are ordered by the library."""

    _pre = "pfhe_extprod_"
    _host, _dev = staticmethod(_host), staticmethod(_dev)

    def _f(self, name):
        return getattr(lib(), self._pre + name)

    def __init__(self, table: U64DcrtTable, rns_base: RNSBase, basis: BigUintApproxSignedBasis,
                 glwe_dimension: int = 1, chunk: int = 0):
        h = C.c_void_p()
        check(self._f("plan_create")(table._h, rns_base._h, basis._h, glwe_dimension, chunk, C.byref(h)))
        self._h = h
        self.table, self.rns_base, self.basis, self.glwe_dimension = table, rns_base, basis, glwe_dimension

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._f("plan_destroy")(h)
            self._h = None

    def scratch_bytes(self) -> int:
        return int(self._f("plan_scratch_bytes")(self._h))

    def in_use(self) -> bool:
        """True while some thread is inside a call on this context."""
        return bool(self._f("plan_in_use")(self._h))

    def glwe_len(self) -> int:
        return (self.glwe_dimension + 1) * self.table.crt_poly_length()

    def ggsw_len(self) -> int:
        return (self.glwe_dimension + 1) * self.basis.decompose_length() * self.glwe_len()


class DcrtGlevContext32(DcrtGlevContext):
    """The same context over a U32DcrtTable, an RNSBase32 and a BigUintApproxSignedBasis32: every function of this
    module takes it in place of a DcrtGlevContext and then works on uint32 arrays / 32-bit CUDA tensors
    (CrtGlwe<u32>::mul_dcrt_ggsw_to, glwe/crt.rs:200-227 with dcrt/prime32.rs:11)."""

    _pre = "pfhe_extprod32_"
    _host, _dev = staticmethod(_host32), staticmethod(_dev32)


def mul_dcrt_ggsw_to(crt_glwe, dcrt_ggsw, result, context: DcrtGlevContext, into_coeff_form: bool = False):
    """CrtGlwe::mul_dcrt_ggsw_to on host numpy arrays (batched: concatenated ciphertexts)."""
    (pa, na), (pk, nk), (pr, nr) = context._host(crt_glwe), context._host(dcrt_ggsw), context._host(result)
    check(context._f("mul_dcrt_ggsw_to")(context._h, pa, na, pk, nk, pr, nr, int(into_coeff_form)))


def mul_dcrt_ggsw_to_dev(crt_glwe, dcrt_ggsw, result, context: DcrtGlevContext, into_coeff_form: bool = False,
                         stream=None):
    """Device-pointer variant (torch CUDA tensors or (ptr, words) tuples), asynchronous."""
    (pa, na), (pk, nk), (pr, nr) = context._dev(crt_glwe), context._dev(dcrt_ggsw), context._dev(result)
    check(context._f("mul_dcrt_ggsw_to_dev")(context._h, pa, na, pk, nk, pr, nr, int(into_coeff_form),
                                                  _stream(stream)))


def profile_mul_dcrt_ggsw_to_dev(crt_glwe, dcrt_ggsw, result, context: DcrtGlevContext, stream=None):
    """Measurement aid: the product (NTT-form output) with HIP events between its kernel groups.  Returns
    (ms of digit extraction + lifting strided pass, ms of block pass + multiply-accumulate, launches of each)."""
    import ctypes as C
    (pa, na), (pk, nk), (pr, nr) = context._dev(crt_glwe), context._dev(dcrt_ggsw), context._dev(result)
    ms = (C.c_double * 2)()
    launches = C.c_size_t(0)
    check(context._f("profile_dev")(context._h, pa, na, pk, nk, pr, nr, ms, C.byref(launches), _stream(stream)))
    return float(ms[0]), float(ms[1]), int(launches.value)


def add_dcrt_glev_mul_crt_poly_assign_dev(acc, dcrt_glev, crt_poly, context: DcrtGlevContext, stream=None):
    """DcrtGlwe::add_dcrt_glev_mul_crt_poly_assign (glwe/dcrt.rs:178-255): acc += glev (x) crt_poly."""
    (pc, nc), (pg, ng), (pp, np_) = context._dev(acc), context._dev(dcrt_glev), context._dev(crt_poly)
    check(context._f("add_dcrt_glev_mul_crt_poly_assign_dev")(context._h, pc, nc, pg, ng, pp, np_,
                                                                   _stream(stream)))


def glev_mul_crt_poly_to_dev(dcrt_glev, crt_poly, result, context: DcrtGlevContext, stream=None):
    """DcrtGlev::mul_crt_poly_to (primus_lattice/src/glev/dcrt.rs:45-110): result = glev (x) crt_poly."""
    (pg, ng), (pp, np_), (pr, nr) = context._dev(dcrt_glev), context._dev(crt_poly), context._dev(result)
    check(context._f("glev_mul_crt_poly_to_dev")(context._h, pg, ng, pp, np_, pr, nr, _stream(stream)))


def add_dcrt_glev_mul_big_uint_poly_assign_dev(acc, dcrt_glev, big_uint_poly, context: DcrtGlevContext, stream=None):
    """DcrtGlwe::add_dcrt_glev_mul_big_uint_poly_assign (glwe/dcrt.rs:258-338): acc += glev (x) big_uint_poly, the
    polynomial given as big integers modulo Q (big_uint_value_len limbs per coefficient)."""
    (pc, nc), (pg, ng), (pp, np_) = context._dev(acc), context._dev(dcrt_glev), context._dev(big_uint_poly)
    check(context._f("add_dcrt_glev_mul_big_uint_poly_assign_dev")(context._h, pc, nc, pg, ng, pp, np_,
                                                                        _stream(stream)))


def glev_mul_big_uint_poly_to_dev(dcrt_glev, big_uint_poly, result, context: DcrtGlevContext, stream=None):
    """DcrtGlev::mul_big_uint_poly_to (primus_lattice/src/glev/dcrt.rs:113-175): result = glev (x) big_uint_poly."""
    (pg, ng), (pp, np_), (pr, nr) = context._dev(dcrt_glev), context._dev(big_uint_poly), context._dev(result)
    check(context._f("glev_mul_big_uint_poly_to_dev")(context._h, pg, ng, pp, np_, pr, nr, _stream(stream)))
