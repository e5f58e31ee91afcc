"""primus-fhe hot path on MI355X: host-side mirror of the reference's operator interface.

The classes keep the reference's names and argument meaning
(`U64NttTable`, `U64DcrtTable`, ... — primus_ntt / primus_poly / primus_rns / primus_decompose /
primus_lattice) and delegate through the C ABI in include/pfhe.h to hand-written HIP kernels
(csrc/).  There is no CPU fallback: if libpfhe_hip.so is missing or no GPU is visible the
constructors raise.
"""
from ._lib import PfheError, build, lib, library_path, status_string  # noqa: F401
from .lattice import (DcrtGlevContext, DcrtGlevContext32, add_dcrt_glev_mul_big_uint_poly_assign_dev,  # noqa: F401
                      add_dcrt_glev_mul_crt_poly_assign_dev, glev_mul_big_uint_poly_to_dev, glev_mul_crt_poly_to_dev,
                      mul_dcrt_ggsw_to, mul_dcrt_ggsw_to_dev, profile_mul_dcrt_ggsw_to_dev)
from .ntt import NttError, U32DcrtTable, U32NttTable, U64DcrtTable, U64NttTable  # noqa: F401
from .rns import (BaseConverter, BaseConverter32, BigUintApproxSignedBasis, BigUintApproxSignedBasis32, RNSBase, RNSBase32,  # noqa: F401
                  RNSError)

__all__ = ["PfheError", "NttError", "RNSError", "U64NttTable", "U64DcrtTable", "U32NttTable", "U32DcrtTable", "RNSBase",
           "BigUintApproxSignedBasis", "BaseConverter", "BaseConverter32", "DcrtGlevContext", "RNSBase32", "BigUintApproxSignedBasis32",
           "DcrtGlevContext32", "mul_dcrt_ggsw_to", "mul_dcrt_ggsw_to_dev",
           "add_dcrt_glev_mul_crt_poly_assign_dev", "glev_mul_crt_poly_to_dev", "add_dcrt_glev_mul_big_uint_poly_assign_dev",
           "glev_mul_big_uint_poly_to_dev", "build", "lib", "library_path", "status_string"]
