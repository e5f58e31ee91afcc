"""primus-fhe hot path on MI355X: host-side mirror of the reference's operator interface.

The classes keep the reference's names and argument meaning
(`U64NttTable`, `U64DcrtTable`, ... — primus_ntt / primus_poly / primus_rns / primus_decompose /
primus_lattice) and delegate through the C ABI in include/pfhe.h to hand-written HIP kernels
(csrc/).  There is no CPU fallback: if libpfhe_hip.so is missing or no GPU is visible the
constructors raise.
"""
from ._lib import PfheError, build, lib, library_path, status_string  # noqa: F401
from .ntt import NttError, U64DcrtTable, U64NttTable  # noqa: F401

__all__ = ["PfheError", "NttError", "U64NttTable", "U64DcrtTable", "build", "lib", "library_path",
           "status_string"]
